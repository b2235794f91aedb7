"""Writes tests/golden/chainer_layout_{generator,discriminator}.npz: weight files with the key layout, shapes and dtypes
that `chainer.serializers.save_npz` produces for the reference's models (srgan_train.py:1351-1361; consumer
deepbedmap.py:402-408) -- built BY HAND from the layout listing ('/'-joined link paths without a
leading slash, `np.savez_compressed`, BatchNormalization persistents `avg_mean`, `avg_var` and the 0-d integer `N`),
not from this framework's own tensor tables, so that `load_npz(strict=True)` is checked against an independent statement
of the format.  No Chainer-written file exists in the reference repository (its weights live on Comet.ml), so this is
the closest available stand-in; values are a seeded pattern that compresses to a few kilobytes.

    python tests/golden/make_chainer_layout_npz.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
N_RRDB = 2  # the file layout is the same for every count; `residual_network/{i}/...` with i = 0 .. n - 1


def pattern(shape, salt):
    n = int(np.prod(shape)) if len(shape) else 1
    v = ((np.arange(n, dtype=np.int64) * 7 + salt) % 13 - 6).astype(np.float32) / np.float32(64.0)
    return v.reshape(shape)


def generator_file():
    f = {}
    salt = [0]

    def conv(path, o, c, kh, kw):
        salt[0] += 1
        f[path + "/W"] = pattern((o, c, kh, kw), salt[0])
        f[path + "/b"] = pattern((o,), salt[0] + 100)

    conv("input_block/conv_on_X", 32, 1, 3, 3)
    conv("input_block/conv_on_W1", 32, 1, 30, 30)
    conv("input_block/conv_on_W2", 32, 2, 6, 6)
    conv("input_block/conv_on_W3", 32, 1, 3, 3)
    conv("pre_residual_conv_layer", 64, 128, 3, 3)
    for i in range(N_RRDB):
        for d in (1, 2, 3):
            base = f"residual_network/{i}/residual_dense_block{d}"
            conv(base + "/conv_layer1", 32, 64, 3, 3)
            conv(base + "/conv_layer2", 32, 96, 3, 3)
            conv(base + "/conv_layer3", 32, 128, 3, 3)
            conv(base + "/conv_layer4", 32, 160, 3, 3)
            conv(base + "/conv_layer5", 64, 192, 3, 3)
    conv("post_residual_conv_layer", 64, 64, 3, 3)
    conv("post_upsample_conv_layer_1", 64, 64, 3, 3)
    conv("post_upsample_conv_layer_2", 64, 64, 3, 3)
    conv("final_conv_layer1/offset_conv", 18, 64, 3, 3)
    conv("final_conv_layer1/deform_conv", 64, 64, 3, 3)
    conv("final_conv_layer2/offset_conv", 18, 64, 3, 3)
    conv("final_conv_layer2/deform_conv", 1, 64, 3, 3)
    return f


def discriminator_file():
    f = {}
    shapes = [(64, 1, 3, 3), (64, 64, 4, 4), (128, 64, 3, 3), (128, 128, 4, 4), (128, 128, 3, 3), (256, 128, 4, 4),
              (256, 256, 3, 3), (512, 256, 4, 4), (512, 512, 3, 3), (512, 512, 4, 4)]
    for i, shp in enumerate(shapes):
        f[f"conv_layer{i}/W"] = pattern(shp, 200 + i)
    f["conv_layer0/b"] = pattern((64,), 300)  # the only convolution with a bias (nobias=True for conv_layer1..9)
    for i in range(1, 10):
        c = shapes[i][0]
        f[f"batch_norm{i}/gamma"] = 1.0 + pattern((c,), 400 + i)
        f[f"batch_norm{i}/beta"] = pattern((c,), 500 + i)
        f[f"batch_norm{i}/avg_mean"] = pattern((c,), 600 + i)
        f[f"batch_norm{i}/avg_var"] = 1.0 + np.abs(pattern((c,), 700 + i))
        f[f"batch_norm{i}/N"] = np.array(3 * i)  # 0-d integer persistent
    f["linear_1/W"] = pattern((100, 512), 800)
    f["linear_1/b"] = pattern((100,), 801)
    f["linear_2/W"] = pattern((1, 100), 802)
    f["linear_2/b"] = pattern((1,), 803)
    return f


if __name__ == "__main__":
    for name, tensors in (("generator", generator_file()), ("discriminator", discriminator_file())):
        path = os.path.join(HERE, f"chainer_layout_{name}.npz")
        np.savez_compressed(path, **tensors)
        print(path, len(tensors), "arrays", os.path.getsize(path), "bytes")
