"""Regenerates tests/golden/esrgan_small.npz.

The reference's numerics live in Chainer, which cannot be imported in this image, so these
vectors are produced by the repo's own oracle (pinned by the reference's known answers, tests/test_oracle_kats.py) --
they freeze the oracle against drift and give the GPU suite size-independent fixed points.  Inputs follow the reference's
fixture recipe (srgan_train.py:1101-1105); weights are the oracle's seeded HeNormal init scaled x5 so that activations,
offsets and gradients are O(1).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model as omodel  # noqa: E402
from oracle import train as otrain  # noqa: E402

N_BLOCKS, G_SEED, D_SEED, SCALE = 1, 11, 12, 5.0


def fixture_arrays(n=2):
    r = lambda *s: np.random.RandomState(seed=42).rand(*s).astype(np.float32)  # noqa: E731
    return {"X": r(n, 1, 11, 11), "W1": r(n, 1, 110, 110), "W2": r(n, 2, 22, 22), "W3": r(n, 1, 11, 11),
            "Y": r(n, 1, 36, 36)}


def build_models():
    g = omodel.GeneratorModel(num_residual_blocks=N_BLOCKS, seed=G_SEED)
    d = omodel.DiscriminatorModel(seed=D_SEED)
    for k in g.params:
        if k.endswith("/W"):
            g.params[k] *= np.float32(SCALE)
    return g, d


def compute():
    arrays = fixture_arrays()
    g, d = build_models()
    out = {}
    out["g_forward"] = g.forward(arrays["X"], arrays["W1"], arrays["W2"], arrays["W3"])
    out["d_logits_train_real"] = d.forward(arrays["Y"], train=True)
    g2, d2 = build_models()
    out["d_step"] = np.array(otrain.train_eval_discriminator(arrays, g2, d2, otrain.Adam(d2.params, alpha=1e-3, eps=1e-7)))
    out["g_step"] = np.array(otrain.train_eval_generator(arrays, g2, d2, otrain.Adam(g2.params, alpha=1e-3, eps=1e-7)))
    out["g_forward_after_step"] = g2.forward(arrays["X"], arrays["W1"], arrays["W2"], arrays["W3"])
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "esrgan_small.npz")
    np.savez_compressed(path, **compute())
    print("wrote", path, os.path.getsize(path), "bytes")
