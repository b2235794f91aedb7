"""NumPy restatement of GeneratorModel / DiscriminatorModel (srgan_train.py:201-699).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Parameters live in a flat dict keyed by
the chainer.serializers.save_npz key layout, e.g.
"residual_network/0/residual_dense_block1/conv_layer1/W".
"""
import collections
import numpy as np
from . import ops


# --------------------------------------------------------------------------------------
# parameter tables
# --------------------------------------------------------------------------------------
def generator_param_shapes(num_residual_blocks=12, out_channels=1):
    """name -> shape, in the reference's construction order (srgan_train.py:463-523)."""
    s = collections.OrderedDict()

    def conv(name, o, c, kh, kw):
        s[name + "/W"] = (o, c, kh, kw)
        s[name + "/b"] = (o,)

    # DeepbedmapInputBlock  srgan_train.py:223-254
    conv("input_block/conv_on_X", 32, 1, 3, 3)
    conv("input_block/conv_on_W1", 32, 1, 30, 30)
    conv("input_block/conv_on_W2", 32, 2, 6, 6)
    conv("input_block/conv_on_W3", 32, 1, 3, 3)
    conv("pre_residual_conv_layer", 64, 128, 3, 3)  # :467-474
    for i in range(num_residual_blocks):  # :475-477  .repeat() -> Sequential children "0".."n-1"
        for d in (1, 2, 3):  # ResInResDenseBlock :383-391
            base = f"residual_network/{i}/residual_dense_block{d}"
            # ResidualDenseBlock :292-331
            conv(base + "/conv_layer1", 32, 64, 3, 3)
            conv(base + "/conv_layer2", 32, 96, 3, 3)
            conv(base + "/conv_layer3", 32, 128, 3, 3)
            conv(base + "/conv_layer4", 32, 160, 3, 3)
            conv(base + "/conv_layer5", 64, 192, 3, 3)
    conv("post_residual_conv_layer", 64, 64, 3, 3)  # :478-485
    conv("post_upsample_conv_layer_1", 64, 64, 3, 3)  # :488-495
    conv("post_upsample_conv_layer_2", 64, 64, 3, 3)  # :496-503
    conv("final_conv_layer1/offset_conv", 18, 64, 3, 3)  # :506-514
    conv("final_conv_layer1/deform_conv", 64, 64, 3, 3)
    conv("final_conv_layer2/offset_conv", 18, 64, 3, 3)  # :515-523
    conv("final_conv_layer2/deform_conv", out_channels, 64, 3, 3)
    return s


D_CONVS = [  # (out, in, k, stride)  srgan_train.py:617-634 ; pad = 1 everywhere
    (64, 1, 3, 1),
    (64, 64, 4, 2),
    (128, 64, 3, 1),
    (128, 128, 4, 2),
    (128, 128, 3, 1),
    (256, 128, 4, 2),
    (256, 256, 3, 1),
    (512, 256, 4, 2),
    (512, 512, 3, 1),
    (512, 512, 4, 2),
]


def discriminator_param_shapes():
    s = collections.OrderedDict()
    for i, (o, c, k, _) in enumerate(D_CONVS):
        s[f"conv_layer{i}/W"] = (o, c, k, k)
        if i == 0:
            s["conv_layer0/b"] = (o,)  # only the first conv has a bias (:623)
    for i in range(1, 10):  # :636-644
        c = D_CONVS[i][0]
        s[f"batch_norm{i}/gamma"] = (c,)
        s[f"batch_norm{i}/beta"] = (c,)
    s["linear_1/W"] = (100, 512)  # :646  in_size inferred = 512*1*1 for 36x36 inputs
    s["linear_1/b"] = (100,)
    s["linear_2/W"] = (1, 100)  # :647
    s["linear_2/b"] = (1,)
    return s


def discriminator_persistent_shapes():
    s = collections.OrderedDict()
    for i in range(1, 10):
        c = D_CONVS[i][0]
        s[f"batch_norm{i}/avg_mean"] = (c,)
        s[f"batch_norm{i}/avg_var"] = (c,)
        s[f"batch_norm{i}/N"] = ()
    return s


def chainer_param_order(names):
    """Link.params() order: own params sorted by name, then children sorted by name,
    recursively (chainer/link.py).  For the flat '/'-joined keys this equals a sort where
    at each level plain params come before child links."""
    def key(n):
        parts = n.split("/")
        out = []
        for i, p in enumerate(parts):
            out.append((0 if i == len(parts) - 1 else 1, p))
        return out
    return sorted(names, key=key)


def he_normal(rng, shape, scale=0.1, dtype=np.float32):
    """chainer.initializers.HeNormal(scale=0.1, fan_option='fan_in')  srgan_train.py:220."""
    fan_in = int(np.prod(shape[1:]))
    std = scale * np.sqrt(2.0 / fan_in)
    return rng.normal(0.0, std, size=shape).astype(dtype)


def init_params(shapes, seed=42, dtype=np.float32):
    """HeNormal weights, zero biases, gamma=1, beta=0; drawn in sorted-key order."""
    rng = np.random.RandomState(seed)
    p = {}
    for name in sorted(shapes):
        shp = shapes[name]
        leaf = name.rsplit("/", 1)[1]
        if leaf == "W":
            p[name] = he_normal(rng, shp, dtype=dtype)
        elif leaf == "gamma":
            p[name] = np.ones(shp, dtype=dtype)
        else:
            p[name] = np.zeros(shp, dtype=dtype)
    return p


# --------------------------------------------------------------------------------------
# Generator
# --------------------------------------------------------------------------------------
class GeneratorModel:
    """srgan_train.py:421-576."""

    def __init__(self, num_residual_blocks=12, residual_scaling=0.1, out_channels=1, seed=42, dtype=np.float32):
        self.num_residual_blocks = num_residual_blocks
        self.residual_scaling = residual_scaling
        self.dtype = dtype
        self.shapes = generator_param_shapes(num_residual_blocks, out_channels)
        self.params = init_params(self.shapes, seed=seed, dtype=dtype)
        self.grads = None
        self.cache = None

    def count_params(self):
        return int(sum(np.prod(s) for s in self.shapes.values()))

    def _conv(self, name, x, stride=1, pad=1):
        return ops.conv2d(x, self.params[name + "/W"], self.params[name + "/b"], stride, pad)

    # ---- forward ----
    def forward(self, x, w1, w2, w3, keep=False):
        P = self.params
        rs = self.dtype(self.residual_scaling)
        c = {} if keep else None
        # input block  :256-266  (valid padding, custom strides)
        x_ = self._conv("input_block/conv_on_X", x, 1, 0)
        w1_ = self._conv("input_block/conv_on_W1", w1, 10, 0)
        w2_ = self._conv("input_block/conv_on_W2", w2, 2, 0)
        w3_ = self._conv("input_block/conv_on_W3", w3, 1, 0)
        a0 = np.concatenate([x_, w1_, w2_, w3_], axis=1)
        a1 = ops.leaky_relu(self._conv("pre_residual_conv_layer", a0))  # :541-542
        h = a1
        rrdb_in, rdb_cat = [], []
        for i in range(self.num_residual_blocks):  # :546
            xin = h
            rrdb_in.append(xin)
            for d in (1, 2, 3):  # :397-399
                base = f"residual_network/{i}/residual_dense_block{d}"
                cat = h  # grows 64 -> 192 channels  :337-353
                for k in (1, 2, 3, 4):
                    ak = ops.leaky_relu(self._conv(f"{base}/conv_layer{k}", cat))
                    cat = np.concatenate([cat, ak], axis=1)
                a5 = self._conv(f"{base}/conv_layer5", cat)
                rdb_cat.append(cat)
                h = a5 * rs + h  # :358
            h = h * rs + xin  # :402
        a2 = h
        a3 = a1 + self._conv("post_residual_conv_layer", a2)  # :550-551
        u1 = ops.upsample_nearest2(a3)  # :556-558
        a41 = ops.leaky_relu(self._conv("post_upsample_conv_layer_1", u1))  # :559-560
        u2 = ops.upsample_nearest2(a41)
        a42 = ops.leaky_relu(self._conv("post_upsample_conv_layer_2", u2))  # :567-568
        off1 = self._conv("final_conv_layer1/offset_conv", a42)  # :572 (link: offset conv then sampler)
        a51 = ops.leaky_relu(
            ops.deform_conv2d(a42, off1, P["final_conv_layer1/deform_conv/W"], P["final_conv_layer1/deform_conv/b"])
        )  # :573
        off2 = self._conv("final_conv_layer2/offset_conv", a51)
        a52 = ops.deform_conv2d(a51, off2, P["final_conv_layer2/deform_conv/W"], P["final_conv_layer2/deform_conv/b"])  # :574
        if keep:
            c.update(x=x, w1=w1, w2=w2, w3=w3, a0=a0, a1=a1, rrdb_in=rrdb_in, rdb_cat=rdb_cat, a2=a2, a3=a3,
                     u1=u1, a41=a41, u2=u2, a42=a42, off1=off1, a51=a51, off2=off2)
            self.cache = c
        return a52

    # ---- backward (fills self.grads with d loss / d param) ----
    def backward(self, gy):
        P, c = self.params, self.cache
        rs = self.dtype(self.residual_scaling)
        G = {}

        def conv_bwd(name, xin, g, stride=1, pad=1, need_gx=True):
            gx, gW, gb = ops.conv2d_backward(xin, P[name + "/W"], g, stride, pad, need_gx)
            G[name + "/W"], G[name + "/b"] = gW, gb
            return gx

        # final_conv_layer2 (deformable) :574
        gx, goff, gW, gb = ops.deform_conv2d_backward(c["a51"], c["off2"], P["final_conv_layer2/deform_conv/W"], gy)
        G["final_conv_layer2/deform_conv/W"], G["final_conv_layer2/deform_conv/b"] = gW, gb
        g_a51 = gx + conv_bwd("final_conv_layer2/offset_conv", c["a51"], goff)
        g = ops.leaky_relu_backward(c["a51"], g_a51)
        # final_conv_layer1 (deformable) :572
        gx, goff, gW, gb = ops.deform_conv2d_backward(c["a42"], c["off1"], P["final_conv_layer1/deform_conv/W"], g)
        G["final_conv_layer1/deform_conv/W"], G["final_conv_layer1/deform_conv/b"] = gW, gb
        g_a42 = gx + conv_bwd("final_conv_layer1/offset_conv", c["a42"], goff)
        g = ops.leaky_relu_backward(c["a42"], g_a42)
        g = ops.upsample_nearest2_backward(conv_bwd("post_upsample_conv_layer_2", c["u2"], g))
        g = ops.leaky_relu_backward(c["a41"], g)
        g_a3 = ops.upsample_nearest2_backward(conv_bwd("post_upsample_conv_layer_1", c["u1"], g))
        g_a1 = g_a3.copy()  # a3 = a1 + post(a2)
        g_h = conv_bwd("post_residual_conv_layer", c["a2"], g_a3)
        ridx = len(c["rdb_cat"])
        for i in reversed(range(self.num_residual_blocks)):
            g_xin = g_h.copy()  # h = h*rs + xin
            g_h = g_h * rs
            for d in (3, 2, 1):
                ridx -= 1
                base = f"residual_network/{i}/residual_dense_block{d}"
                cat = c["rdb_cat"][ridx]  # 192 channels: a0|a1|a2|a3|a4
                # out = a5*rs + a0
                gcat = np.zeros_like(cat)
                gcat[:, :64] += g_h
                g5 = g_h * rs
                gcat += conv_bwd(f"{base}/conv_layer5", cat, g5)
                for k in (4, 3, 2, 1):
                    lo = 64 + 32 * (k - 1)
                    gk = ops.leaky_relu_backward(cat[:, lo:lo + 32], gcat[:, lo:lo + 32])
                    gcat[:, :lo] += conv_bwd(f"{base}/conv_layer{k}", cat[:, :lo], gk)
                g_h = gcat[:, :64]
            g_h = g_h + g_xin
        g_a1 = g_a1 + g_h
        g = ops.leaky_relu_backward(c["a1"], g_a1)
        g_a0 = conv_bwd("pre_residual_conv_layer", c["a0"], g)
        conv_bwd("input_block/conv_on_X", c["x"], g_a0[:, 0:32], 1, 0, need_gx=False)
        conv_bwd("input_block/conv_on_W1", c["w1"], g_a0[:, 32:64], 10, 0, need_gx=False)
        conv_bwd("input_block/conv_on_W2", c["w2"], g_a0[:, 64:96], 2, 0, need_gx=False)
        conv_bwd("input_block/conv_on_W3", c["w3"], g_a0[:, 96:128], 1, 0, need_gx=False)
        self.grads = G
        return G


# --------------------------------------------------------------------------------------
# Discriminator
# --------------------------------------------------------------------------------------
class DiscriminatorModel:
    """srgan_train.py:591-699."""

    def __init__(self, seed=43, dtype=np.float32):
        self.dtype = dtype
        self.shapes = discriminator_param_shapes()
        self.params = init_params(self.shapes, seed=seed, dtype=dtype)
        self.persistent = {}
        for n, s in discriminator_persistent_shapes().items():
            leaf = n.rsplit("/", 1)[1]
            if leaf == "avg_var":
                self.persistent[n] = np.ones(s, dtype=dtype)
            elif leaf == "N":
                self.persistent[n] = np.array(0, dtype=np.int64)  # only touched in finetune mode
            else:
                self.persistent[n] = np.zeros(s, dtype=dtype)
        self.grads = None

    def count_params(self):
        return int(sum(np.prod(s) for s in self.shapes.values()))

    def forward(self, x, train=True, keep=False):
        """Returns logits (N,1); if keep, also a cache usable by backward()."""
        P, S = self.params, self.persistent
        acts = [x]
        h = ops.leaky_relu(ops.conv2d(x, P["conv_layer0/W"], P["conv_layer0/b"], 1, 1))  # :658-659
        acts.append(h)
        bn_cache, pre = [None], [None]
        for i in range(1, 10):  # :663-689
            _, _, _, s = D_CONVS[i]
            z = ops.conv2d(h, P[f"conv_layer{i}/W"], None, s, 1)
            if train:
                zn, bc = ops.batchnorm_train(z, P[f"batch_norm{i}/gamma"], P[f"batch_norm{i}/beta"],
                                             S[f"batch_norm{i}/avg_mean"], S[f"batch_norm{i}/avg_var"])
            else:
                zn, bc = ops.batchnorm_eval(z, P[f"batch_norm{i}/gamma"], P[f"batch_norm{i}/beta"],
                                            S[f"batch_norm{i}/avg_mean"], S[f"batch_norm{i}/avg_var"]), None
            h = ops.leaky_relu(zn)
            acts.append(h)
            bn_cache.append(bc)
        flat = h.reshape(len(h), -1)  # :693
        l1 = ops.leaky_relu(ops.linear(flat, P["linear_1/W"], P["linear_1/b"]))  # :694-695
        out = ops.linear(l1, P["linear_2/W"], P["linear_2/b"])  # :696
        if keep:
            return out, dict(acts=acts, bn=bn_cache, flat=flat, l1=l1)
        return out

    def backward(self, gout, cache, accumulate_into=None):
        """Accumulates parameter grads for one forward call (the D-step has two)."""
        P = self.params
        G = accumulate_into if accumulate_into is not None else {}

        def acc(name, val):
            G[name] = G[name] + val if name in G else val

        g, gW, gb = ops.linear_backward(cache["l1"], P["linear_2/W"], gout)
        acc("linear_2/W", gW), acc("linear_2/b", gb)
        g = ops.leaky_relu_backward(cache["l1"], g)
        g, gW, gb = ops.linear_backward(cache["flat"], P["linear_1/W"], g)
        acc("linear_1/W", gW), acc("linear_1/b", gb)
        g = g.reshape(cache["acts"][10].shape)
        for i in range(9, 0, -1):
            g = ops.leaky_relu_backward(cache["acts"][i + 1], g)
            g, gg, gbeta = ops.batchnorm_train_backward(g, P[f"batch_norm{i}/gamma"], cache["bn"][i])
            acc(f"batch_norm{i}/gamma", gg), acc(f"batch_norm{i}/beta", gbeta)
            g, gW, _ = ops.conv2d_backward(cache["acts"][i], P[f"conv_layer{i}/W"], g, D_CONVS[i][3], 1)
            acc(f"conv_layer{i}/W", gW)
        g = ops.leaky_relu_backward(cache["acts"][1], g)
        _, gW, gb = ops.conv2d_backward(cache["acts"][0], P["conv_layer0/W"], g, 1, 1, need_gx=False)
        acc("conv_layer0/W", gW), acc("conv_layer0/b", gb)
        self.grads = G
        return G
