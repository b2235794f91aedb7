"""Host-side mirror of the reference's srgan_train.py interface for the ESRGAN hot path.

Same names, argument meaning, defaults and error behaviour as /root/reference/srgan_train.py
(GeneratorModel :421-576, DiscriminatorModel :591-699, losses :841-1009, compile_srgan_model
:1014-1055, train_eval_discriminator :1084-1166, train_eval_generator :1170-1263, trainer
:1267-1329, save_model_weights_and_architecture :1333-1383) -- but every number is computed by
libdbm.so (hand-written HIP for gfx950) through its C ABI.  No Chainer, CuPy, Triton or torch on
the compute path; there is no CPU fallback.

Arrays: NumPy arrays are copied to the GPU per call (drop-in behaviour of the reference's CPU
usage); `DeviceArray` (or any object with `.data_ptr()` / `__cuda_array_interface__`, e.g. a torch
CUDA tensor) stays resident and results come back as `DeviceArray`.
"""
import contextlib
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import BF16, DbmError, KEEP_GRAPH, BN_TRAIN, DEVICE_PTRS


# --------------------------------------------------------------------------------------
# chainer.global_config / chainer.using_config stand-ins (srgan_train.py:1125, 1131, 1216, 1228)
# --------------------------------------------------------------------------------------
class _Config:
    train = True
    enable_backprop = True
    ssim_window = "gaussian"  # unpinned by the reference: "gaussian" (sigma 1.5) or "uniform"
    # "float32" (the reference's arithmetic) or "bfloat16": GeneratorModel.forward under enable_backprop=False multiplies
    # in bf16 with fp32 accumulation and fp32 storage (area inference, BASELINE.json config 5)
    dtype = "float32"
    # chainer.global_config.cudnn_deterministic, which srgan_train.py:69 sets to True at import: bitwise reproducible
    # gradients (ordered folds instead of fp32 atomics).  False buys about 2 % of the step time.  Applied by the
    # training / backward entry points.
    cudnn_deterministic = True
    # train_minibatch / trainer: one library call per minibatch (dbm_train_iteration: the generator's backward pass is
    # scheduled underneath the discriminator's) instead of the two step calls; same numbers bit for bit
    fused_iteration = True


global_config = _Config()
config = global_config


_applied_deterministic = [None]


def _apply_config(ctx):
    """Push process-wide library switches that mirror chainer.global_config (cheap: only on change)."""
    want = bool(global_config.cudnn_deterministic)
    if _applied_deterministic[0] != want:
        _lib.check(_lib.lib().dbm_set_deterministic(ctx.handle, int(want)), ctx.handle)
        _applied_deterministic[0] = want


@contextlib.contextmanager
def using_config(name, value):
    old = getattr(global_config, name)
    setattr(global_config, name, value)
    try:
        yield
    finally:
        setattr(global_config, name, old)


# --------------------------------------------------------------------------------------
# device arrays
# --------------------------------------------------------------------------------------
class DeviceArray:
    """float32 C-contiguous array resident in HBM (owned unless wrapping foreign memory)."""

    def __init__(self, shape, ctx=None, ptr=None, owner=None):
        self.ctx = ctx or _lib.default_context()
        self.shape = tuple(int(s) for s in shape)
        self.size = int(np.prod(self.shape)) if self.shape else 1
        self.dtype = np.dtype(np.float32)
        self._own = ptr is None
        self.ptr = self.ctx.malloc(4 * max(self.size, 1)) if ptr is None else int(ptr)
        self._owner = owner
        self._gen = 0  # content version: bumped by every write through this object

    @property
    def nbytes(self):
        return 4 * self.size

    def __len__(self):
        return self.shape[0]

    def data_ptr(self):
        return self.ptr

    def set(self, host):
        host = np.ascontiguousarray(host, dtype=np.float32)
        assert host.size == self.size, (host.shape, self.shape)
        self._gen += 1
        _lib.check(_lib.lib().dbm_memcpy_h2d(self.ctx.handle, C.c_void_p(self.ptr), host.ctypes.data_as(C.c_void_p),
                                             self.nbytes), self.ctx.handle)
        return self

    def get(self):
        out = np.empty(self.shape, dtype=np.float32)
        _lib.check(_lib.lib().dbm_memcpy_d2h(self.ctx.handle, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr),
                                             self.nbytes), self.ctx.handle)
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.get()
        return a.astype(dtype) if dtype is not None else a

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": "<f4", "data": (self.ptr, False), "version": 2, "strides": None}

    def __del__(self):
        try:
            if self._own and self.ptr:
                self.ctx.free(self.ptr)
                self.ptr = 0
        except Exception:
            pass


def to_device(a, ctx=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return DeviceArray(a.shape, ctx).set(a)


def _is_device(a):
    return isinstance(a, DeviceArray) or hasattr(a, "data_ptr") or (
        hasattr(a, "__cuda_array_interface__") and not isinstance(a, np.ndarray))


def _dev_ptr(a):
    if isinstance(a, DeviceArray):
        return a.ptr
    if hasattr(a, "data_ptr"):  # torch CUDA tensor (plumbing only)
        assert str(a.dtype).endswith("float32") and a.is_contiguous(), "device inputs must be contiguous float32"
        return int(a.data_ptr())
    return int(a.__cuda_array_interface__["data"][0])


def _f32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def _hp(a):
    return a.ctypes.data_as(C.c_void_p)


class Variable:
    """Minimal chainer.Variable look-alike: `.array`, `.shape`, `.backward()`."""

    def __init__(self, data, creator=None):
        self.array = data
        self._creator = creator

    @property
    def data(self):
        return self.array

    @property
    def shape(self):
        return tuple(self.array.shape)

    def backward(self):
        if self._creator is None:
            raise DbmError("this Variable is not attached to a retained graph")
        self._creator()

    def __float__(self):
        return float(np.asarray(self.array))

    def __repr__(self):
        return f"variable({np.asarray(self.array)})"


# --------------------------------------------------------------------------------------
# links
# --------------------------------------------------------------------------------------
class DeepbedmapInputBlock:
    """Marker for GeneratorModel(inblock_class=...)  (srgan_train.py:201-266); the block itself runs in HIP."""


class ResidualDenseBlock:
    """Marker (srgan_train.py:275-360)."""


class ResInResDenseBlock:
    """Marker for GeneratorModel(resblock_class=...)  (srgan_train.py:364-404)."""


class Parameter:
    """View of one named tensor of a model (chainer.Parameter look-alike)."""

    def __init__(self, model, name, shape, kind):
        self._model, self.name, self.shape, self.kind = model, name, tuple(shape), kind
        self.size = int(np.prod(shape)) if len(shape) else 1

    @property
    def array(self):
        out = np.empty(self.size, dtype=np.float32)
        _lib.check(_lib.lib().dbm_model_get_tensor(self._model._h, self.name.encode(), _hp(out), out.size),
                   self._model.ctx.handle)
        return out.reshape(self.shape)

    @array.setter
    def array(self, value):
        value = np.asarray(value)
        if tuple(value.shape) != self.shape and not (value.size == 1 and self.size == 1):
            raise ValueError(f"shape mismatch for {self.name}: {tuple(value.shape)} vs {self.shape}")
        v = _f32(value).reshape(-1)
        _lib.check(_lib.lib().dbm_model_set_tensor(self._model._h, self.name.encode(), _hp(v), v.size),
                   self._model.ctx.handle)

    data = array

    @property
    def grad(self):
        out = np.empty(self.size, dtype=np.float32)
        _lib.check(_lib.lib().dbm_model_get_grad(self._model._h, self.name.encode(), _hp(out), out.size),
                   self._model.ctx.handle)
        return out.reshape(self.shape)

    def __getitem__(self, idx):  # the doctests index params directly (srgan_train.py:1113, 1203)
        return Variable(self.array[idx])


def _chainer_order(names):
    """Link.params() order: own params sorted, then children sorted, recursively (chainer/link.py)."""
    def key(n):
        parts = n.split("/")
        return [(0 if i == len(parts) - 1 else 1, p) for i, p in enumerate(parts)]
    return sorted(names, key=key)


class _Link:
    xp = np  # callers only use it to build host arrays (srgan_train.py:1368, 1446)

    def __init__(self, ctx=None):
        self.ctx = ctx or _lib.default_context()
        self._h = C.c_void_p()
        self._optimizer = None

    def _index(self):
        l = _lib.lib()
        n = C.c_int()
        _lib.check(l.dbm_model_num_tensors(self._h, C.byref(n)), self.ctx.handle)
        self._tensors = {}
        for i in range(n.value):
            key = C.c_char_p()
            nd = C.c_int()
            kind = C.c_int()
            shape = (C.c_int64 * 4)()
            _lib.check(l.dbm_model_tensor_info(self._h, i, C.byref(key), C.byref(nd), shape, C.byref(kind)),
                       self.ctx.handle)
            name = key.value.decode()
            self._tensors[name] = Parameter(self, name, [shape[k] for k in range(nd.value)], kind.value)

    def _init_params(self):
        """HeNormal(scale=0.1, fan_in) weights from NumPy's global RNG (as chainer.initializers does), zero
        biases; BatchNorm gamma/avg_var = 1 are set by the library (srgan_train.py:220, 289, 461, 613)."""
        for name, p in self._tensors.items():
            if name.endswith("/W"):
                fan_in = int(np.prod(p.shape[1:]))
                p.array = np.random.normal(0.0, 0.1 * np.sqrt(2.0 / fan_in), size=p.shape)

    def namedparams(self):
        names = _chainer_order([n for n, p in self._tensors.items() if p.kind == 0])
        return [("/" + n, self._tensors[n]) for n in names]

    def params(self):
        return (p for _, p in self.namedparams())

    def count_params(self):
        n = C.c_int64()
        _lib.check(_lib.lib().dbm_model_count_params(self._h, C.byref(n)), self.ctx.handle)
        return int(n.value)

    def cleargrads(self):
        _lib.check(_lib.lib().dbm_model_cleargrads(self._h), self.ctx.handle)

    def to_gpu(self, device=None):
        return self  # parameters always live in HBM

    def to_cpu(self):
        raise DbmError("deepbedmap_amd has no CPU path")

    def grad_arena(self):
        p, n = C.c_void_p(), C.c_size_t()
        _lib.check(_lib.lib().dbm_model_grad_arena(self._h, C.byref(p), C.byref(n)), self.ctx.handle)
        return DeviceArray((n.value,), self.ctx, ptr=p.value, owner=self)

    def param_arena(self):
        p, n = C.c_void_p(), C.c_size_t()
        _lib.check(_lib.lib().dbm_model_param_arena(self._h, C.byref(p), C.byref(n)), self.ctx.handle)
        return DeviceArray((n.value,), self.ctx, ptr=p.value, owner=self)

    def mark_params_changed(self):
        _lib.check(_lib.lib().dbm_model_params_changed(self._h), self.ctx.handle)

    def serialize_dict(self):
        out = {}
        for name, p in self._tensors.items():
            a = p.array
            if name.endswith("/N"):
                a = np.asarray(int(a.reshape(-1)[0]) if a.size else 0)
            out[name] = a
        return out

    def __del__(self):
        try:
            if self._h:
                _lib.lib().dbm_model_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


class GeneratorModel(_Link):
    """srgan_train.py:421-576.

    >>> generator_model = GeneratorModel()
    >>> y_pred = generator_model.forward(x=..., w1=..., w2=..., w3=...)   # (1,1,11,11) ... -> (1,1,36,36)
    >>> generator_model.count_params()
    8907749
    """

    def __init__(self, inblock_class=DeepbedmapInputBlock, resblock_class=ResInResDenseBlock,
                 num_residual_blocks: int = 12, residual_scaling: float = 0.1, out_channels: int = 1, ctx=None,
                 initialize=True):
        super().__init__(ctx)
        # the two block classes are compiled into the HIP kernels: anything but the reference's own is refused loudly
        if inblock_class is not DeepbedmapInputBlock:
            raise ValueError("GeneratorModel: only inblock_class=DeepbedmapInputBlock is implemented (srgan_train.py:201-266)")
        if resblock_class is not ResInResDenseBlock:
            raise ValueError("GeneratorModel: only resblock_class=ResInResDenseBlock is implemented (srgan_train.py:364-404)")
        self.num_residual_blocks = num_residual_blocks
        self.residual_scaling = residual_scaling
        self.out_channels = int(out_channels)  # > 1: forward only (the reference's own training step fails with it)
        _lib.check(_lib.lib().dbm_gen_create(self.ctx.handle, int(num_residual_blocks), float(residual_scaling),
                                             int(out_channels), C.byref(self._h)), self.ctx.handle)
        self._index()
        if initialize:
            self._init_params()
        self._graph_shape = None

    def forward(self, x, w1, w2, w3):
        keep = bool(global_config.enable_backprop)
        flags = KEEP_GRAPH if keep else 0
        if global_config.dtype == "bfloat16":
            if keep:
                raise ValueError("dtype bfloat16 is an inference mode: use using_config('enable_backprop', False)")
            flags |= BF16
        elif global_config.dtype != "float32":
            raise ValueError(f"unknown dtype {global_config.dtype!r}")
        device = _is_device(x)
        n, _, h, w = x.shape
        exp = {"w1": (n, 1, 10 * h, 10 * w), "w2": (n, 2, 2 * h, 2 * w), "w3": (n, 1, h, w)}
        for name, arr in (("w1", w1), ("w2", w2), ("w3", w3)):
            if tuple(arr.shape) != exp[name]:
                raise ValueError(f"Invalid shape for {name}: expected {exp[name]}, got {tuple(arr.shape)}")
        oshape = (n, self.out_channels, 4 * (h - 2), 4 * (w - 2))
        l = _lib.lib()
        if keep and self.out_channels != 1:
            raise ValueError("out_channels > 1 is forward-only: use using_config('enable_backprop', False) (the reference's "
                             "training step fails with it too: mean_absolute_error against the one-channel x_topo)")
        if device:
            y = DeviceArray(oshape, self.ctx)
            self._held_inputs = (x, w1, w2, w3)  # the input-block weight gradient reads them in backward
            _lib.check(l.dbm_gen_forward(self._h, n, h, w, _dev_ptr(x), _dev_ptr(w1), _dev_ptr(w2), _dev_ptr(w3),
                                         y.ptr, flags | DEVICE_PTRS), self.ctx.handle)
            out = y
        else:
            xs = [_f32(a) for a in (x, w1, w2, w3)]
            out = np.empty(oshape, dtype=np.float32)
            _reissue_once(lambda: l.dbm_gen_forward(self._h, n, h, w, _hp(xs[0]), _hp(xs[1]), _hp(xs[2]), _hp(xs[3]), _hp(out), flags),
                          self.ctx)
        self._graph_shape = oshape if keep else None
        v = Variable(out, creator=None)
        v._gen = self if keep else None
        return v

    __call__ = forward

    def backward(self, gy):
        """d loss / d output of the last retained forward -> parameter gradients (accumulated)."""
        _apply_config(self.ctx)
        if _is_device(gy):
            _lib.check(_lib.lib().dbm_gen_backward(self._h, _dev_ptr(gy), DEVICE_PTRS), self.ctx.handle)
        else:
            g = _f32(gy)
            _lib.check(_lib.lib().dbm_gen_backward(self._h, _hp(g), 0), self.ctx.handle)


class DiscriminatorModel(_Link):
    """srgan_train.py:591-699.

    >>> discriminator_model = DiscriminatorModel()
    >>> discriminator_model.forward(x=np.random.rand(2, 1, 36, 36).astype("float32")).shape
    (2, 1)
    >>> discriminator_model.count_params()
    10370761
    """

    def __init__(self, ctx=None, initialize=True):
        super().__init__(ctx)
        _lib.check(_lib.lib().dbm_disc_create(self.ctx.handle, C.byref(self._h)), self.ctx.handle)
        self._index()
        if initialize:
            self._init_params()
        self._next_slot = 0

    def forward(self, x, slot=None):
        if isinstance(x, Variable):
            x = x.array
        train = bool(global_config.train)
        keep = bool(global_config.enable_backprop) and train
        flags = (BN_TRAIN if train else 0) | (KEEP_GRAPH if keep else 0)
        if slot is None:  # the D-step keeps two graphs alive (real, fake): alternate
            slot = self._next_slot
            self._next_slot ^= 1
        n, c, h, w = x.shape
        if c != 1:
            raise ValueError("DiscriminatorModel expects one input channel")
        l = _lib.lib()
        if _is_device(x):
            out = DeviceArray((n, 1), self.ctx)
            _lib.check(l.dbm_disc_forward(self._h, n, h, w, _dev_ptr(x), out.ptr, flags | DEVICE_PTRS, slot),
                       self.ctx.handle)
        else:
            xs = _f32(x)
            out = np.empty((n, 1), dtype=np.float32)
            if train:   # (a training-mode pass updates the running averages: not re-issued blindly -- status 7 goes to the caller)
                _lib.check(l.dbm_disc_forward(self._h, n, h, w, _hp(xs), _hp(out), flags, slot), self.ctx.handle)
            else:
                _reissue_once(lambda: l.dbm_disc_forward(self._h, n, h, w, _hp(xs), _hp(out), flags, slot), self.ctx)
        v = Variable(out)
        v._disc = (self, slot) if keep else None
        return v

    __call__ = forward

    def backward(self, slot, glogits):
        _apply_config(self.ctx)
        if _is_device(glogits):
            _lib.check(_lib.lib().dbm_disc_backward(self._h, slot, _dev_ptr(glogits), DEVICE_PTRS), self.ctx.handle)
        else:
            g = _f32(glogits)
            _lib.check(_lib.lib().dbm_disc_backward(self._h, slot, _hp(g), 0), self.ctx.handle)


def _reissue_once(call, ctx):
    """A host-synchronising forward that reports status 7 (a persistent kernel gave up: include/dbm.h) has produced void
    results and changed nothing: it is issued again, once -- the library has switched to the layer-by-layer kernels."""
    rc = call()
    if rc == 7:
        import warnings

        warnings.warn("libdbm: a persistent kernel timed out during a forward pass; repeating it on the layer-by-layer kernels",
                      RuntimeWarning, stacklevel=3)
        rc = call()
    _lib.check(rc, ctx.handle)


# --------------------------------------------------------------------------------------
# optimizer + serializers
# --------------------------------------------------------------------------------------
class Adam:
    """chainer.optimizers.Adam(alpha, eps).setup(link)  (srgan_train.py:1043-1048)."""

    def __init__(self, alpha=0.001, beta1=0.9, beta2=0.999, eps=1e-8):
        self.alpha, self.beta1, self.beta2, self.eps = alpha, beta1, beta2, eps
        self.target = None
        self.t = 0

    def setup(self, link):
        self.target = link
        link._optimizer = self
        _lib.check(_lib.lib().dbm_adam_setup(link._h, self.alpha, self.beta1, self.beta2, self.eps), link.ctx.handle)
        return self

    def update(self, grad_scale=1.0):
        """optimizer.update() (srgan_train.py:1164, 1257).  DbmError code 9: a persistent kernel timed out during the pass that
        produced the gradients -- nothing was applied (and `t` is unchanged): repeat forward + backward, then update."""
        _lib.check(_lib.lib().dbm_adam_update(self.target._h, float(grad_scale)), self.target.ctx.handle)
        self.t += 1


class optimizers:  # namespace parity with chainer.optimizers
    Adam = Adam


def save_npz(file, obj, compression=True):
    """chainer.serializers.save_npz: flat '/'-joined keys, np.savez_compressed (srgan_train.py:1355-1361)."""
    d = obj.serialize_dict()
    (np.savez_compressed if compression else np.savez)(file, **d)


def infer_num_residual_blocks(file):
    """Number of ResInResDenseBlocks in a generator .npz: the reference does NOT serialise `num_residual_blocks`
    (a plain attribute, srgan_train.py:459-460; deepbedmap.py:397-405 recovers it from Comet's experiment parameters) --
    but the keys `residual_network/{i}/...` of chainer's Sequential say it."""
    with np.load(file) as f:
        idx = {int(k.split("/")[1]) for k in f.files if k.startswith("residual_network/")}
    if not idx or idx != set(range(len(idx))):
        raise ValueError(f"{file}: no contiguous residual_network/0..n-1 keys (not a GeneratorModel .npz?)")
    return len(idx)


def load_trained_model(model_weights_path, num_residual_blocks=None, residual_scaling: float = 0.1, ctx=None):
    """The construct + load part of deepbedmap.py:381-410 (`load_trained_model`, minus the Comet.ML download):
    GeneratorModel(num_residual_blocks=..., residual_scaling=...) + chainer.serializers.load_npz.  num_residual_blocks=None
    reads the block count from the file's keys; residual_scaling is not recoverable from the weights (pass the value the
    model was trained with; the reference's default is 0.1)."""
    n = infer_num_residual_blocks(model_weights_path) if num_residual_blocks is None else int(num_residual_blocks)
    model = GeneratorModel(num_residual_blocks=n, residual_scaling=float(residual_scaling), ctx=ctx, initialize=False)
    load_npz(model_weights_path, model)
    return model


class BlockCountMismatch(KeyError, ValueError):
    """load_npz of a generator file into a model built with another num_residual_blocks.  Chainer raises a bare KeyError
    for the first missing key (strict mode); this subclass says what is actually wrong."""

    def __str__(self):
        return str(self.args[0]) if self.args else ""


def load_npz(file, obj, strict=True):
    """chainer.serializers.load_npz (deepbedmap.py:408, srgan_train.py:1566-1574)."""
    with np.load(file) as f:
        keys = set(f.files)
        if isinstance(obj, GeneratorModel):  # a clear message instead of a KeyError deep in the key list
            idx = {int(k.split("/")[1]) for k in keys if k.startswith("residual_network/")}
            if idx and len(idx) != int(obj.num_residual_blocks) and strict:
                raise BlockCountMismatch(f"{file} holds {len(idx)} residual blocks, the model was built with "
                                         f"num_residual_blocks={obj.num_residual_blocks} (load_trained_model() reads the count "
                                         "from the file)")
        for name, p in obj._tensors.items():
            if name not in keys:
                if strict:
                    raise KeyError(f"{name} is not in the npz file")
                continue
            a = f[name]
            # chainer's deserializer copies into the existing array and raises on a shape mismatch; only the scalar
            # persistent `N` of BatchNormalization is stored 0-d
            if tuple(a.shape) != tuple(p.shape) and not (a.ndim == 0 and p.size == 1):
                raise ValueError(f"shape mismatch for {name}: file {tuple(a.shape)} vs model {tuple(p.shape)}")
            p.array = a
    return obj


class serializers:  # namespace parity with chainer.serializers
    save_npz = staticmethod(save_npz)
    load_npz = staticmethod(load_npz)


# --------------------------------------------------------------------------------------
# losses / metrics
# --------------------------------------------------------------------------------------
def _targets(t, n, name):
    """int32 target array of F.sigmoid_cross_entropy (srgan_train.py:995-1004): any array of 0 / 1 / -1 (ignored) with one
    entry per logit, as Chainer's type check demands."""
    t = np.ascontiguousarray(np.asarray(t).reshape(-1), dtype=np.int32)
    if t.size != n:
        raise ValueError(f"{name}: {t.size} targets for {n} logits")
    if not np.isin(t, (-1, 0, 1)).all():
        raise ValueError(f"{name}: targets must be 0, 1 or -1 (ignored)")
    return t


def _arr(v):
    return v.array if isinstance(v, Variable) else v


def calculate_discriminator_loss(real_labels_pred, fake_labels_pred, real_minus_fake_target, fake_minus_real_target):
    """srgan_train.py:960-1009.  Returns a Variable; `.backward()` back-propagates into the discriminator(s)
    whose retained forwards produced the two logits arrays."""
    real, fake = _arr(real_labels_pred), _arr(fake_labels_pred)
    ctx = _lib.default_context()
    l = _lib.lib()
    n = int(np.prod(real.shape))
    t_rf = _targets(real_minus_fake_target, n, "real_minus_fake_target")
    t_fr = _targets(fake_minus_real_target, n, "fake_minus_real_target")
    out = np.empty(2, dtype=np.float32)
    r, f = _f32(np.asarray(real)).reshape(-1), _f32(np.asarray(fake)).reshape(-1)
    gr, gf = np.empty(n, np.float32), np.empty(n, np.float32)
    _lib.check(l.dbm_discriminator_loss_t(ctx.handle, _hp(r), _hp(f), n, _hp(t_rf), _hp(t_fr), _hp(out), _hp(gr), _hp(gf), 0),
               ctx.handle)
    v = Variable(np.float32(out[0]))
    v.accuracy = float(out[1])
    dr = getattr(real_labels_pred, "_disc", None)
    df = getattr(fake_labels_pred, "_disc", None)

    def creator():
        if dr is None and df is None:
            raise DbmError("no retained discriminator graph to back-propagate into")
        if dr is not None:
            dr[0].backward(dr[1], gr)
        if df is not None:
            df[0].backward(df[1], gf)

    v._creator = creator
    return v


def _gen_loss_call(y_pred, y_true, x_full, real_labels, fake_labels, t_rf, t_fr, weights, want_grad):
    ctx = _lib.default_context()
    l = _lib.lib()
    yp, yt = _f32(np.asarray(_arr(y_pred))), _f32(np.asarray(y_true))
    if yp.shape != yt.shape:
        raise ValueError("Input images must have the same dimensions.")  # srgan_train.py:950-951
    n, c, h, w = yp.shape
    if c != 1:  # chainer: F.mean_absolute_error(pooled (N, c, h/4, w/4), x_topo (N, 1, h/4, w/4)) fails its shape check
        raise ValueError(f"calculate_generator_loss: y_pred has {c} channels, x_topo has 1 (mean_absolute_error needs equal shapes)")
    xf = _f32(x_full)
    fl = _f32(np.asarray(fake_labels)).reshape(-1)
    rl = None if real_labels is None else _f32(np.asarray(real_labels)).reshape(-1)
    out = np.empty(3, dtype=np.float32)
    gy = np.empty_like(yp) if want_grad else None
    wts = (C.c_float * 4)(*[float(v) for v in weights])
    win = {"gaussian": 0, "uniform": 1}[global_config.ssim_window]
    _lib.check(l.dbm_generator_loss_t(ctx.handle, _hp(yp), _hp(yt), _hp(xf), None if rl is None else _hp(rl), _hp(fl), n,
                                      h, w, wts, _hp(t_rf), _hp(t_fr), win, _hp(out), None if gy is None else _hp(gy), 0),
               ctx.handle)
    return out, gy


def calculate_generator_loss(y_pred, y_true, fake_labels, real_labels, fake_minus_real_target,
                             real_minus_fake_target, x_topo, content_loss_weighting: float = 1e-2,
                             adversarial_loss_weighting: float = 2e-2, topographic_loss_weighting: float = 2e-3,
                             structural_loss_weighting: float = 5.25e-0):
    """srgan_train.py:841-902.  `.backward()` on the result runs the generator backward when y_pred came from a
    retained GeneratorModel.forward."""
    n_logits = int(np.prod(np.asarray(_arr(fake_labels)).shape))
    t_fr = _targets(fake_minus_real_target, n_logits, "fake_minus_real_target")
    t_rf = _targets(real_minus_fake_target, n_logits, "real_minus_fake_target")
    xt = _f32(np.asarray(x_topo))
    x_full = np.pad(xt, ((0, 0), (0, 0), (1, 1), (1, 1)))  # the kernel reads x[:, :, 1:-1, 1:-1]
    weights = (content_loss_weighting, adversarial_loss_weighting, topographic_loss_weighting,
               structural_loss_weighting)
    gen = getattr(y_pred, "_gen", None)
    out, gy = _gen_loss_call(y_pred, y_true, x_full, real_labels, fake_labels, t_rf, t_fr, weights, gen is not None)
    v = Variable(np.float32(out[0]))
    if gen is not None:
        v._creator = lambda: gen.backward(gy)
    return v


def psnr(y_pred, y_true, data_range=2 ** 32):
    """srgan_train.py:906-928 (batchwise; 20*log10(data_range / sqrt(mse)))."""
    yp, yt = _f32(np.asarray(_arr(y_pred))), _f32(np.asarray(y_true))
    ctx = _lib.default_context()
    out = np.empty(1, dtype=np.float32)
    _lib.check(_lib.lib().dbm_psnr(ctx.handle, _hp(yp), _hp(yt), yp.size, float(data_range), _hp(out), 0), ctx.handle)
    return float(out[0])


def ssim_loss_func(y_pred, y_true, window_size: int = 9, stride: int = 1):
    """srgan_train.py:932-956: mean SSIM over the valid window_size x window_size windows taken every `stride` pixels
    (ssim.functions.ssim_loss); ValueError on shape mismatch.  Any window_size in [1, 64] and stride >= 1."""
    yp = np.asarray(_arr(y_pred))
    yt = np.asarray(y_true)
    if not yp.shape == yt.shape:
        raise ValueError("Input images must have the same dimensions.")
    window_size, stride = int(window_size), int(stride)
    yp, yt = _f32(yp), _f32(yt)
    n, c, h, w = yp.shape
    if not (1 <= window_size <= 64) or stride < 1 or h < window_size or w < window_size:
        raise ValueError(f"ssim_loss_func: window_size {window_size} / stride {stride} do not fit {h} x {w} images")
    ctx = _lib.default_context()
    out = np.empty(1, dtype=np.float32)
    win = {"gaussian": 0, "uniform": 1}[global_config.ssim_window]
    _lib.check(_lib.lib().dbm_ssim_ex(ctx.handle, _hp(yp), _hp(yt), n * c, h, w, window_size, stride, win, _hp(out), 0),
               ctx.handle)
    return Variable(np.float32(out[0]))
