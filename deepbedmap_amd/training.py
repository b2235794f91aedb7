"""compile_srgan_model / train_eval_discriminator / train_eval_generator / trainer /
save_model_weights_and_architecture with the reference's signatures (srgan_train.py:1014-1383).

The per-minibatch work is ONE C call per model step (dbm_discriminator_step / dbm_generator_step:
forward, losses, backward, all enqueued on the HIP stream) followed by the optional RCCL gradient
all-reduce and the fused Adam kernel.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from .srgan import (Adam, DeviceArray, DiscriminatorModel, GeneratorModel, global_config, save_npz, to_device,
                    _apply_config, _dev_ptr, _is_device)

LOSS_WEIGHTS = (1e-2, 2e-2, 2e-3, 5.25e-0)  # calculate_generator_loss defaults (srgan_train.py:849-852)


def compile_srgan_model(num_residual_blocks: int = 12, residual_scaling: float = 0.1, learning_rate: float = 1.6e-4):
    """srgan_train.py:1014-1055.  Returns (generator_model, generator_optimizer, discriminator_model,
    discriminator_optimizer)."""
    generator_model = GeneratorModel(num_residual_blocks=num_residual_blocks, residual_scaling=residual_scaling)
    discriminator_model = DiscriminatorModel()
    generator_optimizer = Adam(alpha=learning_rate, eps=1e-8).setup(link=generator_model)
    discriminator_optimizer = Adam(alpha=learning_rate, eps=1e-8).setup(link=discriminator_model)
    return generator_model, generator_optimizer, discriminator_model, discriminator_optimizer


_KEYS = ("X", "W1", "W2", "W3", "Y")
_metrics = {}
_prefetch_tokens = {}  # id(generator) -> identity + content version of the arrays its prefetched forward was computed from


def _content_token(dev):
    """(object identity, content version) per array: DeviceArray counts its own writes, torch tensors theirs."""
    tok = []
    for k in _KEYS:
        a = dev[k]
        tok.append((id(a), _dev_ptr(a), getattr(a, "_gen", None), getattr(a, "_version", None)))
    return tuple(tok)


def _metrics_buffer(ctx):
    if id(ctx) not in _metrics:
        _metrics[id(ctx)] = DeviceArray((8,), ctx)
    return _metrics[id(ctx)]


def device_batch(input_arrays, ctx=None):
    """Upload a dict of NumPy arrays once; device arrays pass through (chainer: arrays already `to_gpu`'d,
    srgan_train.py:110-116)."""
    return {k: (v if _is_device(v) else to_device(v, ctx)) for k, v in input_arrays.items()}


def _check_batch(arrs):
    n, c, h, w = arrs["X"].shape
    exp = {"X": (n, 1, h, w), "W1": (n, 1, 10 * h, 10 * w), "W2": (n, 2, 2 * h, 2 * w), "W3": (n, 1, h, w),
           "Y": (n, 1, 4 * (h - 2), 4 * (w - 2))}
    for k in _KEYS:
        if tuple(arrs[k].shape) != exp[k]:
            raise ValueError(f"Invalid shape for {k}: expected {exp[k]}, got {tuple(arrs[k].shape)}")
    return n, h, w


def _repeat_after_timeout(fn):
    """libdbm status 7: a persistent trunk kernel gave up waiting for a neighbouring workgroup (another process starving
    the GPU).  The library reports it ONLY at the entry of a step call (or from Context.check_timeout()), before anything
    of that call has been enqueued: the device is drained, the optimizer launches and BatchNorm running-average writes
    queued since the event were no-ops (no parameter, moment or statistic absorbed an invalid pass; their count is
    reported), the layer-by-layer trunk kernels take over for a while.  The call is therefore simply issued again -- it has
    not run yet.  What is lost are the updates of the minibatches queued between the event and this call: a warning says how
    many, and a MetricsLog passed as `log=` / `metrics=` marks their rows invalid (NaN).  Status 8 (the same in a
    data-parallel run: the replicas have diverged) is not retried."""
    import functools
    import warnings

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        try:
            return fn(*args, **kwargs)
        except _lib.DbmError as e:
            if e.code != 7:
                raise
            g_model = args[1] if len(args) > 1 else kwargs.get("g_model")
            ctx = getattr(g_model, "ctx", None) or _lib.default_context()
            _, nd, ng, _ = ctx.timeout_info()
            _dropped[id(ctx)] = _dropped.get(id(ctx), 0) + max(nd, ng)
            warnings.warn(f"libdbm: a persistent kernel timed out; {nd} discriminator / {ng} generator updates queued since "
                          "were skipped (their minibatches are lost); continuing on the layer-by-layer trunk kernels",
                          RuntimeWarning, stacklevel=2)
            return fn(*args, **kwargs)

    return wrapper


_dropped = {}  # id(ctx) -> optimizer updates dropped by timeouts since the last `pop_dropped_updates`


def pop_dropped_updates(ctx):
    """Number of minibatch updates lost to persistent-kernel timeouts on `ctx` since the last call (0 in a healthy run)."""
    return _dropped.pop(id(ctx), 0)


@_repeat_after_timeout
def train_eval_discriminator(input_arrays, g_model, d_model, d_optimizer=None, train: bool = True, comm=None,
                             sync: bool = True, share_generator_forward: bool = False,
                             prefetch_generator_forward: bool = False, metrics=None):
    """srgan_train.py:1084-1166.  Returns (d_loss, d_accu) as floats (like the reference's float(...) D2H syncs);
    sync=False returns the device metrics buffer instead and keeps the stream running.
    share_generator_forward=True (opt-in, not the reference's behaviour) retains this call's generator forward so
    that train_eval_generator(..., share_generator_forward=True) on the same device arrays reuses it: the generator
    and its inputs do not change in between, so the numbers are bitwise the same and one forward pass is saved.
    prefetch_generator_forward=True (what `trainer` does): the caller promises that train_eval_generator on the SAME
    device arrays follows; that call's generator forward (its own workspace, nothing shared or skipped) is enqueued
    now, on separate HIP streams, so that it runs underneath this step's discriminator passes.  If the promise is
    broken -- other array objects, a DeviceArray refilled with .set() or freed, a torch tensor written in place (its
    _version moves), new parameters -- train_eval_generator does not ask for the prefetched pass and the library
    discards it (dbm_generator_step, bit 2)."""
    global_config.train = train  # srgan_train.py:1125
    if train is True:
        assert d_optimizer is not None  # Optimizer required for neural network training
    dev = device_batch(input_arrays, g_model.ctx)
    n, h, w = _check_batch(dev)
    m = metrics if metrics is not None else _metrics_buffer(g_model.ctx)  # >= 8 floats on the device
    _apply_config(g_model.ctx)
    _lib.check(_lib.lib().dbm_discriminator_step(g_model._h, d_model._h, n, h, w, *[_dev_ptr(dev[k]) for k in _KEYS],
                                                 int(bool(train)) | (2 if share_generator_forward else 0) |
                                                 (4 if prefetch_generator_forward else 0) |
                                                 (comm.step_flags(g_model.ctx) if hasattr(comm, "step_flags") else
                                                  (8 if comm is not None else 0)), m.ptr),
               g_model.ctx.handle)
    # the G-step may consume the prefetched forward only for these very arrays with this very content
    _prefetch_tokens[id(g_model)] = _content_token(dev) if (prefetch_generator_forward and train) else None
    if train is True:
        scale = comm.allreduce_grads(d_model) if comm is not None else 1.0
        d_optimizer.update(grad_scale=scale)
    if not sync:
        return m
    out = m.get()
    g_model.ctx.check_timeout()  # (synchronous form: a timeout during THIS call is reported by this call, which is then repeated)
    return float(out[0]), float(out[1])


@_repeat_after_timeout
def train_eval_generator(input_arrays, g_model, d_model, g_optimizer=None, train: bool = True, comm=None,
                         sync: bool = True, share_generator_forward: bool = False, metrics=None):
    """srgan_train.py:1170-1263.  Returns (g_loss, g_psnr, g_ssim)."""
    global_config.train = train  # srgan_train.py:1216
    if train is True:
        assert g_optimizer is not None  # Optimizer required for neural network training
    dev = device_batch(input_arrays, g_model.ctx)
    n, h, w = _check_batch(dev)
    m = metrics if metrics is not None else _metrics_buffer(g_model.ctx)
    wts = (C.c_float * 4)(*LOSS_WEIGHTS)
    win = {"gaussian": 0, "uniform": 1}[global_config.ssim_window]
    _apply_config(g_model.ctx)
    use_prefetched = _prefetch_tokens.pop(id(g_model), None) == _content_token(dev)
    _lib.check(_lib.lib().dbm_generator_step(g_model._h, d_model._h, n, h, w, *[_dev_ptr(dev[k]) for k in _KEYS], wts,
                                             win, int(bool(train)) | (2 if share_generator_forward else 0) |
                                             (4 if use_prefetched else 0), m.ptr),
               g_model.ctx.handle)
    if train is True:
        scale = comm.allreduce_grads(g_model) if comm is not None else 1.0
        g_optimizer.update(grad_scale=scale)
    if not sync:
        return m
    out = m.get()
    g_model.ctx.check_timeout()
    return float(out[2]), float(out[3]), float(out[4])


# --------------------------------------------------------------------------------------
# chainer.iterators.SerialIterator / chainer.dataset.concat_examples stand-ins (srgan_train.py:132-166, 1286-1288)
# --------------------------------------------------------------------------------------
class SerialIterator:
    """Batches of indices over a dict-of-arrays dataset with Chainer's SerialIterator semantics
    (repeat=True: batches are always full; the tail of an epoch is completed from the next, reshuffled, order)."""

    def __init__(self, dataset, batch_size, repeat=True, shuffle=True, seed=None):
        self.dataset = dataset
        self.n = len(next(iter(dataset.values())))
        self.batch_size, self.repeat, self.shuffle = batch_size, repeat, shuffle
        self._rng = np.random.RandomState(seed)
        self.reset()

    def reset(self):
        self.epoch = 0
        self.is_new_epoch = False
        self.pos = 0
        self.order = self._rng.permutation(self.n) if self.shuffle else np.arange(self.n)

    def next(self):
        if not self.repeat and self.epoch > 0:
            raise StopIteration
        i, e = self.pos, self.pos + self.batch_size
        idx = self.order[i:e]
        if e >= self.n:
            if self.repeat:
                rest = e - self.n
                self.order = self._rng.permutation(self.n) if self.shuffle else np.arange(self.n)
                if rest > 0:
                    idx = np.concatenate([idx, self.order[:rest]])
                self.pos = rest
            else:
                self.pos = 0
            self.epoch += 1
            self.is_new_epoch = True
        else:
            self.is_new_epoch = False
            self.pos = e
        return idx

    __next__ = next


def concat_examples(dataset, batch):
    """Gather the rows `batch` (index array) of every array in the dataset dict (chainer.dataset.concat_examples,
    srgan_train.py:1286-1288).  Arrays that live on the device (the reference moves the whole dataset `to_gpu`,
    srgan_train.py:107-121) are gathered there -- one kernel per array, no host round trip -- and stay there."""
    out = {}
    idx = None
    for k, v in dataset.items():
        if _is_device(v):
            if idx is None:
                idx = np.ascontiguousarray(batch, dtype=np.int32)
                if idx.size and (idx.min() < 0 or idx.max() >= len(v)):
                    raise IndexError("concat_examples: index out of range")
            dst = DeviceArray((len(idx),) + tuple(v.shape[1:]), v.ctx)
            row_bytes = 4 * int(np.prod(v.shape[1:]))
            _lib.check(_lib.lib().dbm_gather_rows(v.ctx.handle, C.c_void_p(dst.ptr), C.c_void_p(v.ptr),
                                                  idx.ctypes.data_as(C.POINTER(C.c_int)), len(idx), row_bytes), v.ctx.handle)
            out[k] = dst
        else:
            out[k] = np.ascontiguousarray(v[batch])
    return out


def dataset_to_device(dataset, ctx=None):
    """`chainer.backend.cuda.to_gpu` over the five arrays of the DictDataset (srgan_train.py:107-121)."""
    return {k: (v if _is_device(v) else to_device(v, ctx)) for k, v in dataset.items()}


def split_dataset_random(dataset, first_size: int, seed=None):
    """chainer.datasets.split_dataset_random: one seeded permutation, the first `first_size` examples and the rest."""
    n = len(next(iter(dataset.values())))
    if not 0 <= first_size <= n:
        raise ValueError("first_size must be in [0, len(dataset)]")
    order = np.random.RandomState(seed).permutation(n)
    return concat_examples(dataset, order[:first_size]), concat_examples(dataset, order[first_size:])


def get_train_dev_iterators(dataset, first_size: int, batch_size: int = 128, seed: int = 42):
    """srgan_train.py:132-166: seeded random train / dev split, a shuffling and a sequential repeating iterator."""
    train_set, dev_set = split_dataset_random(dataset, first_size=first_size, seed=seed)
    train_iter = SerialIterator(dataset=train_set, batch_size=batch_size, repeat=True, shuffle=True)
    dev_iter = SerialIterator(dataset=dev_set, batch_size=batch_size, repeat=True, shuffle=False)
    n_train, n_dev = train_iter.n, dev_iter.n
    print(f"Training dataset: {n_train} tiles,", f"Development dataset: {n_dev} tiles")
    return train_iter, n_train, dev_iter, n_dev


class MetricsLog:
    """Device-resident log of per-minibatch metrics: row r = [d_loss, d_accu, g_loss, g_psnr, g_ssim, -, -, -] of the
    r-th minibatch.  The fused steps write their metrics straight into a row; nothing is copied to the host until
    `fetch()` -- `trainer` only needs the numbers at the end of the epoch (np.mean per column, srgan_train.py:1617-1619),
    so the GPU never waits for a host round trip between minibatches."""

    def __init__(self, ctx, rows=1024):
        self.ctx, self.rows, self.n = ctx, rows, 0
        self.buf = DeviceArray((rows, 8), ctx)

    def next_row(self):
        if self.n == self.rows:  # grow: keep what is there
            old = self.fetch()
            self.rows *= 2
            self.buf = DeviceArray((self.rows, 8), self.ctx)
            pad = np.zeros((self.rows, 8), np.float32)
            pad[:self.n] = old
            self.buf.set(pad)
        row = DeviceArray((8,), self.ctx, ptr=self.buf.ptr + 32 * self.n, owner=self.buf)
        self.n += 1
        return row

    invalid = ()  # row indices whose minibatch was computed by / after a timed-out persistent kernel (their update was skipped)

    def invalidate_last(self, k, keep_last=0):
        """Marks the k rows before the last `keep_last` ones invalid: fetch() returns them as NaN."""
        hi = self.n - keep_last
        self.invalid = tuple(sorted(set(self.invalid) | set(range(max(0, hi - k), hi))))

    def fetch(self):
        """(n, 8) float32 array of the rows written so far (synchronises the stream); invalidated rows are NaN."""
        rows = self.buf.get()[:self.n].copy()
        for r in self.invalid:
            rows[r] = np.nan
        return rows


@_repeat_after_timeout
def train_iteration(train_arrays, g_model, g_optimizer, d_model, d_optimizer, metrics=None, share_generator_forward=False):
    """dbm_train_iteration: D-step, discriminator update, G-step, generator update of one minibatch of DEVICE arrays as one
    library call.  Returns the device metrics buffer [d_loss, d_accu, g_loss, g_psnr, g_ssim, ...] (no host
    synchronisation).  On a context with a communicator (DataParallel.attach: "rccl", or "gloo" on a GPU) the call is one
    data-parallel iteration: both gradient arenas are summed over ranks inside it and both updates take 1 / world.
    share_generator_forward=True (opt-in, DBM_ONE_GEN_FORWARD): the generator runs once, its retained forward also supplies the
    D-step's fakes (both forwards of srgan_train.py:1131 / :1222 see the same weights and inputs) -- bit for bit the two step
    calls with share_generator_forward=True, the default iteration up to fp32 rounding."""
    global_config.train = True
    assert d_optimizer is not None and g_optimizer is not None  # Optimizer required for neural network training
    n, h, w = _check_batch(train_arrays)
    m = metrics if metrics is not None else _metrics_buffer(g_model.ctx)
    wts = (C.c_float * 4)(*LOSS_WEIGHTS)
    win = {"gaussian": 0, "uniform": 1}[global_config.ssim_window]
    _apply_config(g_model.ctx)
    _prefetch_tokens.pop(id(g_model), None)
    _lib.check(_lib.lib().dbm_train_iteration(g_model._h, d_model._h, n, h, w, *[_dev_ptr(train_arrays[k]) for k in _KEYS], wts,
                                              win, _lib.ONE_GEN_FORWARD if share_generator_forward else 0, m.ptr), g_model.ctx.handle)
    d_optimizer.t += 1
    g_optimizer.t += 1
    return m


def train_minibatch(train_arrays, g_model, g_optimizer, d_model, d_optimizer, comm=None, share_generator_forward=False,
                    prefetch_generator_forward=True, log=None, fused=None):
    """The body of `trainer`'s training loop (srgan_train.py:1286-1309) for one minibatch of device arrays:
    train_eval_discriminator, then train_eval_generator.  With `log` (a MetricsLog) the five metrics stay on the device
    (one row of the log) and None is returned; without, they are fetched with ONE device-to-host copy and returned as
    (d_loss, d_accu, g_loss, g_psnr, g_ssim) floats."""
    prefetch = prefetch_generator_forward and not share_generator_forward
    row = log.next_row() if log is not None else None
    if fused is None:
        fused = bool(global_config.fused_iteration)
    # (a communicator the library drives itself -- DataParallel "rccl", or "gloo" on a GPU -- is exchanged inside the call;
    # torch's own all-reduce ("nccl") and sync_batch_stats need the two step calls)
    comm_ok = comm is None or (hasattr(comm, "exchanges_in_step") and comm.exchanges_in_step(g_model.ctx)
                               and not getattr(comm, "sync_batch_stats", False))
    if (fused and (prefetch or share_generator_forward) and comm_ok and g_optimizer is not None and d_optimizer is not None
            and all(_is_device(train_arrays[k]) for k in _KEYS)):
        # ONE library call for the whole minibatch (dbm_train_iteration): the same numbers as the two calls below, bit for
        # bit, with the generator's backward pass scheduled underneath the discriminator's
        m = train_iteration(train_arrays, g_model, g_optimizer, d_model, d_optimizer, metrics=row,
                            share_generator_forward=share_generator_forward)
    else:
        train_eval_discriminator(train_arrays, g_model, d_model, d_optimizer, comm=comm, sync=False,
                                 share_generator_forward=share_generator_forward, prefetch_generator_forward=prefetch,
                                 metrics=row)
        m = train_eval_generator(train_arrays, g_model, d_model, g_optimizer, comm=comm, sync=False,
                                 share_generator_forward=share_generator_forward, metrics=row)
    if log is not None:
        return None
    out = m.get()
    return float(out[0]), float(out[1]), float(out[2]), float(out[3]), float(out[4])


def trainer(i: int, columns: list, train_iter, dev_iter, g_model, g_optimizer, d_model, d_optimizer, comm=None):
    """srgan_train.py:1267-1329: one epoch of D-step/G-step minibatches, then the dev-set evaluation."""
    metrics_dict = {mn: [] for mn in columns}
    log = MetricsLog(g_model.ctx)
    while i == train_iter.epoch:
        train_arrays = device_batch(concat_examples(train_iter.dataset, train_iter.next()), g_model.ctx)
        # both steps of every minibatch are enqueued back to back; the metrics stay on the device until the epoch ends
        # (the reference's `float(...)` after each step is a host round trip during which the GPU idles; its only
        # consumer is the per-epoch mean)
        lost = pop_dropped_updates(g_model.ctx)
        if lost:  # a timeout was observed by the previous call: the rows of the minibatches whose updates were skipped
            log.invalidate_last(lost, keep_last=1)  # (the last row is the re-issued, valid, minibatch)
        train_minibatch(train_arrays, g_model, g_optimizer, d_model, d_optimizer, comm=comm, log=log)
    lost = pop_dropped_updates(g_model.ctx)  # (a timeout observed by the epoch's LAST minibatch call: same bookkeeping, this epoch's log)
    if lost:
        log.invalidate_last(lost, keep_last=1)
    rows = log.fetch()
    try:  # the epoch's last iterations have no following step call that would notice a timeout
        g_model.ctx.check_timeout()
    except _lib.DbmError as e:
        if e.code != 7:
            raise
        _, nd, ng, _ = g_model.ctx.timeout_info()
        log.invalidate_last(max(nd, ng))
        rows = log.fetch()
    keep = [r for r in range(len(rows)) if r not in set(log.invalid)]  # (rows of minibatches a timeout made void are left out)
    for col, name in enumerate(("discriminator_loss", "discriminator_accu", "generator_loss", "generator_psnr", "generator_ssim")):
        metrics_dict[name].extend(float(v) for v in rows[keep, col])
    while i == dev_iter.epoch:
        dev_arrays = concat_examples(dev_iter.dataset, dev_iter.next())
        d_dev_loss, d_dev_accu = train_eval_discriminator(dev_arrays, g_model, d_model, train=False)
        metrics_dict["val_discriminator_loss"].append(d_dev_loss)
        metrics_dict["val_discriminator_accu"].append(d_dev_accu)
        g_dev_loss, g_dev_psnr, g_dev_ssim = train_eval_generator(dev_arrays, g_model, d_model, train=False)
        metrics_dict["val_generator_loss"].append(g_dev_loss)
        metrics_dict["val_generator_psnr"].append(g_dev_psnr)
        metrics_dict["val_generator_ssim"].append(g_dev_ssim)
    return metrics_dict


class TrialPruned(RuntimeError):
    """The stand-in for optuna.structs.TrialPruned (srgan_train.py:1698-1706): diverged run."""


METRIC_NAMES = ("discriminator_loss", "discriminator_accu", "generator_loss", "generator_psnr", "generator_ssim")


def train_epochs(epochs: int, train_iter, dev_iter, g_model, g_optimizer, d_model, d_optimizer, score_fn=None,
                 save_path: str = "model/weights", best_score: float = 250.0, comm=None, progress=None):
    """The epoch loop of the reference's `objective` (srgan_train.py:1592-1706) without its SaaS parts (Comet, Optuna,
    GMT plots): per epoch one `trainer` call, the MEAN of every metric column over the epoch's minibatches
    (`dataframe.loc[i] = [np.mean(metrics_dict[metric]) ...]`, :1617-1619), the keep-best checkpoint
    (`if rmse_test < best_rmse_test: ... save_model_weights_and_architecture(...)`, :1655-1666; `best_rmse_test` starts
    at 250) and the divergence guard (:1698-1706: PSNR < 0 or NaN losses -> TrialPruned).

    score_fn(g_model) -> float is the reference's `get_deepbedmap_test_result` (RMSE on the test area; lower is better);
    None uses the epoch's mean validation generator loss.  progress(i, epoch_metrics) is called once per epoch.
    Returns (table, best_score, saved_paths): table = {column: np.ndarray[epochs]} of epoch means."""
    columns = list(METRIC_NAMES) + [f"val_{m}" for m in METRIC_NAMES]
    table = {c: np.full(epochs, np.nan, dtype=np.float64) for c in columns}
    train_iter.reset()
    dev_iter.reset()
    saved = None
    for i in range(epochs):
        metrics_dict = trainer(i=i, columns=columns, train_iter=train_iter, dev_iter=dev_iter, g_model=g_model,
                               g_optimizer=g_optimizer, d_model=d_model, d_optimizer=d_optimizer, comm=comm)
        for c in columns:
            table[c][i] = np.mean(metrics_dict[c]) if len(metrics_dict[c]) else np.nan
        epoch_metrics = {c: float(table[c][i]) for c in columns}
        if progress is not None:
            progress(i, epoch_metrics)
        score = float(score_fn(g_model)) if score_fn is not None else epoch_metrics["val_generator_loss"]
        if score < best_score:  # save generator and discriminator weights, and the generator's architecture (:1655-1666)
            best_score = score
            saved = save_model_weights_and_architecture(generator_model=g_model, discriminator_model=d_model, save_path=save_path)
        if (epoch_metrics["generator_psnr"] < 0 or np.isnan(epoch_metrics["generator_loss"])
                or np.isnan(epoch_metrics["discriminator_loss"])):
            raise TrialPruned(f"epoch {i}: generator_psnr {epoch_metrics['generator_psnr']}, losses "
                              f"{epoch_metrics['generator_loss']} / {epoch_metrics['discriminator_loss']}")
    return table, best_score, saved


def save_model_weights_and_architecture(generator_model, discriminator_model, save_path: str = "model/weights"):
    """srgan_train.py:1333-1383: two Chainer-layout .npz files (+ a plain layer listing as .dot)."""
    os.makedirs(name=save_path, exist_ok=True)
    gpath = os.path.join(save_path, "srgan_generator_model_weights.npz")
    save_npz(file=gpath, obj=generator_model)
    dpath = os.path.join(save_path, "srgan_discriminator_model_weights.npz")
    save_npz(file=dpath, obj=discriminator_model)
    apath = os.path.join(save_path, "srgan_generator_model_architecture.dot")
    with open(apath, "w") as f:
        names = [n for n, _ in generator_model.namedparams() if n.endswith("/W")]
        f.write("digraph generator {\n" + "".join(f'  "{n}";\n' for n in names) + "}\n")
    return gpath, dpath, apath
