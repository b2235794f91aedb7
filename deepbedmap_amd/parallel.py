"""Data parallelism over the GPUs of one node: one process per GPU, tile minibatches sharded by rank,
the flat gradient arena is summed over ranks once per optimizer step -- by libdbm's native RCCL communicator
(dbm_comm_init: librccl over xGMI), bucket by bucket underneath the backward passes; torch.distributed ("gloo") is only
the host-side rendezvous.  The reference trains on a single
GPU (srgan_train.py:58-61, 1039-1040); this is new design.

No collective touches the data path: each rank runs the whole D-step/G-step on its own tiles; only
gradients are exchanged.  BatchNorm statistics and the RaGAN batch means stay per-rank (standard
data-parallel semantics; they differ from one process at the global batch).
"""
import os

import numpy as np


def shard_slice(n_total, rank, world):
    """Rank-contiguous slice of a global batch (remainder spread over the first ranks)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return slice(lo, lo + base + (1 if rank < rem else 0))


def shard_batch(arrays, rank, world):
    n = len(next(iter(arrays.values())))
    s = shard_slice(n, rank, world)
    return {k: v[s] for k, v in arrays.items()}


class _ForeignCuda:
    """Hands a raw device pointer to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class DataParallel:
    """comm object accepted by train_eval_discriminator / train_eval_generator (`comm=`).

    Three transports, chosen by `backend` (default: environment DBM_DIST_BACKEND, else "rccl" on a GPU, "gloo" on CPU):

    * "rccl"  (the product path) -- libdbm's NATIVE communicator (dbm_comm_init: librccl over xGMI).  The fused steps
      sum their gradient buckets over ranks themselves, underneath the backward passes, on a library stream; torch is
      only the host-side rendezvous (a gloo process group carries the 128-byte id, barriers and the timing maximum).
    * "gloo" with device arrays -- the same in-step bucket schedule, but every bucket goes through a Python hook
      (dbm_comm_set_hook) that synchronises the stream and reduces with gloo: for tests with several ranks on ONE GPU,
      where RCCL refuses to build a communicator.
    * "nccl" -- torch.distributed's RCCL on a stream shared with libdbm: one whole-arena all-reduce after each backward
      (round-1 behaviour, kept for comparison); and "gloo" on CPU tensors for the host-logic tests.
    """

    def __init__(self, backend=None, device=None, sync_batch_stats=False):
        """sync_batch_stats=True: the discriminator's BatchNorm statistics and the relativistic-average means are taken
        over the GLOBAL batch: world x N tiles then train exactly like one process at batch world*N (up to fp32
        summation order), at the price of ~40 small all-reduces per D-step.  Default False: per-rank statistics."""
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.on_gpu = torch.cuda.is_available() if device is None else (device != "cpu")
        if backend is None:
            backend = os.environ.get("DBM_DIST_BACKEND") or ("rccl" if self.on_gpu else "gloo")
        self.backend = backend
        self.native = self.on_gpu and backend == "rccl"        # libdbm's own RCCL communicator
        self.hooked = self.on_gpu and backend == "gloo"        # in-step buckets through a gloo hook
        if self.on_gpu:
            torch.cuda.set_device(self.local_rank)
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            dist.init_process_group(backend="gloo" if backend == "rccl" else backend, rank=self.rank, world_size=self.world)
        self._views = {}
        self._shared_stream = set()
        self._in_step = set()   # contexts whose fused steps exchange the gradients themselves
        self.sync_batch_stats = bool(sync_batch_stats)
        self._hooks = {}

    # ---- wiring a context ----
    def attach(self, ctx):
        """Give `ctx` its communicator.  rccl: dbm_comm_init (collective: every rank must call it).  gloo on a GPU: the
        bucket hook.  nccl: libdbm runs on torch's current HIP stream, so that torch orders the RCCL collective after the
        backward kernels and the Adam kernel after the collective by stream order alone (no host synchronisation)."""
        import ctypes as C

        from . import _lib

        lib = _lib.lib()
        if self.native:
            ident = [None]
            if self.rank == 0:
                buf = C.create_string_buffer(128)
                _lib.check(lib.dbm_comm_unique_id(buf), None)
                ident[0] = buf.raw
            if self.world > 1:
                self.dist.broadcast_object_list(ident, src=0)
            _lib.check(lib.dbm_comm_init(ctx.handle, self.rank, self.world, C.c_char_p(ident[0])), ctx.handle)
            self._in_step.add(id(ctx))
            if self.sync_batch_stats and self.world > 1:
                _lib.check(lib.dbm_set_sync_batch_stats(ctx.handle, self.world, None, None), ctx.handle)
            return
        if self.hooked:
            dev = f"cuda:{self.local_rank}"

            def bucket_hook(user, ptr, n, stream):  # blocking: everything enqueued so far, then a host-side sum
                self.torch.cuda.synchronize()
                t = self.torch.as_tensor(_ForeignCuda(ptr, n), device=dev)
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
                self.torch.cuda.synchronize()

            cb = _lib.COMM_HOOK(bucket_hook)
            self._hooks[("grad", id(ctx))] = cb
            _lib.check(lib.dbm_comm_set_hook(ctx.handle, self.rank, self.world, C.cast(cb, C.c_void_p), None), ctx.handle)
            self._in_step.add(id(ctx))
            if self.sync_batch_stats and self.world > 1:
                self._install_sync_hook(ctx)
            return
        if self.on_gpu:
            st = self.torch.cuda.current_stream()
            if st.cuda_stream == 0:
                # torch's default stream is the legacy NULL stream: it has no handle libdbm could be given
                # (dbm_set_stream(ctx, NULL) means "the context's own stream").  Make a real stream current for this
                # thread, so that the collectives torch issues and libdbm's kernels share one stream.
                self._stream = self.torch.cuda.Stream(device=self.local_rank)
                self.torch.cuda.set_stream(self._stream)
                st = self._stream
            ctx.set_stream(st.cuda_stream)
            self._shared_stream.add(id(ctx))
        if self.sync_batch_stats and self.on_gpu and self.world > 1:
            self._install_sync_hook(ctx)

    def exchanges_in_step(self, ctx):
        """True when dbm_discriminator_step / dbm_generator_step on `ctx` sum the gradients over ranks themselves."""
        return id(ctx) in self._in_step

    def step_flags(self, ctx):
        """Extra `train` bits of the fused steps: 8 = the caller's collectives run on a stream of their own (torch's RCCL)."""
        return 8 if (self.on_gpu and not self.exchanges_in_step(ctx)) else 0

    def _install_sync_hook(self, ctx):
        """dbm_set_sync_batch_stats: libdbm calls back with a small device buffer of per-rank sums; the collective goes
        onto the stream libdbm shares with torch (attach), so nothing synchronises the host."""
        import ctypes as C

        from . import _lib

        dev = f"cuda:{self.local_rank}"

        def hook(user, ptr, n):
            if self.hooked:
                self.torch.cuda.synchronize()
            t = self.torch.as_tensor(_ForeignCuda(ptr, n), device=dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            if self.hooked:
                self.torch.cuda.synchronize()

        cb = _lib.ALLREDUCE_HOOK(hook)
        self._hooks[id(ctx)] = cb  # keep the trampoline alive as long as the context may call it
        _lib.check(_lib.lib().dbm_set_sync_batch_stats(ctx.handle, self.world, C.cast(cb, C.c_void_p), None), ctx.handle)

    def detach(self, ctx):
        from . import _lib

        if id(ctx) in self._in_step:
            _lib.check(_lib.lib().dbm_comm_destroy(ctx.handle), ctx.handle)
            self._in_step.discard(id(ctx))

    # ---- collectives ----
    def grad_view(self, model):
        """torch view of the model's flat gradient arena (device memory owned by libdbm)."""
        key = id(model)
        if key not in self._views:
            arena = model.grad_arena()
            if isinstance(arena, self.torch.Tensor):
                t = arena
            else:
                t = self.torch.as_tensor(_ForeignCuda(arena.ptr, arena.size), device=f"cuda:{self.local_rank}")
                assert t.data_ptr() == arena.ptr, "torch copied the gradient arena instead of aliasing it"
            self._views[key] = t
        return self._views[key]

    def allreduce_grads(self, model):
        """Sum the gradient arena over ranks; returns the scale (1/world) the optimizer applies.  After a fused step on a
        context with its own communicator there is nothing left to do: the step has exchanged every bucket."""
        ctx = getattr(model, "ctx", None)
        if self.world > 1 and not (ctx is not None and self.exchanges_in_step(ctx)):
            t = self.grad_view(model)
            shared = (not self.on_gpu) or id(model.ctx) in self._shared_stream
            if self.on_gpu and not shared:
                model.ctx.synchronize()  # libdbm's own stream: order the collective after it by a host wait
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            if self.on_gpu and not shared:
                self.torch.cuda.current_stream().synchronize()
        return 1.0 / self.world

    def allreduce_grads_now(self, model):
        """Whole-arena exchange for callers that ran model.backward() themselves (no fused step): dbm_allreduce_grads."""
        import ctypes as C

        from . import _lib

        ctx = getattr(model, "ctx", None)
        if ctx is not None and self.exchanges_in_step(ctx):
            scale = C.c_double(1.0)
            _lib.check(_lib.lib().dbm_allreduce_grads(model._h, C.byref(scale)), ctx.handle)
            return float(scale.value)
        return self.allreduce_grads(model)

    def broadcast_params(self, model, src=0):
        """Make every rank start from rank `src`'s parameters."""
        if self.world > 1:
            arena = model.param_arena()
            ctx = getattr(model, "ctx", None)
            if self.native and ctx is not None and self.exchanges_in_step(ctx):
                import ctypes as C

                from . import _lib

                _lib.check(_lib.lib().dbm_comm_broadcast(ctx.handle, C.c_void_p(arena.ptr), arena.size, int(src)), ctx.handle)
                ctx.synchronize()
                model.mark_params_changed()
                return
            t = arena if isinstance(arena, self.torch.Tensor) else self.torch.as_tensor(
                _ForeignCuda(arena.ptr, arena.size), device=f"cuda:{self.local_rank}")
            if self.on_gpu:
                model.ctx.synchronize()
            self.dist.broadcast(t, src=src)
            if self.on_gpu:
                self.torch.cuda.current_stream().synchronize()
            if hasattr(model, "mark_params_changed"):
                model.mark_params_changed()

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, value):
        if self.world == 1:
            return value
        on_dev = self.on_gpu and self.backend == "nccl"  # the gloo control group reduces host tensors
        t = self.torch.tensor([value], dtype=self.torch.float64, device=f"cuda:{self.local_rank}" if on_dev else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())
