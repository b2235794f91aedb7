"""ctypes binding of libdbm.so (C ABI in include/dbm.h).

The HIP library is the ONLY compute path of this package: if it is missing or no MI355X is
visible, importing the symbols works (so CPU-only tooling can inspect the ABI) but creating a
context raises -- there is no CPU fallback.
"""
import ctypes as C
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdbm.so")
# Measurement tools only (tools/README.md): DBM_LIB=<path>/libdbm_measure.so loads the -DDBM_MEASURE build, whose work-skipping
# ablation switches the product library does not contain.  bench.py refuses to run with DBM_LIB set.
MEASURE_LIB_PATH = os.path.join(_HERE, "libdbm_measure.so")
CSRC = os.path.join(_HERE, "csrc")

DEVICE_PTRS = 1
KEEP_GRAPH = 2
BN_TRAIN = 4
BF16 = 8
ONE_GEN_FORWARD = 16  # dbm_train_iteration: one generator forward per minibatch (opt-in)

c_float_p = C.POINTER(C.c_float)
c_void_pp = C.POINTER(C.c_void_p)

# name -> (argtypes) ; every function returns int except dbm_last_error
# void allreduce_sum(void* user, float* dev, int n): the sync_batch_stats hook (dbm_set_sync_batch_stats)
ALLREDUCE_HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int)

# void allreduce_sum(void* user, float* dev, size_t n, void* hip_stream): the gradient-exchange hook (dbm_comm_set_hook)
COMM_HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

SIGNATURES = {
    "dbm_comm_unique_id": [C.c_void_p],
    "dbm_comm_init": [C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "dbm_comm_set_hook": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    "dbm_comm_destroy": [C.c_void_p],
    "dbm_comm_broadcast": [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int],
    "dbm_comm_allreduce": [C.c_void_p, C.c_void_p, C.c_size_t],
    "dbm_comm_stats": [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_int],
    "dbm_allreduce_grads": [C.c_void_p, C.POINTER(C.c_double)],
    "dbm_init": [C.c_int, c_void_pp],
    "dbm_shutdown": [C.c_void_p],
    "dbm_set_stream": [C.c_void_p, C.c_void_p],
    "dbm_synchronize": [C.c_void_p],
    "dbm_set_deterministic": [C.c_void_p, C.c_int],
    "dbm_memcpy2d_d2d": [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t],
    "dbm_fill_f32": [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float],
    "dbm_clip_min_f32": [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float],
    "dbm_gather_rows": [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_size_t],
    "dbm_profile_begin": [C.c_void_p],
    "dbm_profile_begin_serial": [C.c_void_p],
    "dbm_profile_end": [C.c_void_p, C.POINTER(C.c_double)],
    "dbm_profile_end_ex": [C.c_void_p, C.POINTER(C.c_double), C.c_int],
    "dbm_profile_end_records": [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)],
    "dbm_set_sync_batch_stats": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p],
    "dbm_f32_to_i16": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
    "dbm_lzw_encode_tiles": [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_int],
    "dbm_lzw_decode": [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)],
    "dbm_debug_inject_timeout": [C.c_void_p],
    "dbm_debug_inject_timeout_async": [C.c_void_p],
    "dbm_check_timeout": [C.c_void_p],
    "dbm_timeout_info": [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "dbm_timer": [C.c_void_p, C.c_int, C.POINTER(C.c_double)],
    "dbm_phase_marks": [C.c_void_p, C.c_int, C.c_char_p, C.c_int],
    "dbm_malloc": [C.c_void_p, C.c_size_t, c_void_pp],
    "dbm_free": [C.c_void_p, C.c_void_p],
    "dbm_memcpy_h2d": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
    "dbm_memcpy_d2h": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
    "dbm_gen_create": [C.c_void_p, C.c_int, C.c_float, C.c_int, c_void_pp],
    "dbm_disc_create": [C.c_void_p, c_void_pp],
    "dbm_model_destroy": [C.c_void_p],
    "dbm_model_num_tensors": [C.c_void_p, C.POINTER(C.c_int)],
    "dbm_model_tensor_info": [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int64),
                              C.POINTER(C.c_int)],
    "dbm_model_set_tensor": [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t],
    "dbm_model_get_tensor": [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t],
    "dbm_model_get_grad": [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t],
    "dbm_model_count_params": [C.c_void_p, C.POINTER(C.c_int64)],
    "dbm_model_cleargrads": [C.c_void_p],
    "dbm_model_param_arena": [C.c_void_p, c_void_pp, C.POINTER(C.c_size_t)],
    "dbm_model_grad_arena": [C.c_void_p, c_void_pp, C.POINTER(C.c_size_t)],
    "dbm_model_params_changed": [C.c_void_p],
    "dbm_gen_forward": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                        C.c_void_p, C.c_int],
    "dbm_gen_backward": [C.c_void_p, C.c_void_p, C.c_int],
    "dbm_disc_forward": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int],
    "dbm_disc_backward": [C.c_void_p, C.c_int, C.c_void_p, C.c_int],
    "dbm_discriminator_loss": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_int],
    "dbm_discriminator_loss_t": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_int],
    "dbm_generator_loss_t": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                             C.c_int, c_float_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int],
    "dbm_ssim_ex": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                    C.c_int],
    "dbm_generator_loss": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                           C.c_int, c_float_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int],
    "dbm_psnr": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_void_p, C.c_int],
    "dbm_ssim": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int],
    "dbm_adam_setup": [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double],
    "dbm_adam_update": [C.c_void_p, C.c_double],
    "dbm_discriminator_step": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_int, C.c_void_p],
    "dbm_generator_step": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                           C.c_void_p, C.c_void_p, c_float_p, C.c_int, C.c_int, C.c_void_p],
    "dbm_train_iteration": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_void_p, c_float_p, C.c_int, C.c_int, C.c_void_p],
    "dbm_op_conv2d": [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int] * 10,
    "dbm_op_conv2d_backward": [C.c_void_p] + [C.c_void_p] * 6 + [C.c_int] * 9,
    "dbm_op_conv2d_cl16": [C.c_void_p] + [C.c_void_p] * 4 + [C.c_float, C.c_void_p] + [C.c_int] * 6,
    "dbm_op_conv2d_cl16x3": [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int] * 7,
    "dbm_op_deform_conv2d": [C.c_void_p] + [C.c_void_p] * 5 + [C.c_int] * 5,
    "dbm_op_deform_conv2d_form": [C.c_void_p] + [C.c_void_p] * 5 + [C.c_int] * 6,
    "dbm_op_deform_conv2d_backward": [C.c_void_p] + [C.c_void_p] * 8 + [C.c_int] * 5,
}

_lib = None


class DbmError(RuntimeError):
    # libdbm status.  7: a persistent kernel timed out; the call that reports it enqueued nothing (re-issue it), the
    # optimizer updates queued since the event were skipped (`Context.timeout_info()`).  8: the same in a data-parallel
    # run -- fatal, the replicas have diverged.  9 (dbm_adam_update only): the gradients about to be applied come from a
    # void pass; nothing was applied -- repeat forward + backward, then update.
    code = None


def build(verbose=False, fresh=False):
    """Compile libdbm.so in-tree for gfx950 (hipcc cross-compiles without a GPU).  fresh=True: every translation unit is
    compiled from scratch (`make -B`, ~20 s on eight cores) -- what `__graft_entry__.build()` does, so that "it builds" never
    depends on objects that happened to lie in the tree.  Returns the library's path; `build.compiled` = the sources compiled
    by this call."""
    jobs = str(min(8, os.cpu_count() or 1))
    res = subprocess.run(["make", "-C", CSRC, "-j", jobs] + (["-B"] if fresh else []), capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise DbmError("building libdbm.so failed")
    build.compiled = sorted(set(re.findall(r"-c (\S+\.hip)", res.stdout)))
    return LIB_PATH


build.compiled = []


def library_path(environ=None):
    """libdbm.so, unless DBM_LIB names an existing libdbm_measure.so (measurement tools only; anything else is refused)."""
    override = (os.environ if environ is None else environ).get("DBM_LIB")
    if not override:
        return LIB_PATH
    if os.path.basename(override) != "libdbm_measure.so" or not os.path.exists(override):
        raise DbmError(f"DBM_LIB={override}: only an existing libdbm_measure.so (tools/build_measure.sh) may replace libdbm.so")
    return override


def lib():
    """The loaded library with argtypes set.  Raises if libdbm.so has not been built."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise DbmError(
                f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or make -C deepbedmap_amd/csrc).  deepbedmap_amd has no CPU fallback."
            )
        l = C.CDLL(path)
        l._dbm_path = path
        for name, args in SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = C.c_int
        l.dbm_last_error.argtypes = [C.c_void_p]
        l.dbm_last_error.restype = C.c_char_p
        _lib = l
    return _lib


def check(rc, ctx=None):
    if rc != 0:
        msg = lib().dbm_last_error(ctx)
        err = DbmError(f"libdbm error {rc}: {msg.decode() if msg else '?'}")
        err.code = rc
        raise err


_default_ctx = None


class Context:
    """One GPU + one HIP stream (dbm_ctx)."""

    def __init__(self, device=None):
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        self.handle = C.c_void_p()
        l = lib()
        rc = l.dbm_init(int(device), C.byref(self.handle))
        if rc != 0:
            msg = l.dbm_last_error(None)
            raise DbmError(f"dbm_init failed ({rc}): {msg.decode() if msg else '?'}")
        self.device = int(device)

    def set_stream(self, stream_ptr):
        check(lib().dbm_set_stream(self.handle, C.c_void_p(stream_ptr)), self.handle)

    def synchronize(self):
        check(lib().dbm_synchronize(self.handle), self.handle)

    def check_timeout(self):
        """Raises DbmError (code 7 / 8) if a persistent kernel gave up since the last step call (include/dbm.h)."""
        check(lib().dbm_check_timeout(self.handle), self.handle)

    def timeout_info(self):
        """(events, discriminator updates skipped, generator updates skipped, persistent kernels paused) of the last event."""
        ev, d, g, off = C.c_long(0), C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib().dbm_timeout_info(self.handle, C.byref(ev), C.byref(d), C.byref(g), C.byref(off)), self.handle)
        return int(ev.value), int(d.value), int(g.value), bool(off.value)

    def profile_records(self):
        """Ends dbm_profile_begin / _begin_serial; one dict per bracketed launch: family, flops, bytes (both algorithmic), ms, wgs (workgroups), tag."""
        n = C.c_size_t(0)
        check(lib().dbm_profile_end_records(self.handle, None, 0, C.byref(n)), self.handle)
        buf = C.create_string_buffer(n.value + 1)
        check(lib().dbm_profile_end_records(self.handle, buf, n.value + 1, C.byref(n)), self.handle)
        recs = []
        for line in buf.value.decode().splitlines():
            f, fl, by, ms, wgs, tag = line.split(" ", 5)
            recs.append({"family": int(f), "flops": float(fl), "bytes": float(by), "ms": float(ms), "wgs": int(wgs), "tag": tag})
        return recs

    def malloc(self, nbytes):
        p = C.c_void_p()
        check(lib().dbm_malloc(self.handle, nbytes, C.byref(p)), self.handle)
        return p.value

    def free(self, ptr):
        check(lib().dbm_free(self.handle, C.c_void_p(ptr)), self.handle)


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx
