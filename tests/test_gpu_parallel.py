"""-m gpu: the data-parallel product path with TWO ranks sharing the one GPU of the test box (gloo moves the device
tensors; RCCL refuses two ranks on one device): libdbm on torch's current stream, the aliased gradient arena, one sum
all-reduce per optimizer step, 1/world folded into Adam, the one-stream generator prefetch that DataParallel selects.
Replicas must stay bitwise identical, and differ from where they started."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import deepbedmap_amd as dbm

    torch.cuda.set_device(0)
    comm = dbm.DataParallel(backend="gloo")
    ctx = dbm.Context(0)
    dbm._lib._default_ctx = ctx
    comm.attach(ctx)
    np.random.seed(100 + rank)  # ranks start from DIFFERENT weights and must all end up with rank 0's
    g, g_opt, d, d_opt = dbm.compile_srgan_model(num_residual_blocks=1, residual_scaling=0.3, learning_rate=5e-4)
    comm.broadcast_params(g)
    comm.broadcast_params(d)
    start = {k: np.array(v, copy=True) for k, v in g.serialize_dict().items()}
    r = np.random.RandomState(7)
    full = {"X": r.rand(4, 1, 11, 11), "W1": r.rand(4, 1, 110, 110), "W2": r.rand(4, 2, 22, 22), "W3": r.rand(4, 1, 11, 11),
            "Y": r.rand(4, 1, 36, 36)}
    batch = dbm.device_batch({k: v.astype(np.float32) for k, v in dbm.shard_batch(full, rank, world).items()}, ctx)
    metrics = []
    for _ in range(2):
        metrics += list(dbm.train_eval_discriminator(batch, g, d, d_opt, comm=comm, prefetch_generator_forward=True))
        metrics += list(dbm.train_eval_generator(batch, g, d, g_opt, comm=comm))
    assert np.isfinite(metrics).all()
    changed = any(not np.array_equal(start[k], v) for k, v in g.serialize_dict().items())
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), changed=np.asarray(changed),
             **{"g/" + k: v for k, v in g.serialize_dict().items()}, **{"d/" + k: v for k, v in d.serialize_dict().items()})
    comm.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_on_one_gpu_stay_identical(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = dict(np.load(tmp_path / "rank0.npz"))
    r1 = dict(np.load(tmp_path / "rank1.npz"))
    assert bool(r0["changed"]) and bool(r1["changed"])
    for k in r0:
        if k.endswith("avg_mean") or k.endswith("avg_var") or k.endswith("/N") or k == "changed":
            continue  # BatchNorm running statistics are per-rank (standard data parallelism)
        assert np.array_equal(r0[k], r1[k]), k
