"""The N > 1 path on CPU: two processes, gloo backend, no GPU.  Checks the host logic of deepbedmap_amd.parallel
(rank-contiguous sharding, flat gradient-arena sum all-reduce, 1/world scaling applied by Adam, parameter broadcast)
with the oracle standing in for the per-rank compute (allowed in tests only): two ranks at batch 2 must give the
generator the same update as one process at batch 4 (the generator has no batch-coupled layer, and every generator
loss term is a batch mean)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _arrays(n):
    r = lambda *s: np.random.RandomState(seed=42).rand(*s).astype(np.float32)  # noqa: E731
    return {"X": r(n, 1, 11, 11), "W1": r(n, 1, 110, 110), "W2": r(n, 2, 22, 22), "W3": r(n, 1, 11, 11),
            "Y": r(n, 1, 36, 36)}


class _OracleModel:
    """Adapter exposing the two arenas DataParallel needs, as CPU torch tensors."""

    def __init__(self, model):
        self.model = model
        self.names = sorted(model.params)
        self._grad = torch.zeros(sum(model.params[k].size for k in self.names), dtype=torch.float32)

    def grad_arena(self):
        return self._grad

    def param_arena(self):
        self._p = torch.from_numpy(np.concatenate([self.model.params[k].ravel() for k in self.names]))
        return self._p

    def mark_params_changed(self):
        off = 0
        for k in self.names:
            n = self.model.params[k].size
            self.model.params[k][...] = self._p[off:off + n].numpy().reshape(self.model.params[k].shape)
            off += n

    def load_grads(self, G):
        self._grad.copy_(torch.from_numpy(np.concatenate([G[k].ravel() for k in self.names]).astype(np.float32)))

    def grads_dict(self, scale):
        out, off = {}, 0
        g = self._grad.numpy() * np.float32(scale)
        for k in self.names:
            n = self.model.params[k].size
            out[k] = g[off:off + n].reshape(self.model.params[k].shape)
            off += n
        return out


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from deepbedmap_amd.parallel import DataParallel, shard_batch
    from oracle import model as omodel
    from oracle import train as otrain

    comm = DataParallel(backend="gloo", device="cpu")
    assert comm.world == world and comm.rank == rank
    g = omodel.GeneratorModel(num_residual_blocks=1, seed=100 + rank)  # ranks start from DIFFERENT weights ...
    for k in g.params:
        if k.endswith("/W"):
            g.params[k] *= np.float32(5.0)
    wrap = _OracleModel(g)
    comm.broadcast_params(wrap, src=0)  # ... and must all end up with rank 0's
    arrays = shard_batch(_arrays(4), rank, world)
    assert len(arrays["X"]) == 2
    y = g.forward(arrays["X"], arrays["W1"], arrays["W2"], arrays["W3"], keep=True)
    gy = otrain.calculate_generator_loss_backward(y, arrays["Y"], arrays["X"][:, :, 1:-1, 1:-1])
    wrap.load_grads(g.backward(gy))
    scale = comm.allreduce_grads(wrap)
    assert scale == 1.0 / world
    opt = otrain.Adam(g.params, alpha=1e-3, eps=1e-7)
    opt.update(wrap.grads_dict(scale))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **g.params)
    assert abs(comm.max_over_ranks(float(rank)) - (world - 1)) < 1e-12
    comm.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_single_process(tmp_path):
    sys.path.insert(0, ROOT)
    from oracle import model as omodel
    from oracle import train as otrain

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = dict(np.load(tmp_path / "rank0.npz"))
    r1 = dict(np.load(tmp_path / "rank1.npz"))
    for k in r0:  # replicas stay identical
        assert np.array_equal(r0[k], r1[k]), k
    # single process, whole batch of 4
    g = omodel.GeneratorModel(num_residual_blocks=1, seed=100)
    for k in g.params:
        if k.endswith("/W"):
            g.params[k] *= np.float32(5.0)
    p0 = {k: v.copy() for k, v in g.params.items()}
    arrays = _arrays(4)
    y = g.forward(arrays["X"], arrays["W1"], arrays["W2"], arrays["W3"], keep=True)
    gy = otrain.calculate_generator_loss_backward(y, arrays["Y"], arrays["X"][:, :, 1:-1, 1:-1])
    G = g.backward(gy)
    otrain.Adam(g.params, alpha=1e-3, eps=1e-7).update(G)
    for k in g.params:
        strong = np.abs(G[k]) > 0.05 * np.abs(G[k]).max()
        a, b = (r0[k] - p0[k])[strong], (g.params[k] - p0[k])[strong]
        assert np.allclose(a, b, atol=2e-5), k  # same Adam step wherever the gradient is not rounding noise


def test_shard_slices_cover_the_batch():
    sys.path.insert(0, ROOT)
    from deepbedmap_amd.parallel import shard_slice

    for n in (1, 7, 64, 512):
        for world in (1, 2, 3, 8):
            idx = np.concatenate([np.arange(n)[shard_slice(n, r, world)] for r in range(world)])
            assert np.array_equal(idx, np.arange(n))
            sizes = [len(np.arange(n)[shard_slice(n, r, world)]) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _run_bench(args, env=None):
    import json
    import subprocess

    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)  # `python bench.py --gpus N` as the driver starts it: no launcher environment
    e.update(env or {})
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e,
                         timeout=300)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    return res.returncode, [json.loads(ln) for ln in lines], res.stderr


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without torch.distributed.run: bench.py starts one child per rank itself (fresh
    processes, nothing exec'ed), they rendezvous on 127.0.0.1 over gloo, rank 0's single JSON line is forwarded."""
    rc, lines, err = _run_bench(["--gpus", "2", "--selftest-launcher"])
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines
    out = lines[0]
    assert out["n_gpus"] == 2 and out["ranks_seen"] == [1, 1] and out["max_over_ranks"] == 1.0
    assert out["local_rank_env"] == "0" and out["master"] == "127.0.0.1"


def test_bench_launcher_reports_a_failed_rank():
    rc, lines, err = _run_bench(["--gpus", "2", "--selftest-launcher"], env={"DBM_SELFTEST_FAIL_RANK": "1"})
    assert rc != 0


def test_bench_under_an_external_launcher_keeps_its_world():
    """Started by torch.distributed.run (the driver's N > 1 form) the environment is already there: no second spawn."""
    import subprocess

    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--selftest-launcher"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and '"n_gpus": 2' in lines[0]
