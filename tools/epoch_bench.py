"""End-to-end epoch throughput: `trainer` over a device-resident synthetic dataset of the reference's size
(3826 training / 202 development tiles, batch 128: paper/tc-2020-74.tex:629-631, srgan_train.py:132-166, 1267-1329), i.e.
what the reference's "150 epochs in about 30 min on a V100" (≈319 tiles/s) measures minus its plotting / uploads.
usage (GPU box): python tools/epoch_bench.py [epochs]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepbedmap_amd as dbm  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = 4028
r = np.random.RandomState(42)
ds = {"X": r.rand(n, 1, 11, 11), "W1": r.rand(n, 1, 110, 110), "W2": r.rand(n, 2, 22, 22), "W3": r.rand(n, 1, 11, 11),
      "Y": r.rand(n, 1, 36, 36)}
ds = dbm.dataset_to_device({k: v.astype(np.float32) for k, v in ds.items()})
g, go, d, do = dbm.compile_srgan_model(num_residual_blocks=12, residual_scaling=0.1, learning_rate=1.6e-4)
train_iter, n_train, dev_iter, n_dev = dbm.get_train_dev_iterators(ds, first_size=int(n * 0.95), batch_size=128, seed=42)
cols = ["discriminator_loss", "discriminator_accu", "generator_loss", "generator_psnr", "generator_ssim",
        "val_discriminator_loss", "val_discriminator_accu", "val_generator_loss", "val_generator_psnr", "val_generator_ssim"]
dbm.trainer(0, cols, train_iter, dev_iter, g, go, d, do)  # warm-up epoch (allocations, plans)
g.ctx.synchronize()
t0 = time.perf_counter()
for i in range(1, 1 + epochs):
    m = dbm.trainer(i, cols, train_iter, dev_iter, g, go, d, do)
g.ctx.synchronize()
dt = (time.perf_counter() - t0) / epochs
print(f"{dt:.3f} s per epoch ({n_train} training + {n_dev} development tiles, batch 128): {n_train / dt:.0f} training tiles/s "
      f"end to end; last epoch: g_loss {np.mean(m['generator_loss']):.4f} psnr {np.mean(m['val_generator_psnr']):.2f}")
