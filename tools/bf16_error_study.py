#!/usr/bin/env python3
"""How many METRES does the bf16 sweep mode (BASELINE config 5) cost at the reference's data range, and which layers have
to keep fp32 arithmetic?  One 288 x 288 crop (deepbedmap.py:706-728), the DEM-range generator of the fixtures
(tests/golden/make_golden_full.models_dem5), two kinds of input grids:

  white   X ~ U[-2000, 2000] with -5000 gap-fill blocks, W1 ~ U[0, 4000], ... independent per pixel (SURVEY 8d's "DEM-like"
          distribution: the worst case for a sub-pixel sampler -- neighbouring pixels differ by kilometres);
  smooth  the same ranges as smooth relief (bilinear x8 upsampling of a coarse field + 2 % white noise): what BEDMAP2 / REMA /
          MEaSUREs grids look like at 1 km .. 100 m.

Reference = the fp32 HIP path of the same process (pinned to the float64 oracle at this data range by tests/test_gpu_dem.py).  DBM_BF16_FP32_LAYERS is
read once per process: run this script once per mask (tools/bf16_error_study.sh).  Prints one JSON line.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from dem_model import dem_arrays, dem_generator  # noqa: E402


def smooth_field(r, shape, lo, hi, coarse=8, noise=0.02):
    n, c, h, w = shape
    ch, cw = h // coarse + 2, w // coarse + 2
    z = r.uniform(lo, hi, (n, c, ch, cw))
    ys = (np.arange(h) + 0.5) / coarse
    xs = (np.arange(w) + 0.5) / coarse
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    fy, fx = (ys - y0)[None, None, :, None], (xs - x0)[None, None, None, :]
    a = z[:, :, y0][:, :, :, x0] * (1 - fy) * (1 - fx) + z[:, :, y0 + 1][:, :, :, x0] * fy * (1 - fx) + \
        z[:, :, y0][:, :, :, x0 + 1] * (1 - fy) * fx + z[:, :, y0 + 1][:, :, :, x0 + 1] * fy * fx
    return (a + r.normal(0, noise * (hi - lo), shape)).astype(np.float32)


def smooth_arrays(seed, h=288, w=288):
    r = np.random.RandomState(seed)
    a = {"X": smooth_field(r, (1, 1, h, w), -2000, 2000), "W1": np.clip(smooth_field(r, (1, 1, 10 * h, 10 * w), 0, 4000, 80), 0, None),
         "W2": np.clip(smooth_field(r, (1, 2, 2 * h, 2 * w), 0, 1000, 16), 0, None),
         "W3": np.clip(smooth_field(r, (1, 1, h, w), 0, 500), 0, None)}
    a["X"][0, 0, 40:70, 100:160] = -5000.0  # a BEDMAP2 gap-fill block
    return a


def main():
    import deepbedmap_amd as dbm

    g = dem_generator(dbm, seed=909)
    out = {"DBM_BF16_FP32_LAYERS": os.environ.get("DBM_BF16_FP32_LAYERS", "(default)")}
    for kind, arrays in (("white", dem_arrays(1, 7500)), ("smooth", smooth_arrays(7600))):
        ins = [dbm.to_device(arrays[k]) for k in ("X", "W1", "W2", "W3")]
        with dbm.using_config("enable_backprop", False):
            y32 = g.forward(*ins).array.get().astype(np.float64)
            with dbm.using_config("dtype", "bfloat16"):
                g.forward(*ins)
                g.ctx.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    y16d = g.forward(*ins).array
                y16 = y16d.get().astype(np.float64)
                ms = (time.perf_counter() - t0) / 3 * 1e3
        e = y16 - y32
        out[kind] = {"range_m": float(np.abs(y32).max()), "std_m": float(y32.std()), "bf16_max_m": float(np.abs(e).max()),
                     "bf16_rms_m": float(np.sqrt((e * e).mean())), "bf16_p99_m": float(np.percentile(np.abs(e), 99)),
                     "rel_rms": float(np.sqrt((e * e).mean()) / y32.std()), "ms_per_crop_incl_alloc": ms,
                     "neighbour_step_rms_m": float(np.sqrt((np.diff(y32, axis=3) ** 2).mean()))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
