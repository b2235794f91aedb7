#!/usr/bin/env python3
"""One interior 288 x 288 crop of the continent sweep, fp32 and bf16, a few times (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import deepbedmap_amd as dbm  # noqa: E402

ctx = dbm.Context(0)
dbm._lib._default_ctx = ctx
np.random.seed(1)
g = dbm.GeneratorModel(num_residual_blocks=12)
import json
print(json.dumps(bench.sweep_leg(dbm, ctx, g, crops=int(os.environ.get("CROPS", "5")))))
