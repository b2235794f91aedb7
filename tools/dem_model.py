"""A generator whose activations stay at the reference's DATA magnitude (O(10^3): metres, m/yr, kg/m2/yr) from the input block to
the output, like a trained model's -- for the measurement tools (bf16 error in metres, continent sweep).  Built with the PRODUCT
only: the reference initialisation of deepbedmap_amd.GeneratorModel, then the recipe of the DEM fixtures
(tests/golden/make_golden_full.oracle_generator_dem): layers on the main path (input block, pre-residual, upsampling, deformable
GEMMs) x 10 (He scale 1), residual branches (trunk, post-residual) x 3, offset convolutions x 10 x 1e-3 (offsets of about a
pixel), biases ~ N(0, 0.1).  Nothing under oracle/ is imported."""
import numpy as np


def dem_generator(dbm, seed=909, num_residual_blocks=12):
    np.random.seed(seed)
    g = dbm.GeneratorModel(num_residual_blocks=num_residual_blocks)
    r = np.random.RandomState(seed + 1)
    for k in sorted(g._tensors):
        p = g._tensors[k]
        a = np.asarray(p.array, dtype=np.float32)
        if k.endswith("/W"):
            sc = 3.0 if (k.startswith("residual_network/") or k.startswith("post_residual")) else 10.0
            if "offset_conv" in k:
                sc *= 1e-3
            p.array = a * np.float32(sc)
        elif k.endswith("/b"):
            p.array = a + r.normal(0, 0.1, a.shape).astype(np.float32)
    return g


def dem_arrays(n, seed, h=288, w=288):
    """Raw physical units as the reference feeds them (deepbedmap.py:164-169, 663-665): X ~ U[-2000, 2000] m with -5000 m gap-fill
    blocks, W1 ~ U[0, 4000] m, W2 ~ U[0, 1000] m/yr, W3 ~ U[0, 500] kg/m2/yr, independent per pixel."""
    r = [np.random.RandomState(seed + i) for i in range(5)]
    f = np.float32
    a = {"X": r[0].uniform(-2000, 2000, (n, 1, h, w)).astype(f), "W1": r[1].uniform(0, 4000, (n, 1, 10 * h, 10 * w)).astype(f),
         "W2": r[2].uniform(0, 1000, (n, 2, 2 * h, 2 * w)).astype(f), "W3": r[3].uniform(0, 500, (n, 1, h, w)).astype(f)}
    for i in range(n):
        for _ in range(6):
            bh, bw = r[4].randint(2, max(3, h // 4)), r[4].randint(2, max(3, w // 4))
            y0, x0 = r[4].randint(0, h - bh + 1), r[4].randint(0, w - bw + 1)
            a["X"][i, 0, y0:y0 + bh, x0:x0 + bw] = -5000.0
    return a
