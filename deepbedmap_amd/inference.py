"""Whole-area tiled inference with the generator (reference deepbedmap.py:634-741).

The reference cuts Antarctica into 1000 x 1000 px (250 km) output tiles, feeds each one's low-resolution crop --
extended by `xtrapad` low-resolution pixels of halo plus the 1-pixel border the valid input convolutions consume --
through `model.forward`, trims 4*xtrapad output pixels of halo on every side and pastes the rest into a NaN-filled
canvas.  Same loop, same index arithmetic, in two forms: `predict_tiled` copies the crops to the GPU per tile
exactly like the reference's `xp.asarray(...)` (deepbedmap.py:707-722); `predict_tiled_resident` keeps the four
input grids and the output canvas in HBM (12.5 GB for the whole continent: 4 % of an MI355X) and crops / pastes
with pitched device-to-device copies, so that nothing crosses PCIe between the upload and the final download.
"""
import ctypes as C
import dataclasses

import numpy as np

from . import _lib
from .srgan import DeviceArray, to_device, using_config


@dataclasses.dataclass(frozen=True)
class Shape:  # deepbedmap.py:680-684
    y: int
    x: int


def clip_inputs(W1_tile, W2_tile, W3_tile):
    """deepbedmap.py:663-665: ice surface elevation, velocity and accumulation clipped to >= 0.  NumPy arrays are clipped
    on the host (new arrays, like np.clip); grids that already live in HBM (DeviceArray) are clipped there, in place
    (dbm_clip_min_f32) -- nothing crosses PCIe."""
    out = []
    for a in (W1_tile, W2_tile, W3_tile):
        if isinstance(a, DeviceArray):
            _lib.check(_lib.lib().dbm_clip_min_f32(a.ctx.handle, C.c_void_p(a.ptr), a.size, 0.0), a.ctx.handle)
            a._gen += 1
            out.append(a)
        else:
            out.append(np.clip(a=a, a_min=0.0, a_max=None))
    return tuple(out)


def tile_steps(final_shape: Shape, stride: Shape):
    """deepbedmap.py:700-703"""
    return [Shape(y=y_step, x=x_step) for y_step in range(0, final_shape.y, stride.y)
            for x_step in range(0, final_shape.x, stride.x)]


def crop_bounds(step: Shape, final_shape: Shape, ary_shape: Shape, xtrapad: Shape):
    """Low-resolution crop window of one output tile (deepbedmap.py:706-711)."""
    y0 = max(0, (step.y // 4) - xtrapad.y - 1)
    y1 = min(final_shape.y // 4, ((step.y + ary_shape.y) // 4) + xtrapad.y + 1)
    x0 = max(0, (step.x // 4) - xtrapad.x - 1)
    x1 = min(final_shape.x // 4, ((step.x + ary_shape.x) // 4) + xtrapad.x + 1)
    return y0, y1, x0, x1


def group_tiles_by_crop_shape(final_shape: Shape, ary_shape: Shape, stride: Shape, xtrapad: Shape, rank=0, world=1):
    """The rank's tiles (dealt round-robin in the reference's loop order, :700-703) as {(crop height, crop width): [(y0, y1, x0, x1), ...]}
    (tiles of one shape keep the loop's order; the groups are pasted one after the other, which equals the reference's paste order
    whenever pasted regions do not overlap, i.e. stride >= ary_shape -- predict_tiled_resident checks it):
    what predict_tiled_resident batches.  The continent (18000 x 22000, 1000-pixel tiles, xtrapad 18): 396 tiles, 320 of them
    288 x 288 low-resolution pixels, 40 + 32 along two edges (269 x 288 / 288 x 269) and 4 corner crops of 269 x 269."""
    groups = {}
    for i, step in enumerate(tile_steps(final_shape, stride)):
        if i % world != rank:
            continue
        y0, y1, x0, x1 = crop_bounds(step, final_shape, ary_shape, xtrapad)
        groups.setdefault((y1 - y0, x1 - x0), []).append((y0, y1, x0, x1))
    return groups


def predict_tiled(model, X_tile, W1_tile, W2_tile, W3_tile, final_shape=Shape(y=18000, x=22000),
                  ary_shape=Shape(y=1000, x=1000), stride=Shape(y=1000, x=1000), xtrapad=Shape(y=18, x=18), rank=0,
                  world=1, dtype="float32"):
    """deepbedmap.py:689-741.  X (1,1,H,W), W1 (1,1,10H,10W), W2 (1,2,2H,2W), W3 (1,1,H,W) with
    (4H, 4W) == final_shape.  Returns Y_hat (1, 4H, 4W) float32, NaN where nothing was written (the outer frame and,
    for world > 1, the tiles of the other ranks: tiles are dealt round-robin, no collective on the data path).
    dtype="bfloat16": the convolutions multiply in bf16 (fp32 accumulation and storage), BASELINE.json config 5."""
    Y_hat = np.full(shape=(1, final_shape.y, final_shape.x), fill_value=np.nan, dtype=np.float32)
    steps = tile_steps(final_shape, stride)
    for i, step in enumerate(steps):
        if i % world != rank:
            continue
        y0, y1, x0, x1 = crop_bounds(step, final_shape, ary_shape, xtrapad)
        X_crop = np.ascontiguousarray(X_tile[:, :, y0:y1, x0:x1], dtype=np.float32)
        W1_crop = np.ascontiguousarray(W1_tile[:, :, y0 * 10:y1 * 10, x0 * 10:x1 * 10], dtype=np.float32)
        W2_crop = np.ascontiguousarray(W2_tile[:, :, y0 * 2:y1 * 2, x0 * 2:x1 * 2], dtype=np.float32)
        W3_crop = np.ascontiguousarray(W3_tile[:, :, y0:y1, x0:x1], dtype=np.float32)
        with using_config(name="enable_backprop", value=False), using_config(name="dtype", value=dtype):
            Y_pred = model.forward(x=X_crop, w1=W1_crop, w2=W2_crop, w3=W3_crop)
        y_slice = slice((y0 + xtrapad.y + 1) * 4, (y1 - xtrapad.y - 1) * 4)
        x_slice = slice((x0 + xtrapad.x + 1) * 4, (x1 - xtrapad.x - 1) * 4)
        Y_pred_uncut = np.asarray(Y_pred.array)[0, :, :, :]
        Y_hat[:, y_slice, x_slice] = Y_pred_uncut[:, xtrapad.y * 4:-xtrapad.y * 4, xtrapad.x * 4:-xtrapad.x * 4]
    model.ctx.check_timeout()  # (see predict_tiled_resident)
    return Y_hat


def predict_tiled_resident(model, X_tile, W1_tile, W2_tile, W3_tile, final_shape=Shape(y=18000, x=22000),
                           ary_shape=Shape(y=1000, x=1000), stride=Shape(y=1000, x=1000), xtrapad=Shape(y=18, x=18), rank=0,
                           world=1, download=True, dtype="float32", clip=False, crops_per_batch=1):
    """predict_tiled with the grids resident in HBM.  Inputs are NumPy arrays (uploaded once) or DeviceArrays of the
    same shapes as for predict_tiled.  Returns Y_hat as a NumPy array (download=True) or as the device canvas.
    clip=True: W1, W2, W3 are clipped to >= 0 (deepbedmap.py:663-665) on the device, after the upload.  NOTE: DeviceArrays passed in
    are clipped IN PLACE (np.clip in the reference returns new arrays) -- a caller that reuses its resident W1 / W2 / W3 grids sees
    the clipped values afterwards; pass copies if the raw grids are still needed.
    crops_per_batch > 1: crops of equal shape go through the generator that many at a time (the reference's loop, :704-741, is one
    crop per forward; per crop the arithmetic is the same, so is the canvas) -- 6.5 -> 5.9 ms per 288 x 288 bf16 crop at 8."""
    ctx = model.ctx
    lib = _lib.lib()
    if stride.y < ary_shape.y or stride.x < ary_shape.x:
        # overlapping pastes: the reference's loop order decides which tile's pixels survive; grouping by crop shape would change it
        raise ValueError("predict_tiled_resident needs stride >= ary_shape (non-overlapping pasted regions); use predict_tiled")
    grids = [a if isinstance(a, DeviceArray) else to_device(a, ctx) for a in (X_tile, W1_tile, W2_tile, W3_tile)]
    if clip:
        grids[1:] = clip_inputs(*grids[1:])
    scale = (1, 10, 2, 1)  # pixels of each grid per low-resolution pixel
    canvas = DeviceArray((1, final_shape.y, final_shape.x), ctx)
    _lib.check(lib.dbm_fill_f32(ctx.handle, C.c_void_p(canvas.ptr), canvas.size, float("nan")), ctx.handle)

    def copy2d(dst, dpitch, src, spitch, width, height):  # in floats
        _lib.check(lib.dbm_memcpy2d_d2d(ctx.handle, C.c_void_p(dst), 4 * dpitch, C.c_void_p(src), 4 * spitch, 4 * width,
                                        height), ctx.handle)

    # the rank's tiles grouped by crop shape (320 of the continent's 396 crops are 288 x 288), each group in batches of
    # `crops_per_batch` crops per generator forward: same arithmetic per crop, fewer and fuller launches
    groups = group_tiles_by_crop_shape(final_shape, ary_shape, stride, xtrapad, rank, world)
    B = max(1, int(crops_per_batch))
    for (h, w), tiles in groups.items():
        nb_max = min(B, len(tiles))
        # crop staging buffers, reused by every batch of this shape
        bufs = [DeviceArray((nb_max, g.shape[1], k * h, k * w), ctx) for g, k in zip(grids, scale)]
        Ho, Wo = 4 * (h - 2), 4 * (w - 2)
        rows, cols = Ho - 8 * xtrapad.y, Wo - 8 * xtrapad.x
        for b0 in range(0, len(tiles), nb_max):
            batch = tiles[b0:b0 + nb_max]
            nb = len(batch)
            for j, (y0, y1, x0, x1) in enumerate(batch):
                for g, k, b in zip(grids, scale, bufs):
                    H, W = g.shape[2], g.shape[3]
                    nc = g.shape[1]
                    for c in range(nc):
                        copy2d(b.ptr + 4 * ((j * nc + c) * (k * h) * (k * w)), k * w, g.ptr + 4 * (c * H * W + (k * y0) * W + k * x0), W,
                               k * w, k * h)
            ins = bufs if nb == nb_max else [DeviceArray((nb,) + b.shape[1:], ctx, ptr=b.ptr, owner=b) for b in bufs]
            with using_config(name="enable_backprop", value=False), using_config(name="dtype", value=dtype):
                Y_pred = model.forward(x=ins[0], w1=ins[1], w2=ins[2], w3=ins[3])
            if rows <= 0 or cols <= 0:   # (a crop clamped by the area's border down to its halo: nothing left to paste)
                continue
            for j, (y0, y1, x0, x1) in enumerate(batch):
                ys, xs = (y0 + xtrapad.y + 1) * 4, (x0 + xtrapad.x + 1) * 4
                copy2d(canvas.ptr + 4 * (ys * final_shape.x + xs), final_shape.x,
                       Y_pred.array.ptr + 4 * (j * Ho * Wo + 4 * xtrapad.y * Wo + 4 * xtrapad.x), Wo, cols, rows)
    # small crops (9 x 9 trunk planes) run the persistent trunk kernels, whose workgroups wait for each other: a time-out (another
    # process holding the GPU) is reported here instead of in a canvas with wrong tiles -- status 7: run the sweep again
    ctx.synchronize()
    ctx.check_timeout()
    return canvas.get() if download else canvas


def merge_ranks(parts):
    """Combine the per-rank canvases of predict_tiled (each pixel is written by exactly one rank)."""
    out = np.array(parts[0], copy=True)
    for p in parts[1:]:
        m = ~np.isnan(p)
        out[m] = p[m]
    return out
