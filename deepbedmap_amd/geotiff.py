"""GeoTIFF writer for the stitched DEM (reference deepbedmap.py:749-756 -> data_prep.py:779-834).

The reference saves `Y_hat.astype(np.int16)` through rasterio / GDAL: driver GTiff, one band, `dtype=int16`,
`nodata=-2000`, `tiled=True`, `compress=lzw`, `bigtiff=YES`, polar stereographic CRS, the affine transform of
`rasterio.transform.from_bounds(*window_bound, height, width)`.  rasterio / GDAL are not part of this framework: the
container is written here (classic TIFF or BigTIFF, little endian, 256 x 256 tiles as GDAL's default block size, GeoTIFF
tags ModelPixelScale / ModelTiepoint / GeoKeyDirectory with ProjectedCSType = EPSG:3031, GDAL_NODATA), the LZW streams come
from libdbm (dbm_lzw_encode_tiles: TIFF 6.0 LZW, host threads over tiles) and the int16 cast of a device-resident canvas
from the GPU (dbm_f32_to_i16: NumPy's astype semantics, NaN frame -> 0).  `read_geotiff` decodes the file again
(bit-exact round trip; the tests also decode it with Pillow / libtiff).
"""
import ctypes as C
import os
import struct

import numpy as np

from . import _lib

TILE = 256  # GDAL's default block size for tiled=True
EPSG_ANTARCTIC_POLAR_STEREOGRAPHIC = 3031  # "+proj=stere +lat_0=-90 +lat_ts=-71 +lon_0=0 ..." (data_prep.py:784)

_TYPES = {1: ("B", 1), 2: ("s", 1), 3: ("H", 2), 4: ("I", 4), 12: ("d", 8), 16: ("Q", 8)}


def canvas_to_int16(canvas):
    """`Y_hat.astype(np.int16)` (deepbedmap.py:752) for a NumPy array or a DeviceArray (converted on the GPU: half the
    bytes cross PCIe).  Returns a NumPy int16 array of the same shape."""
    from .srgan import DeviceArray

    if isinstance(canvas, DeviceArray):
        ctx = canvas.ctx
        n = canvas.size
        dst = ctx.malloc(2 * n + 16)
        try:
            lib = _lib.lib()
            _lib.check(lib.dbm_f32_to_i16(ctx.handle, C.c_void_p(canvas.ptr), C.c_void_p(dst), n), ctx.handle)
            out = np.empty(canvas.shape, dtype=np.int16)
            _lib.check(lib.dbm_memcpy_d2h(ctx.handle, out.ctypes.data_as(C.c_void_p), C.c_void_p(dst), 2 * n), ctx.handle)
        finally:
            ctx.free(dst)
        return out
    with np.errstate(invalid="ignore"):
        return np.asarray(canvas).astype(np.int16)


def lzw_encode_tiles(tiles, nthreads=None):
    """tiles: (ntiles, tile_bytes) uint8 -> list of bytes objects (TIFF 6.0 LZW streams)."""
    tiles = np.ascontiguousarray(tiles, dtype=np.uint8)
    nt, nb = tiles.shape
    stride = nb * 3 // 2 + 64
    out = np.empty((nt, stride), dtype=np.uint8)
    sizes = (C.c_size_t * max(nt, 1))()
    rc = _lib.lib().dbm_lzw_encode_tiles(tiles.ctypes.data_as(C.c_void_p), nb, nt, out.ctypes.data_as(C.c_void_p), stride, sizes,
                                         int(nthreads or min(16, os.cpu_count() or 1)))
    if rc != 0:
        raise _lib.DbmError(f"dbm_lzw_encode_tiles failed ({rc})")
    return [out[i, :sizes[i]].tobytes() for i in range(nt)]


def lzw_decode(stream, nbytes):
    src = np.frombuffer(stream, dtype=np.uint8)
    dst = np.empty(nbytes + 16, dtype=np.uint8)
    got = C.c_size_t()
    rc = _lib.lib().dbm_lzw_decode(src.ctypes.data_as(C.c_void_p), src.size, dst.ctypes.data_as(C.c_void_p), dst.size, C.byref(got))
    if rc != 0 or got.value != nbytes:
        raise _lib.DbmError(f"dbm_lzw_decode failed ({rc}, {got.value} of {nbytes} bytes)")
    return dst[:nbytes]


def _tiles_of(band, th, tw):
    """(H, W) -> (ntiles, th, tw), row-major tile order, edge tiles zero padded (TIFF 6.0 section 15)."""
    H, W = band.shape
    ny, nx = (H + th - 1) // th, (W + tw - 1) // tw
    padded = np.zeros((ny * th, nx * tw), dtype=band.dtype)
    padded[:H, :W] = band
    return padded.reshape(ny, th, nx, tw).transpose(0, 2, 1, 3).reshape(ny * nx, th, tw), ny, nx


def _epsg_of(crs):
    """EPSG code of `crs`: an int, "EPSG:3031", or the reference's default PROJ string for Antarctic polar stereographic
    (data_prep.py:784: "+proj=stere +lat_0=-90 +lat_ts=-71 +lon_0=0 ... +datum=WGS84 ..." = EPSG:3031).  Other PROJ strings
    would need a projection database: refused with a clear message."""
    if isinstance(crs, (int, np.integer)):
        return int(crs)
    text = str(crs).strip()
    if text.isdigit():
        return int(text)
    if text.upper().startswith("EPSG:") and text[5:].strip().isdigit():
        return int(text[5:])
    if text.startswith("+"):
        kv = dict((t.lstrip("+").split("=") + [""])[:2] for t in text.split())
        f = lambda k, d=0.0: float(kv.get(k, d) or d)  # noqa: E731
        if (kv.get("proj") == "stere" and f("lat_0") == -90.0 and f("lat_ts") == -71.0 and f("lon_0") == 0.0 and f("k", 1.0) == 1.0
                and f("x_0") == 0.0 and f("y_0") == 0.0 and kv.get("datum", kv.get("ellps", "WGS84")) == "WGS84"
                and kv.get("units", "m") == "m"):
            return EPSG_ANTARCTIC_POLAR_STEREOGRAPHIC
    raise ValueError(f"save_array_to_grid: crs {crs!r} is not an EPSG code, 'EPSG:n' or the Antarctic polar stereographic "
                     "PROJ string of the reference (EPSG:3031)")


def save_array_to_grid(outfilepath, window_bound, array, save_netcdf=False, crs=EPSG_ANTARCTIC_POLAR_STEREOGRAPHIC, dtype=None,
                       nodataval=-2000, tiled=False, compression="none", bigtiff=True, nthreads=None):
    """data_prep.py:779-834 without rasterio: writes `{outfilepath}.tif` and returns its path.

    window_bound = (minx, miny, maxx, maxy); array is CHW with one channel (a NumPy array, or a DeviceArray canvas when
    dtype is int16); compression "none" or "lzw" (rasterio.enums.Compression values); tiled=False writes one strip per
    row block of 256 rows."""
    if save_netcdf:
        raise NotImplementedError("NetCDF output (xarray) is outside this framework; convert the GeoTIFF with GDAL")
    assert len(array.shape) == 3 and array.shape[0] == 1  # one band, CHW (data_prep.py:800-801)
    dt = np.dtype(dtype if dtype is not None else getattr(array, "dtype", np.float32))
    if dt == np.int16 and not isinstance(array, np.ndarray):
        band = canvas_to_int16(array)[0]
    else:
        band = np.asarray(array)[0]
        if band.dtype != dt:
            with np.errstate(invalid="ignore"):
                band = band.astype(dt)
    band = np.ascontiguousarray(band.astype(dt.newbyteorder("<"), copy=False))
    H, W = band.shape
    if dt.kind == "f":
        sample_format = 3
    elif dt.kind == "i":
        sample_format = 2
    elif dt.kind == "u":
        sample_format = 1
    else:
        raise ValueError(f"unsupported dtype {dt}")
    epsg = _epsg_of(crs)
    th, tw = (TILE, TILE) if tiled else (min(TILE, H), W)
    blocks, ny, nx = _tiles_of(band, th, tw)
    raw = blocks.reshape(len(blocks), -1).view(np.uint8)
    # strips: the last one holds only the rows that exist (TIFF 6.0: StripByteCounts of H % RowsPerStrip rows, what GDAL
    # writes); tiles are always whole (zero padded)
    last_rows = H - (ny - 1) * th
    short_last = (not tiled) and last_rows < th
    if str(compression).lower() == "lzw":
        if short_last:
            streams = lzw_encode_tiles(raw[:-1], nthreads) if ny > 1 else []
            streams += lzw_encode_tiles(np.ascontiguousarray(raw[-1:, :last_rows * tw * dt.itemsize]), nthreads)
        else:
            streams = lzw_encode_tiles(raw, nthreads)
        comp = 5
    elif str(compression).lower() in ("none", "1"):
        streams, comp = [r.tobytes() for r in raw], 1
        if short_last:
            streams[-1] = streams[-1][:last_rows * tw * dt.itemsize]
    else:
        raise ValueError(f"unsupported compression {compression!r} (none, lzw)")
    minx, miny, maxx, maxy = (float(v) for v in window_bound)
    px, py = (maxx - minx) / W, (maxy - miny) / H  # rasterio.transform.from_bounds
    nodata = (repr(int(nodataval)) if float(nodataval).is_integer() else repr(float(nodataval))).encode() + b"\0"
    geokeys = [1, 1, 0, 3, 1024, 0, 1, 1, 1025, 0, 1, 1, 3072, 0, 1, epsg]  # projected, PixelIsArea, ProjectedCSType
    off_t = 16 if bigtiff else 4  # LONG8 / LONG
    tags = [
        (256, 4, [W]), (257, 4, [H]), (258, 3, [8 * dt.itemsize]), (259, 3, [comp]), (262, 3, [1]), (277, 3, [1]),
        (284, 3, [1]), (339, 3, [sample_format]),
        (33550, 12, [px, py, 0.0]), (33922, 12, [0.0, 0.0, 0.0, minx, maxy, 0.0]), (34735, 3, geokeys), (42113, 2, [nodata]),
    ]
    if tiled:
        tags += [(322, 4, [tw]), (323, 4, [th]), (324, off_t, None), (325, off_t, [len(s) for s in streams])]
    else:
        tags += [(278, 4, [th]), (273, off_t, None), (279, off_t, [len(s) for s in streams])]
    tags.sort(key=lambda t: t[0])
    path = f"{outfilepath}.tif"
    with open(path, "wb") as f:
        # header, then the pixel data, then the IFD (offsets known by then)
        f.write(struct.pack("<2sHHHQ", b"II", 43, 8, 0, 0) if bigtiff else struct.pack("<2sHI", b"II", 42, 0))
        offsets = []
        for s in streams:
            if f.tell() % 2:
                f.write(b"\0")
            offsets.append(f.tell())
            f.write(s)
        if f.tell() % 2:
            f.write(b"\0")

        def payload(typ, vals):
            fmt, _ = _TYPES[typ]
            if typ == 2:
                return vals[0]
            return struct.pack("<%d%s" % (len(vals), fmt), *vals)

        entries, extra = [], b""
        inline = 8 if bigtiff else 4
        ifd_pos = f.tell()
        ntags = len(tags)
        ifd_size = (8 + 20 * ntags + 8) if bigtiff else (2 + 12 * ntags + 4)
        extra_pos = ifd_pos + ifd_size
        for tag, typ, vals in tags:
            if vals is None:
                vals = offsets
            data = payload(typ, vals)
            count = len(data) if typ == 2 else len(vals)
            if len(data) <= inline:
                field = data + b"\0" * (inline - len(data))
            else:
                if (extra_pos + len(extra)) % 2:
                    extra += b"\0"
                field = struct.pack("<Q" if bigtiff else "<I", extra_pos + len(extra))
                extra += data
            entries.append(struct.pack("<HHQ" if bigtiff else "<HHI", tag, typ, count) + field)
        f.write(struct.pack("<Q" if bigtiff else "<H", ntags) + b"".join(entries) + struct.pack("<Q" if bigtiff else "<I", 0) + extra)
        f.seek(8 if bigtiff else 4)
        f.write(struct.pack("<Q" if bigtiff else "<I", ifd_pos))
    return path


def read_geotiff(path):
    """Decodes a file written by save_array_to_grid.  Returns (array (1, H, W), info dict with the GeoTIFF tags)."""
    with open(path, "rb") as f:
        buf = f.read()
    big = struct.unpack_from("<H", buf, 2)[0] == 43
    assert buf[:2] == b"II" and struct.unpack_from("<H", buf, 2)[0] in (42, 43)
    ifd = struct.unpack_from("<Q", buf, 8)[0] if big else struct.unpack_from("<I", buf, 4)[0]
    n = struct.unpack_from("<Q" if big else "<H", buf, ifd)[0]
    pos = ifd + (8 if big else 2)
    tags = {}
    for _ in range(n):
        tag, typ, count = struct.unpack_from("<HHQ" if big else "<HHI", buf, pos)
        fmt, size = _TYPES[typ]
        fpos = pos + (12 if big else 8)
        if count * size > (8 if big else 4):
            fpos = struct.unpack_from("<Q" if big else "<I", buf, fpos)[0]
        tags[tag] = buf[fpos:fpos + count] if typ == 2 else list(struct.unpack_from("<%d%s" % (count, fmt), buf, fpos))
        pos += 20 if big else 12
    W, H, bits, comp, fmtc = tags[256][0], tags[257][0], tags[258][0], tags[259][0], tags.get(339, [1])[0]
    dt = np.dtype({(16, 2): "<i2", (16, 1): "<u2", (32, 3): "<f4", (32, 2): "<i4", (8, 1): "u1", (64, 3): "<f8"}[(bits, fmtc)])
    if 322 in tags:
        tw, th, offs, cnts = tags[322][0], tags[323][0], tags[324], tags[325]
    else:
        tw, th, offs, cnts = W, tags[278][0], tags[273], tags[279]
    ny, nx = (H + th - 1) // th, (W + tw - 1) // tw
    out = np.zeros((ny * th, nx * tw), dtype=dt)
    nbytes = th * tw * dt.itemsize
    for i, (o, c) in enumerate(zip(offs, cnts)):
        ty, tx = divmod(i, nx)
        rows = th if 322 in tags else min(th, H - ty * th)  # (the last strip holds only the rows that exist)
        raw = lzw_decode(buf[o:o + c], rows * tw * dt.itemsize) if comp == 5 else np.frombuffer(buf[o:o + c], dtype=np.uint8)
        out[ty * th:ty * th + rows, tx * tw:(tx + 1) * tw] = raw.view(dt).reshape(rows, tw)
    info = {"pixel_scale": tags.get(33550), "tiepoint": tags.get(33922), "geokeys": tags.get(34735),
            "nodata": tags.get(42113, b"").rstrip(b"\0").decode(), "bigtiff": big, "compression": comp, "tile": (th, tw)}
    return out[None, :H, :W], info
