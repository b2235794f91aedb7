// Host-side codec of the DEM's output format (reference deepbedmap.py:749-756 -> data_prep.py:779-834: GeoTIFF, int16,
// tiled, compress=lzw): TIFF 6.0 LZW (MSB-first codes, ClearCode 256, EndOfInformation 257, 9..12 bits with the
// "early change" of the TIFF specification), one call per batch of tiles, tiles spread over host threads; and the
// float32 -> int16 conversion of the stitched canvas on the GPU with NumPy's astype semantics.
#include "model.h"
#include <thread>

namespace {

struct BitWriter {
  uint8_t* out;
  size_t cap, n = 0;
  uint32_t acc = 0;
  int bits = 0;
  bool overflow = false;
  void put(uint32_t code, int width) {
    acc = (acc << width) | code;
    bits += width;
    while (bits >= 8) {
      bits -= 8;
      if (n < cap) out[n] = (uint8_t)(acc >> bits); else overflow = true;
      ++n;
    }
  }
  void flush() {
    if (bits > 0) {
      if (n < cap) out[n] = (uint8_t)(acc << (8 - bits)); else overflow = true;
      ++n;
      bits = 0;
    }
  }
};

// Encoder with an open-addressing hash of (prefix code, byte) -> code.  Returns the encoded size, or 0 if `cap` is too small.
size_t lzw_encode_one(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  constexpr int HSIZE = 1 << 13;  // 8192 slots for at most 3837 entries
  static thread_local int32_t hkey[HSIZE];
  static thread_local uint16_t hval[HSIZE];
  auto reset = [&]() { for (int i = 0; i < HSIZE; ++i) hkey[i] = -1; };
  BitWriter w{dst, cap};
  int width = 9, next = 258;
  reset();
  w.put(256, width);
  if (n == 0) { w.put(257, width); w.flush(); return w.overflow ? 0 : w.n; }
  int omega = src[0];
  for (size_t i = 1; i < n; ++i) {
    const int k = src[i];
    const int32_t key = (omega << 8) | k;
    int h = (int)(((uint32_t)key * 2654435761u) >> 19) & (HSIZE - 1);
    int found = -1;
    while (hkey[h] != -1) {
      if (hkey[h] == key) { found = hval[h]; break; }
      h = (h + 1) & (HSIZE - 1);
    }
    if (found >= 0) { omega = found; continue; }
    w.put((uint32_t)omega, width);
    hkey[h] = key; hval[h] = (uint16_t)next;
    ++next;
    // TIFF 6.0 section 13 ("early change"): the code width grows as soon as table entry 511 / 1023 / 2047 has been
    // added, one code before it would be needed; when entry 4093 has been added a ClearCode is written and the table
    // starts over (the same points as libtiff's encoder, whose decoder GDAL and Pillow use)
    if (next == 4094) {
      w.put(256, width);
      reset();
      width = 9; next = 258;
    } else if (next == 512 || next == 1024 || next == 2048) {
      ++width;
    }
    omega = k;
  }
  w.put((uint32_t)omega, width);
  // the decoder adds a table entry after this code as well: the EndOfInformation code may need the wider field
  ++next;
  if (next == 4094) { w.put(256, width); width = 9; }
  else if (next == 512 || next == 1024 || next == 2048) ++width;
  w.put(257, width);
  w.flush();
  return w.overflow ? 0 : w.n;
}

// Decoder (TIFF 6.0 pseudo code).  Returns the decoded size, or (size_t)-1 on a malformed stream / overflow.
size_t lzw_decode_one(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  static thread_local uint16_t prefix[4096];
  static thread_local uint8_t suffix[4096], first[4096];
  static thread_local uint8_t stack[4096];
  size_t pos = 0, outn = 0;
  uint32_t acc = 0;
  int bits = 0, width = 9, next = 258, old = -1;
  auto get = [&]() -> int {
    while (bits < width) {
      if (pos >= n) return -1;
      acc = (acc << 8) | src[pos++];
      bits += 8;
    }
    bits -= width;
    return (int)((acc >> bits) & ((1u << width) - 1));
  };
  for (;;) {
    const int code = get();
    if (code < 0 || code == 257) break;
    if (code == 256) { width = 9; next = 258; old = -1; continue; }
    int cur = code, sp = 0;
    if (old < 0) {
      if (code > 255) return (size_t)-1;
      if (outn >= cap) return (size_t)-1;
      dst[outn++] = (uint8_t)code;
      old = code;
      continue;
    }
    if (code >= next) {  // KwKwK
      if (code != next) return (size_t)-1;
      stack[sp++] = old < 256 ? (uint8_t)old : first[old];
      cur = old;
    }
    while (cur >= 256) { stack[sp++] = suffix[cur]; cur = prefix[cur]; if (sp >= 4095) return (size_t)-1; }
    stack[sp++] = (uint8_t)cur;
    const uint8_t f = (uint8_t)cur;
    if (outn + sp > cap) return (size_t)-1;
    while (sp) dst[outn++] = stack[--sp];
    if (next < 4096) {
      prefix[next] = (uint16_t)old; suffix[next] = f; first[next] = old < 256 ? (uint8_t)old : first[old];
      ++next;
      if (next == 511 || next == 1023 || next == 2047) ++width;
    }
    old = code;
  }
  return outn;
}

__global__ __launch_bounds__(256) void f32_to_i16_kernel(const float* __restrict__ src, short* __restrict__ dst, long n) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const float x = src[e];
    // numpy.ndarray.astype(int16) on x86-64: truncation to int32 (cvttss2si: NaN and |x| >= 2^31 give INT32_MIN), then
    // the low 16 bits -- NaN (the canvas frame that no tile covers), +-inf and out-of-range values become 0
    const int v = (x == x && x > -2147483648.f && x < 2147483648.f) ? (int)x : (int)0x80000000;
    dst[e] = (short)(v & 0xffff);
  }
}

}  // namespace

extern "C" {

int dbm_lzw_encode_tiles(const void* tiles, size_t tile_bytes, int ntiles, void* out, size_t out_stride, size_t* out_sizes,
                         int nthreads) {
  if (!tiles || !out || !out_sizes || ntiles < 0) return 1;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > ntiles) nthreads = ntiles > 0 ? ntiles : 1;
  auto work = [&](int t0) {
    for (int t = t0; t < ntiles; t += nthreads)
      out_sizes[t] = lzw_encode_one((const uint8_t*)tiles + (size_t)t * tile_bytes, tile_bytes, (uint8_t*)out + (size_t)t * out_stride, out_stride);
  };
  std::vector<std::thread> th;
  for (int i = 1; i < nthreads; ++i) th.emplace_back(work, i);
  work(0);
  for (auto& t : th) t.join();
  for (int t = 0; t < ntiles; ++t)
    if (out_sizes[t] == 0) return 2;  // out_stride too small for an incompressible tile (needs ~ 1.41 x tile_bytes)
  return 0;
}

int dbm_lzw_decode(const void* src, size_t nbytes, void* dst, size_t cap, size_t* out_bytes) {
  if (!src || !dst || !out_bytes) return 1;
  const size_t n = lzw_decode_one((const uint8_t*)src, nbytes, (uint8_t*)dst, cap);
  if (n == (size_t)-1) return 2;
  *out_bytes = n;
  return 0;
}

int dbm_f32_to_i16(dbm_ctx* ctx, const float* src_dev, void* dst_dev, size_t n) {
  if (!ctx) return 1;
  if (n == 0) return 0;
  long blocks = ((long)n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(f32_to_i16_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, src_dev, (short*)dst_dev, (long)n);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // extern "C"
