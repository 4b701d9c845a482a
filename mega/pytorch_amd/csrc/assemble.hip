// Pool / key-set assembly for the relation aggregation: many small 2-D block copies in ONE launch.
//
// The reference re-concatenates its pools with torch.cat on every key frame (roi_box_feature_extractors.py:676,
// :687-688, :812-814; detector/generalized_rcnn_mega.py:213-216).  The batched aggregation lays the windows, memory pools
// and key sets of a whole step-batch out as a few "tapes" -- but each tape was still one torch.cat launch (53 per
// step-batch, 5-8 us of GPU time each, the largest launch family of the aggregation).  Here every concatenation is a
// list of (source block -> destination block) segments, and all segments of all concatenations that do not depend on
// each other travel in one kernel argument: one launch instead of 3-6.
//
// A segment is a [rows][row_bytes] byte block with independent source / destination row strides, so both row
// concatenation (dim 0) and column concatenation (dim 1: the key-contiguous V^T blocks, whose 150-byte rows are only
// 2-byte aligned) are the same thing.  Since the end of round 4 all segments of a call go out in ONE launch of
// copy_any_kernel (16-byte units at whatever alignment a segment has); the per-width form before it (MEGA_COPY_ANY=0: one
// launch per copy width 16 / 8 / 4 / 2 bytes, the largest that divides a segment's addresses, strides and row length) is
// kept for A/B.  Pure data movement, bit-exact by construction.
#include "common.h"

namespace {

constexpr int COPY_MAXSEG = 56;

struct CopySeg {
  const unsigned char* src;
  unsigned char* dst;
  long long src_stride, dst_stride;   // bytes
  int rows, units_per_row;            // units of W bytes
  int row_bytes, reserved;            // (copy_any_kernel: units of 16 bytes, the last one of a row may be shorter)
};
struct CopyBatch {
  int n;
  unsigned ubase[COPY_MAXSEG + 1];    // first unit of each segment in the launch's flat unit index
  CopySeg s[COPY_MAXSEG];
};

template <typename V>
__global__ __launch_bounds__(256) void copy_segments_kernel(CopyBatch b) {
  const unsigned total = b.ubase[b.n];
  for (unsigned unit = blockIdx.x * 256u + threadIdx.x; unit < total; unit += gridDim.x * 256u) {
    int lo = 0, hi = b.n;               // segment of this unit: ubase[lo] <= unit < ubase[lo + 1]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (b.ubase[mid] <= unit) lo = mid; else hi = mid;
    }
    const CopySeg& s = b.s[lo];
    const unsigned u = unit - b.ubase[lo];
    const unsigned r = u / (unsigned)s.units_per_row, c = u - r * (unsigned)s.units_per_row;
    *reinterpret_cast<V*>(s.dst + (long long)r * s.dst_stride + (size_t)c * sizeof(V)) =
        *reinterpret_cast<const V*>(s.src + (long long)r * s.src_stride + (size_t)c * sizeof(V));
  }
}

// The same for ANY element-aligned geometry in one launch: a row is cut into 16-byte units from its first byte, whatever
// the addresses -- gfx950 serves 16-byte global loads AND stores at 2-byte-aligned addresses correctly, at ~90 % of the
// aligned rate (tools/probes/unaligned_load.hip) -- and the last unit of a row carries its remaining bytes in 2-byte (and at
// most one 1-byte) pieces.  Round 4: before this, every copy width that occurred in a call was its own launch, and the
// 150-byte V^T column blocks of the memory tapes (75 keys) went through torch.cat's element-wise kernel.
__global__ __launch_bounds__(256) void copy_any_kernel(CopyBatch b) {
  const unsigned total = b.ubase[b.n];
  for (unsigned unit = blockIdx.x * 256u + threadIdx.x; unit < total; unit += gridDim.x * 256u) {
    int lo = 0, hi = b.n;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (b.ubase[mid] <= unit) lo = mid; else hi = mid;
    }
    const CopySeg& s = b.s[lo];
    const unsigned u = unit - b.ubase[lo];
    const unsigned r = u / (unsigned)s.units_per_row, c = u - r * (unsigned)s.units_per_row;
    const unsigned char* sp = s.src + (long long)r * s.src_stride + (size_t)c * 16;
    unsigned char* dp = s.dst + (long long)r * s.dst_stride + (size_t)c * 16;
    const int nb = s.row_bytes - (int)c * 16;
    if (nb >= 16) {
      *reinterpret_cast<u32x4_t*>(dp) = *reinterpret_cast<const u32x4_t*>(sp);
    } else {
      int i = 0;
      for (; i + 2 <= nb; i += 2) *reinterpret_cast<unsigned short*>(dp + i) = *reinterpret_cast<const unsigned short*>(sp + i);
      if (i < nb) dp[i] = sp[i];
    }
  }
}

// f32 -> bf16 (round to nearest even, the hardware conversion every epilogue uses), 8 elements per thread
template <typename HT>      // bf16_t, or f16_t (IEEE half)
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n8,
                                                            size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
    u32x4_t o;
    o[0] = Half16<HT>::pack2(a.x, a.y); o[1] = Half16<HT>::pack2(a.z, a.w); o[2] = Half16<HT>::pack2(b.x, b.y); o[3] = Half16<HT>::pack2(b.z, b.w);
    reinterpret_cast<u32x4_t*>(dst)[i] = o;
  }
  if (blockIdx.x == 0)       // tail (n not a multiple of 8)
    for (size_t i = n8 * 8 + threadIdx.x; i < n; i += 256) dst[i] = Half16<HT>::cvt(src[i]);
}

// copy_segments with an f32 -> bf16 conversion on the way: a concatenation of f32 row blocks delivered as bf16 (the key / value
// sources of the relation modules when the head's activation stream is f32: only their rounded copy is ever read).  A segment
// is [rows][8 * units_per_row] f32 elements; each thread converts 8 elements (32 B in, 16 B out).
template <typename HT>
__global__ __launch_bounds__(256) void copy_cast_segments_kernel(CopyBatch b) {
  const unsigned total = b.ubase[b.n];
  for (unsigned unit = blockIdx.x * 256u + threadIdx.x; unit < total; unit += gridDim.x * 256u) {
    int lo = 0, hi = b.n;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (b.ubase[mid] <= unit) lo = mid; else hi = mid;
    }
    const CopySeg& s = b.s[lo];
    const unsigned u = unit - b.ubase[lo];
    const unsigned r = u / (unsigned)s.units_per_row, c = u - r * (unsigned)s.units_per_row;
    const float4* src = reinterpret_cast<const float4*>(s.src + (long long)r * s.src_stride + (size_t)c * 32);
    const float4 a = src[0], q = src[1];
    u32x4_t o;
    o[0] = Half16<HT>::pack2(a.x, a.y); o[1] = Half16<HT>::pack2(a.z, a.w); o[2] = Half16<HT>::pack2(q.x, q.y); o[3] = Half16<HT>::pack2(q.z, q.w);
    *reinterpret_cast<u32x4_t*>(s.dst + (long long)r * s.dst_stride + (size_t)c * 16) = o;
  }
}

// f32 row [K] -> bf16 row [3K] = [hi | lo | hi], hi = bf16(v), lo = bf16(v - hi): with the weight rows laid out
// [Wh | Wh | Wl] a plain bf16 GEMM over 3K computes hi.Wh + lo.Wh + hi.Wl = v.W to ~2^-16 (f32 accumulation of exact
// bf16 products; only the lo.Wl term, 2^-18, is dropped) at the bf16 MFMA rate.  8 elements per thread.
template <int PLANES, typename HT = bf16_t>
__global__ __launch_bounds__(256) void split3_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int rows,
                                                              int K8) {
  const size_t total = (size_t)rows * K8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / K8;
    const int c = (int)(i - r * K8);
    const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    u32x4_t hi, lo;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const unsigned h = Half16<HT>::pack2(v[2 * d], v[2 * d + 1]);
      hi[d] = h;
      lo[d] = Half16<HT>::pack2(v[2 * d] - Half16<HT>::lo(h), v[2 * d + 1] - Half16<HT>::hi(h));
    }
    u32x4_t* row = reinterpret_cast<u32x4_t*>(dst + r * (size_t)(K8 * 8 * PLANES));
    row[c] = hi;
    row[K8 + c] = lo;
    if (PLANES == 3) row[2 * K8 + c] = hi;
  }
}

}  // namespace

// dst[r][0:K] = hi, dst[r][K:2K] = lo, dst[r][2K:3K] = hi of src[r][0:K] (f32, K a multiple of 8, 16-byte aligned): the A
// operand of a split-precision bf16 GEMM against weights [Wh | Wh | Wl] (the stage FCs of the head's f32 activation
// stream, roi_box_feature_extractors.py:826-827).
extern "C" int mega_split_f32_to_bf16x3(const float* src, void* dst, int rows, int K, void* stream) {
  mega_clear_error();
  if (rows == 0) return MEGA_OK;
  if (!src || !dst || rows < 0 || K <= 0 || K % 8 || (reinterpret_cast<size_t>(src) & 15) || (reinterpret_cast<size_t>(dst) & 15))
    return MEGA_ERR_ARG;
  const size_t total = (size_t)rows * (K / 8);
  size_t nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(split3_f32_bf16_kernel<3>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, rows, K / 8);
  return mega_check_launch();
}

// dst[r][0:K] = hi, dst[r][K:2K] = lo of src[r][0:K]: the split-precision PLANES form of an f32 activation (hi = bf16(x),
// lo = bf16(x - hi)) that mega_conv2d_nhwc_sp reads as its input / residual and writes as its output.
extern "C" int mega_split_f32_to_planes_dt(const float* src, void* dst, int rows, int K, int dtype, void* stream) {
  mega_clear_error();
  if (rows == 0) return MEGA_OK;
  if (dtype != MEGA_BF16 && dtype != MEGA_F16) return MEGA_ERR_ARG;
  if (!src || !dst || rows < 0 || K <= 0 || K % 8 || (reinterpret_cast<size_t>(src) & 15) || (reinterpret_cast<size_t>(dst) & 15))
    return MEGA_ERR_ARG;
  const size_t total = (size_t)rows * (K / 8);
  size_t nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  if (dtype == MEGA_F16)
    hipLaunchKernelGGL((split3_f32_bf16_kernel<2, f16_t>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, rows, K / 8);
  else
    hipLaunchKernelGGL((split3_f32_bf16_kernel<2, bf16_t>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, rows, K / 8);
  return mega_check_launch();
}

extern "C" int mega_split_f32_to_planes(const float* src, void* dst, int rows, int K, void* stream) {
  return mega_split_f32_to_planes_dt(src, dst, rows, K, MEGA_BF16, stream);
}

// dst[i] = bf16(src[i]) for n contiguous elements (both 16-byte aligned).  The aggregation head keeps its activation
// stream in f32 (cfg.HEAD_STREAM) and feeds the bf16 projections (Wq / Wk / Wv) a rounded copy: the rounding the bf16
// GEMM's A operand needs anyway, without rounding the stream itself.
extern "C" int mega_cast_f32_to_half(const float* src, void* dst, size_t n, int dtype, void* stream) {
  mega_clear_error();
  if (n == 0) return MEGA_OK;
  if (dtype != MEGA_BF16 && dtype != MEGA_F16) return MEGA_ERR_ARG;
  if (!src || !dst || (reinterpret_cast<size_t>(src) & 15) || (reinterpret_cast<size_t>(dst) & 15)) return MEGA_ERR_ARG;
  const size_t n8 = n / 8;
  size_t nb = (n8 + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  if (dtype == MEGA_F16)
    hipLaunchKernelGGL(cast_f32_bf16_kernel<f16_t>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n8, n);
  else
    hipLaunchKernelGGL(cast_f32_bf16_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n8, n);
  return mega_check_launch();
}

extern "C" int mega_cast_f32_to_bf16(const float* src, void* dst, size_t n, void* stream) {
  return mega_cast_f32_to_half(src, dst, n, MEGA_BF16, stream);
}

struct MegaCopySegC {
  const void* src; void* dst; long long src_stride, dst_stride; int rows, row_bytes;
};

// segs[n] as for mega_copy_segments, but the source blocks are f32 and the destination blocks bf16: row_bytes counts the
// SOURCE bytes of a row (a multiple of 32: 8 elements per thread), strides are bytes of the respective tensor, both sides
// 16-byte aligned.  dst = bf16(src), round to nearest even.
extern "C" int mega_copy_cast_segments_dt(const void* segs, int n, int dtype, void* stream) {
  mega_clear_error();
  if (n == 0) return MEGA_OK;
  if (!segs || n < 0 || (dtype != MEGA_BF16 && dtype != MEGA_F16)) return MEGA_ERR_ARG;
  const MegaCopySegC* d = (const MegaCopySegC*)segs;
  for (int i = 0; i < n; ++i) {
    const MegaCopySegC& g = d[i];
    if (g.rows < 0 || g.row_bytes < 0 || (g.row_bytes & 31)) return MEGA_ERR_ARG;
    if (g.rows > 0 && g.row_bytes > 0) {
      if (!g.src || !g.dst || ((size_t)g.src & 15) || ((size_t)g.dst & 15)) return MEGA_ERR_ARG;
      if (g.rows > 1 && ((g.src_stride & 15) || (g.dst_stride & 15))) return MEGA_ERR_ARG;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  CopyBatch b;
  b.n = 0;
  unsigned long long units = 0;
  auto flush = [&]() {
    if (b.n == 0) return;
    b.ubase[b.n] = (unsigned)units;
    unsigned long long nb = (units + 1023) / 1024;
    if (nb > 2048) nb = 2048;
    if (dtype == MEGA_F16) hipLaunchKernelGGL(copy_cast_segments_kernel<f16_t>, dim3((unsigned)(nb < 1 ? 1 : nb)), dim3(256), 0, st, b);
    else hipLaunchKernelGGL(copy_cast_segments_kernel<bf16_t>, dim3((unsigned)(nb < 1 ? 1 : nb)), dim3(256), 0, st, b);
    b.n = 0;
    units = 0;
  };
  for (int i = 0; i < n; ++i) {
    const MegaCopySegC& g = d[i];
    if (g.rows == 0 || g.row_bytes == 0) continue;
    const unsigned long long u = (unsigned long long)g.rows * (unsigned long long)(g.row_bytes / 32);
    if (u >= 0xFFFFFFFFull) return MEGA_ERR_ARG;
    if (b.n == COPY_MAXSEG || units + u >= 0xFFFFFFFFull) flush();
    CopySeg& sg = b.s[b.n];
    sg.src = (const unsigned char*)g.src; sg.dst = (unsigned char*)g.dst;
    sg.src_stride = g.src_stride; sg.dst_stride = g.dst_stride;
    sg.rows = g.rows; sg.units_per_row = (int)(g.row_bytes / 32);
    b.ubase[b.n] = (unsigned)units;
    units += u;
    ++b.n;
  }
  flush();
  return mega_check_launch();
}

extern "C" int mega_copy_cast_segments(const void* segs, int n, void* stream) {
  return mega_copy_cast_segments_dt(segs, n, MEGA_BF16, stream);
}

// segs[n]: copy rows x row_bytes bytes from src (+ r * src_stride) to dst (+ r * dst_stride).  Segments must not
// overlap each other's destinations.  Any n (launched in groups of 56 segments).
extern "C" int mega_copy_segments(const void* segs, int n, void* stream) {
  mega_clear_error();
  if (n == 0) return MEGA_OK;
  if (!segs || n < 0) return MEGA_ERR_ARG;
  const MegaCopySegC* d = (const MegaCopySegC*)segs;
  hipStream_t st = (hipStream_t)stream;
  auto seg_align = [](const MegaCopySegC& g) -> unsigned long long {
    unsigned long long bits = (unsigned long long)(size_t)g.src | (unsigned long long)(size_t)g.dst |
                              (unsigned long long)g.row_bytes;
    if (g.rows > 1) bits |= (unsigned long long)g.src_stride | (unsigned long long)g.dst_stride;
    unsigned long long a = 16;
    while (a > 1 && (bits & (a - 1))) a >>= 1;
    return a;
  };
  for (int i = 0; i < n; ++i) {
    const MegaCopySegC& g = d[i];
    if (g.rows < 0 || g.row_bytes < 0 || (g.rows > 0 && g.row_bytes > 0 && (!g.src || !g.dst))) return MEGA_ERR_ARG;
    if (g.rows > 0 && g.row_bytes > 0 && seg_align(g) < 2 && getenv("MEGA_COPY_ANY") != nullptr && getenv("MEGA_COPY_ANY")[0] == '0')
      return MEGA_ERR_ARG;   // (the per-width form: bf16 / f32 / i32 data, >= 2 bytes)
  }
  static const bool any_form = !(getenv("MEGA_COPY_ANY") != nullptr && getenv("MEGA_COPY_ANY")[0] == '0');
  if (any_form) {
    // all segments in one launch (groups of COPY_MAXSEG), 16-byte units at whatever alignment the segment has
    CopyBatch b;
    b.n = 0;
    unsigned long long units = 0;
    auto flush = [&]() {
      if (b.n == 0) return;
      b.ubase[b.n] = (unsigned)units;
      unsigned long long nb = (units + 1023) / 1024;        // ~4 units per thread
      if (nb > 2048) nb = 2048;
      hipLaunchKernelGGL(copy_any_kernel, dim3((unsigned)(nb < 1 ? 1 : nb)), dim3(256), 0, st, b);
      b.n = 0;
      units = 0;
    };
    for (int i = 0; i < n; ++i) {
      const MegaCopySegC& g = d[i];
      if (g.rows == 0 || g.row_bytes == 0) continue;
      if (g.row_bytes >= (1ll << 31)) return MEGA_ERR_ARG;
      const unsigned long long upr = (unsigned long long)(g.row_bytes + 15) / 16;
      const unsigned long long u = (unsigned long long)g.rows * upr;
      if (u >= 0xFFFFFFFFull) return MEGA_ERR_ARG;
      if (b.n == COPY_MAXSEG || units + u >= 0xFFFFFFFFull) flush();
      CopySeg& sg = b.s[b.n];
      sg.src = (const unsigned char*)g.src; sg.dst = (unsigned char*)g.dst;
      sg.src_stride = g.src_stride; sg.dst_stride = g.dst_stride;
      sg.rows = g.rows; sg.units_per_row = (int)upr; sg.row_bytes = (int)g.row_bytes; sg.reserved = 0;
      b.ubase[b.n] = (unsigned)units;
      units += u;
      ++b.n;
    }
    flush();
    return mega_check_launch();
  }
  // (MEGA_COPY_ANY=0, the form before round 4's end) one launch per copy width that occurs (a 2-byte-aligned V^T column
  // block must not drag the 16-byte-aligned row blocks of the same call down to 2 bytes per thread), in groups of
  // COPY_MAXSEG segments
  for (unsigned long long align = 16; align >= 2; align >>= 1) {
    CopyBatch b;
    b.n = 0;
    unsigned long long units = 0;
    auto flush = [&]() {
      if (b.n == 0) return;
      b.ubase[b.n] = (unsigned)units;
      unsigned long long nb = (units + 1023) / 1024;        // ~4 units per thread
      if (nb > 2048) nb = 2048;
      const dim3 grid((unsigned)(nb < 1 ? 1 : nb));
      if (align == 16) hipLaunchKernelGGL((copy_segments_kernel<uint4>), grid, dim3(256), 0, st, b);
      else if (align == 8) hipLaunchKernelGGL((copy_segments_kernel<uint2>), grid, dim3(256), 0, st, b);
      else if (align == 4) hipLaunchKernelGGL((copy_segments_kernel<unsigned>), grid, dim3(256), 0, st, b);
      else hipLaunchKernelGGL((copy_segments_kernel<unsigned short>), grid, dim3(256), 0, st, b);
      b.n = 0;
      units = 0;
    };
    for (int i = 0; i < n; ++i) {
      const MegaCopySegC& g = d[i];
      if (g.rows == 0 || g.row_bytes == 0 || seg_align(g) != align) continue;
      const unsigned long long u = (unsigned long long)g.rows * (unsigned long long)(g.row_bytes / align);
      if (u >= 0xFFFFFFFFull) return MEGA_ERR_ARG;
      if (b.n == COPY_MAXSEG || units + u >= 0xFFFFFFFFull) flush();
      CopySeg& sg = b.s[b.n];
      sg.src = (const unsigned char*)g.src; sg.dst = (unsigned char*)g.dst;
      sg.src_stride = g.src_stride; sg.dst_stride = g.dst_stride;
      sg.rows = g.rows; sg.units_per_row = (int)(g.row_bytes / align);
      b.ubase[b.n] = (unsigned)units;
      units += u;
      ++b.n;
    }
    flush();
  }
  return mega_check_launch();
}
