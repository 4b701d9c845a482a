// Spatial (non-GEMM) kernels of the backbone / box head:
//   * stem: 7x7 stride-2 conv (3 -> 64) + FrozenBN + ReLU      (mega_core/modeling/backbone/resnet.py:347-366)
//   * 3x3 stride-2 max-pool                                      (resnet.py:365)
//   * ROIAlign forward                                           (mega_core/csrc/cuda/ROIAlign_cuda.cu:15-122,
//                                                                 csrc/cpu/ROIAlign_cpu.cpp:18-219)
// Activations are NHWC so that a pixel's channels are one contiguous, coalesced run.
#include <cstdlib>

#include "common.h"

namespace {

// ------------------------------------------------------------------------------------ stem conv
// in : NCHW f32 [N][3][H][W] (the reference's image tensor layout)
// w  : [147][64] f32, tap = (c*7 + r)*7 + s   (host repack of OIHW [64][3][7][7])
// out: NHWC [N][Ho][Wo][64]
constexpr int ST_T = 16;                      // output tile edge
constexpr int ST_P = ST_T * 2 + 5;            // 37 input rows/cols per tile
constexpr int ST_PW = ST_P + 1;               // padded row length

template <typename OT>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ bias, OT* __restrict__ out, int N,
                                                        int H, int W, int Ho, int Wo) {
  __shared__ float patch[3][ST_P][ST_PW];
  const int tid = threadIdx.x;
  const int n = blockIdx.z;
  const int oy0 = blockIdx.y * ST_T, ox0 = blockIdx.x * ST_T;
  const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
  for (int e = tid; e < 3 * ST_P * ST_P; e += 256) {
    const int c = e / (ST_P * ST_P);
    const int rem = e - c * ST_P * ST_P;
    const int y = rem / ST_P, x = rem - y * ST_P;
    const int iy = iy0 + y, ix = ix0 + x;
    float v = 0.f;
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = in[((size_t)(n * 3 + c) * H + iy) * W + ix];
    patch[c][y][x] = v;
  }
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;
  const int oy = oy0 + ty, ox = ox0 + tx;
  float acc[64];
#pragma unroll
  for (int k = 0; k < 64; ++k) acc[k] = 0.f;
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 7; ++r) {
#pragma unroll
      for (int s = 0; s < 7; ++s) {
        const float v = patch[c][2 * ty + r][2 * tx + s];
        const float* wp = w + ((c * 7 + r) * 7 + s) * 64;  // wave-uniform address -> scalar loads
#pragma unroll
        for (int k = 0; k < 64; ++k) acc[k] = fmaf(v, wp[k], acc[k]);
      }
    }
  if (oy < Ho && ox < Wo) {
    OT* o = out + (((size_t)n * Ho + oy) * Wo + ox) * 64;
#pragma unroll
    for (int k = 0; k < 64; ++k) {
      float v = acc[k] * scale[k] + bias[k];
      v = fmaxf(v, 0.f);
      Elem<OT>::st(o + k, v);
    }
  }
}

// ------------------------------------------------------------------------------------ max-pool 3x3 s2 p1
template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H,
                                                      int W, int C, int Ho, int Wo) {
  constexpr int VE = Elem<T>::VE;
  const int cv = C / VE;
  const size_t total = (size_t)N * Ho * Wo * cv;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int v = (int)(idx % cv);
    size_t pix = idx / cv;
    const int wo = (int)(pix % Wo);
    pix /= Wo;
    const int ho = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    float m[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) m[e] = -INFINITY;
    for (int dy = 0; dy < 3; ++dy) {
      const int hi = ho * 2 - 1 + dy;
      if ((unsigned)hi >= (unsigned)H) continue;
      for (int dx = 0; dx < 3; ++dx) {
        const int wi = wo * 2 - 1 + dx;
        if ((unsigned)wi >= (unsigned)W) continue;
        const uint4 raw = *reinterpret_cast<const uint4*>(in + (((size_t)n * H + hi) * W + wi) * C + v * VE);
        const T* rv = reinterpret_cast<const T*>(&raw);
#pragma unroll
        for (int e = 0; e < VE; ++e) m[e] = fmaxf(m[e], Elem<T>::ld(rv + e));
      }
    }
    uint4 o;
    T* ov = reinterpret_cast<T*>(&o);
#pragma unroll
    for (int e = 0; e < VE; ++e) Elem<T>::st(ov + e, m[e]);
    *reinterpret_cast<uint4*>(out + (((size_t)n * Ho + ho) * Wo + wo) * C + v * VE) = o;
  }
}

// ------------------------------------------------------------------------------------ ROIAlign forward
// feat: NHWC [B][H][W][C] (in_nhwc=1) or NCHW [B][C][H][W] (in_nhwc=0)
// rois: [K][5] f32 = (batch_idx, x1, y1, x2, y2)
// out : bin-major [K][ph*pw][C] (out_nhwc=1; fc weights are permuted to match) or the reference's
//       [K][C][ph][pw] (out_nhwc=0).
// One block per (roi, ph, pw) bin; threads stride over channels.  The sample geometry is block-uniform.
// Arithmetic follows ROIAlign_cuda.cu:64-122 term by term: no rounding of the scaled ROI, min size 1,
// adaptive grid ceil(roi/pooled) when sampling_ratio <= 0, bilinear weights hy*hx.., mean over samples.
template <typename T, typename OT>
__global__ __launch_bounds__(256) void roi_align_kernel(const T* __restrict__ feat, const float* __restrict__ rois,
                                                        OT* __restrict__ out, int K, int C, int H, int W,
                                                        float spatial_scale, int PH, int PW, int sampling_ratio,
                                                        int in_nhwc, int out_nhwc) {
  const int bin = blockIdx.x;
  const int pw = bin % PW;
  const int ph = (bin / PW) % PH;
  const int k = bin / (PW * PH);
  const float* roi = rois + (size_t)k * 5;
  const int b = (int)roi[0];
  const float roi_start_w = roi[1] * spatial_scale;
  const float roi_start_h = roi[2] * spatial_scale;
  const float roi_end_w = roi[3] * spatial_scale;
  const float roi_end_h = roi[4] * spatial_scale;
  const float roi_width = fmaxf(roi_end_w - roi_start_w, 1.f);
  const float roi_height = fmaxf(roi_end_h - roi_start_h, 1.f);
  const float bin_size_h = roi_height / (float)PH;
  const float bin_size_w = roi_width / (float)PW;
  const int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)PH);
  const int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)PW);
  const float count = (float)(grid_h * grid_w);

  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float output_val = 0.f;
    for (int iy = 0; iy < grid_h; ++iy) {
      const float y0 = roi_start_h + ph * bin_size_h + (iy + .5f) * bin_size_h / (float)grid_h;
      for (int ix = 0; ix < grid_w; ++ix) {
        const float x0 = roi_start_w + pw * bin_size_w + (ix + .5f) * bin_size_w / (float)grid_w;
        float y = y0, x = x0;
        if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;  // contributes 0
        if (y <= 0.f) y = 0.f;
        if (x <= 0.f) x = 0.f;
        int y_low = (int)y, x_low = (int)x, y_high, x_high;
        if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else { y_high = y_low + 1; }
        if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else { x_high = x_low + 1; }
        const float ly = y - y_low, lx = x - x_low, hy = 1.f - ly, hx = 1.f - lx;
        float v1, v2, v3, v4;
        if (in_nhwc) {
          const T* base = feat + (size_t)b * H * W * C + c;
          v1 = Elem<T>::ld(base + ((size_t)y_low * W + x_low) * C);
          v2 = Elem<T>::ld(base + ((size_t)y_low * W + x_high) * C);
          v3 = Elem<T>::ld(base + ((size_t)y_high * W + x_low) * C);
          v4 = Elem<T>::ld(base + ((size_t)y_high * W + x_high) * C);
        } else {
          const T* base = feat + ((size_t)b * C + c) * H * W;
          v1 = Elem<T>::ld(base + y_low * W + x_low);
          v2 = Elem<T>::ld(base + y_low * W + x_high);
          v3 = Elem<T>::ld(base + y_high * W + x_low);
          v4 = Elem<T>::ld(base + y_high * W + x_high);
        }
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        output_val += (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
      }
    }
    output_val /= count;
    if (out_nhwc)
      Elem<OT>::st(out + ((size_t)k * PH * PW + ph * PW + pw) * C + c, output_val);
    else
      Elem<OT>::st(out + (((size_t)k * C + c) * PH + ph) * PW + pw, output_val);
  }
}

// ------------------------------------------------------------------------------------ avg-pool 2x2 s2, ceil_mode
// nn.AvgPool2d(2, stride=2, ceil_mode=True) of FlowNetS (mega_core/modeling/backbone/flownet.py:52,:56,:112) on
// NHWC: windows hanging over the edge average the in-bounds taps only (count_include_pad semantics with pad 0).
template <typename T>
__global__ __launch_bounds__(256) void avgpool2_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H,
                                                       int W, int C, int Ho, int Wo) {
  constexpr int VE = Elem<T>::VE;
  const int cv = C / VE;
  const size_t total = (size_t)N * Ho * Wo * cv;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int v = (int)(idx % cv);
    size_t pix = idx / cv;
    const int wo = (int)(pix % Wo);
    pix /= Wo;
    const int ho = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
    int cnt = 0;
    for (int dy = 0; dy < 2; ++dy) {
      const int hi = ho * 2 + dy;
      if (hi >= H) continue;
      for (int dx = 0; dx < 2; ++dx) {
        const int wi = wo * 2 + dx;
        if (wi >= W) continue;
        const uint4 raw = *reinterpret_cast<const uint4*>(in + (((size_t)n * H + hi) * W + wi) * C + v * VE);
        const T* rv = reinterpret_cast<const T*>(&raw);
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[e] += Elem<T>::ld(rv + e);
        ++cnt;
      }
    }
    uint4 o;
    T* ov = reinterpret_cast<T*>(&o);
#pragma unroll
    for (int e = 0; e < VE; ++e) Elem<T>::st(ov + e, acc[e] / (float)cnt);
    *reinterpret_cast<uint4*>(out + (((size_t)n * Ho + ho) * Wo + wo) * C + v * VE) = o;
  }
}

// Channel-vectorised NHWC -> bin-major variant (the hot-path layout): one work item = (bin, 16-byte channel
// vector); a wave reads 64 consecutive 16-B vectors of one pixel = 1 KiB fully coalesced per neighbour.
// Same arithmetic, same order of operations as roi_align_kernel.
//
// XCD_SLICED: the eight XCDs have private L2s and consecutive blocks go to consecutive XCDs.  In the plain item order
// every XCD ends up reading every channel of every map (PMC: 3.5 GB fetched for 196 MB of maps at 20 frames: the
// overlapping bilinear taps of neighbouring bins / ROIs miss L2).  Sliced, XCD x (= blockIdx.x & 7) owns the channel
// slice [x C/8, (x+1) C/8): its working set per frame is 1/8 of the map (1.2 MB for 2048 channels: L2-resident), a
// block covers 256 / (CV/8) consecutive bins of that slice.  Same arithmetic, same results.
// PLANES (f32 input only): the pooled values leave as split-precision planes -- row k of `out` is bf16 [hi | lo] of the
// [PH PW C] f32 values (hi = bf16(v), lo = bf16(v - hi)): the operand mega_conv2d_nhwc_sp's fc0 reads, written here instead
// of an f32 tensor that a second pass would have to read back and split (3 + 3 GB per 40-frame batch at C = 2048).
// PT: the 16-bit type of the planes (bf16_t, or f16_t for the fp16 two-pass mode)
template <typename T, bool XCD_SLICED, bool PLANES = false, typename PT = bf16_t>
__global__ __launch_bounds__(256) void roi_align_nhwc_vec_kernel(const T* __restrict__ feat,
                                                                 const float* __restrict__ rois, T* __restrict__ out,
                                                                 int K, int C, int H, int W, float spatial_scale,
                                                                 int PH, int PW, int sampling_ratio) {
  constexpr int VE = Elem<T>::VE;
  const int CV = C / VE;
  const long long total = (long long)K * PH * PW * CV;
  const long long nitems = XCD_SLICED ? total / 8 : total;            // per XCD when sliced
  const long long first = XCD_SLICED ? (long long)(blockIdx.x >> 3) * blockDim.x + threadIdx.x
                                     : (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long step = XCD_SLICED ? (long long)(gridDim.x >> 3) * blockDim.x : (long long)gridDim.x * blockDim.x;
  for (long long item = first; item < nitems; item += step) {
    int cv, bin;
    if (XCD_SLICED) {
      const int cvs = CV >> 3;                                         // vectors per slice
      cv = (int)(blockIdx.x & 7) * cvs + (int)(item % cvs);
      bin = (int)(item / cvs);
    } else {
      cv = (int)(item % CV);
      bin = (int)(item / CV);
    }
    const int pw = bin % PW;
    const int ph = (bin / PW) % PH;
    const int k = bin / (PW * PH);
    const float* roi = rois + (size_t)k * 5;
    const int b = (int)roi[0];
    const float roi_start_w = roi[1] * spatial_scale;
    const float roi_start_h = roi[2] * spatial_scale;
    const float roi_end_w = roi[3] * spatial_scale;
    const float roi_end_h = roi[4] * spatial_scale;
    const float roi_width = fmaxf(roi_end_w - roi_start_w, 1.f);
    const float roi_height = fmaxf(roi_end_h - roi_start_h, 1.f);
    const float bin_size_h = roi_height / (float)PH;
    const float bin_size_w = roi_width / (float)PW;
    const int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)PH);
    const int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)PW);
    const float count = (float)(grid_h * grid_w);
    const T* base = feat + (size_t)b * H * W * C + (size_t)cv * VE;
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
    for (int iy = 0; iy < grid_h; ++iy) {
      const float y0 = roi_start_h + ph * bin_size_h + (iy + .5f) * bin_size_h / (float)grid_h;
      for (int ix = 0; ix < grid_w; ++ix) {
        const float x0 = roi_start_w + pw * bin_size_w + (ix + .5f) * bin_size_w / (float)grid_w;
        float y = y0, x = x0;
        if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
        if (y <= 0.f) y = 0.f;
        if (x <= 0.f) x = 0.f;
        int y_low = (int)y, x_low = (int)x, y_high, x_high;
        if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else { y_high = y_low + 1; }
        if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else { x_high = x_low + 1; }
        const float ly = y - y_low, lx = x - x_low, hy = 1.f - ly, hx = 1.f - lx;
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        const uint4 r1 = *reinterpret_cast<const uint4*>(base + ((size_t)y_low * W + x_low) * C);
        const uint4 r2 = *reinterpret_cast<const uint4*>(base + ((size_t)y_low * W + x_high) * C);
        const uint4 r3 = *reinterpret_cast<const uint4*>(base + ((size_t)y_high * W + x_low) * C);
        const uint4 r4 = *reinterpret_cast<const uint4*>(base + ((size_t)y_high * W + x_high) * C);
        const T* e1 = reinterpret_cast<const T*>(&r1);
        const T* e2 = reinterpret_cast<const T*>(&r2);
        const T* e3 = reinterpret_cast<const T*>(&r3);
        const T* e4 = reinterpret_cast<const T*>(&r4);
#pragma unroll
        for (int e = 0; e < VE; ++e)
          acc[e] += (w1 * Elem<T>::ld(e1 + e) + w2 * Elem<T>::ld(e2 + e) + w3 * Elem<T>::ld(e3 + e) +
                     w4 * Elem<T>::ld(e4 + e));
      }
    }
    if constexpr (PLANES) {
      static_assert(!PLANES || sizeof(T) == 4, "planes output: f32 features");
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[e] / count;
      const unsigned h0 = Half16<PT>::pack2(v[0], v[1]), h1 = Half16<PT>::pack2(v[2], v[3]);
      const unsigned l0 = Half16<PT>::pack2(v[0] - Half16<PT>::lo(h0), v[1] - Half16<PT>::hi(h0));
      const unsigned l1 = Half16<PT>::pack2(v[2] - Half16<PT>::lo(h1), v[3] - Half16<PT>::hi(h1));
      const size_t plane = (size_t)PH * PW * C;                       // elements per plane of one ROI row
      unsigned short* row = reinterpret_cast<unsigned short*>(out) + (size_t)k * 2 * plane +
                            (size_t)(bin - k * PH * PW) * C + (size_t)cv * 4;
      *reinterpret_cast<uint2*>(row) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(row + plane) = make_uint2(l0, l1);
    } else {
      uint4 o;
      T* oe = reinterpret_cast<T*>(&o);
#pragma unroll
      for (int e = 0; e < VE; ++e) Elem<T>::st(oe + e, acc[e] / count);
      *reinterpret_cast<uint4*>(out + (size_t)bin * C + (size_t)cv * VE) = o;
    }
  }
}


// bf16-mode variant with SEPARABLE weights, computed ONCE per ROI.  A bin's value is (1/count) sum over its
// grid_h x grid_w samples of the bilinear blend of 4 pixels; the blend weights factor into (hy | ly) x (hx | lx), so
//   bin = (1/count) sum_y sum_x Wy[y] Wx[x] f(y, x),   Wy[y] = sum of hy over samples whose lower row is y + ly over
// those whose upper row is y (same for x).  With the adaptive grid (sampling_ratio 0: grid = ceil(roi / bins)) adjacent
// samples are at most one pixel apart, so the touched pixels form a dense (<= grid_h + 1) x (<= grid_w + 1) patch:
// 16 loads instead of 36 for a 3 x 3 grid.
// A block owns ONE ROI (all pooled_h x pooled_w bins) and one slice of the channels: 140 threads build the pooled_h row
// tables and the pooled_w column tables once (the column tables are shared by all rows of bins), then every thread runs
// (load, 8 fma) per patch pixel over ~6 (bin, channel vector) items with broadcast LDS reads of the weights.  Round 2's
// form had one block per ROW of bins: the per-block latency chain (ROI read -> tables -> barrier) was paid 7 times per
// ROI and dominated (630 us for 3750 ROIs).  The patch loops are branch-free: rows / columns run to the BLOCK's maximum
// count at clamped (valid) addresses -- unused slots carry weight 0 in the tables -- so a row's loads issue back to back
// instead of under one exec-mask branch and wait per column.
// The sum is re-associated, hence bf16 mode only: the exact-f32 path keeps the reference's term order
// (roi_align_nhwc_vec_kernel).  A ROI whose patch does not fit the tables (grid > 9: a box larger than the map) takes
// the sample-by-sample loop in the same kernel; sampling_ratio > 0 (sparse samples) never comes here.
constexpr int RS_MAXP = 10;     // patch rows / columns per bin (grid <= 9)
constexpr int RS_MAXPW = 8;     // pooled_h, pooled_w <= 8

template <typename T, int NC, int RU>
__device__ __forceinline__ void roi_patch_accumulate(const T* __restrict__ base, int W, int H, int C, int y0, int x0,
                                                     int NR, const float* __restrict__ wy, const float* __restrict__ wx,
                                                     float (&acc)[Elem<T>::VE]) {
  constexpr int VE = Elem<T>::VE;
  int xo[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) xo[c] = min(x0 + c, W - 1) * C;
  for (int r = 0; r < NR; r += RU) {
    uint4 rv[RU][NC];                       // RU patch rows are requested before the first FMA
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const T* rowp = base + (size_t)min(y0 + r + u, H - 1) * W * C;
#pragma unroll
      for (int c = 0; c < NC; ++c) rv[u][c] = *reinterpret_cast<const uint4*>(rowp + xo[c]);
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const float wyr = r + u < RS_MAXP ? wy[r + u] : 0.f;      // (RU = 2 may look one slot past an odd NR: weight 0)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const T* ev = reinterpret_cast<const T*>(&rv[u][c]);
        const float wgt = wyr * wx[c];
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[e] = fmaf(wgt, Elem<T>::ld(ev + e), acc[e]);
      }
    }
  }
}

// PLANES (T = float, round 6): the pooled values leave as split-precision planes [hi | lo] of PT (bf16 / f16), as in
// roi_align_nhwc_vec_kernel<float, ., true>: the split-precision mode's ROIAlign in the separable form (the exact-term-order
// kernel it replaces there was 3.5x the bf16 kernel's time; the separable sums differ from it at f32 round-off, ~1e-7 relative).
template <typename T, int NSLICE, bool PLANES = false, typename PT = bf16_t>
__global__ __launch_bounds__(256, 4) void roi_align_nhwc_sep_kernel(const T* __restrict__ feat,
                                                                 const float* __restrict__ rois, T* __restrict__ out,
                                                                 int K, int C, int H, int W, float spatial_scale,
                                                                 int PH, int PW) {
  static_assert(!PLANES || sizeof(T) == 4, "planes output: f32 features");
  constexpr int VE = Elem<T>::VE;
  // one pooled 16-byte vector -> out (plain rows, or the hi / lo planes of ROI k's row)
  auto store = [&](int k_, int bin, int cv, const float (&acc)[Elem<T>::VE], float inv_count) {
    if constexpr (PLANES) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[e] * inv_count;
      const unsigned h0 = Half16<PT>::pack2(v[0], v[1]), h1 = Half16<PT>::pack2(v[2], v[3]);
      const unsigned l0 = Half16<PT>::pack2(v[0] - Half16<PT>::lo(h0), v[1] - Half16<PT>::hi(h0));
      const unsigned l1 = Half16<PT>::pack2(v[2] - Half16<PT>::lo(h1), v[3] - Half16<PT>::hi(h1));
      const size_t plane = (size_t)PH * PW * C;
      unsigned short* row = reinterpret_cast<unsigned short*>(out) + (size_t)k_ * 2 * plane + (size_t)bin * C + (size_t)cv * 4;
      *reinterpret_cast<uint2*>(row) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(row + plane) = make_uint2(l0, l1);
    } else {
      uint4 o;
      T* oe = reinterpret_cast<T*>(&o);
#pragma unroll
      for (int e = 0; e < VE; ++e) Elem<T>::st(oe + e, acc[e] * inv_count);
      *reinterpret_cast<uint4*>(out + ((size_t)k_ * PH * PW + bin) * C + (size_t)cv * VE) = o;
    }
  };
  __shared__ float s_wy[RS_MAXPW][RS_MAXP + 2];   // (+2: the two-row unroll reads one slot past an odd row count)
  __shared__ float s_wx[RS_MAXPW][RS_MAXP];
  __shared__ int s_y[RS_MAXPW][2];          // per bin row: first patch row, number of rows
  __shared__ int s_x[RS_MAXPW][2];          // per bin column: first patch column, number of columns
  const int CV = C / VE;
  // NSLICE == 8: the block handles one eighth of the channels, slice = blockIdx.x & 7 = the XCD the block runs on
  // (consecutive blocks go to consecutive XCDs): each XCD's private L2 then only ever sees its own 1/8 of every
  // feature map (1.2 MB per frame at 2048 channels) instead of all of it.  Placement is a speed heuristic only.
  const int slice = NSLICE == 8 ? (int)(blockIdx.x & 7) : 0;
  const int k = NSLICE == 8 ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int cvs = CV / NSLICE;                 // channel vectors of this block
  const float* roi = rois + (size_t)k * 5;
  const int b = (int)roi[0];
  const float roi_start_w = roi[1] * spatial_scale;
  const float roi_start_h = roi[2] * spatial_scale;
  const float roi_end_w = roi[3] * spatial_scale;
  const float roi_end_h = roi[4] * spatial_scale;
  const float roi_width = fmaxf(roi_end_w - roi_start_w, 1.f);
  const float roi_height = fmaxf(roi_end_h - roi_start_h, 1.f);
  const float bin_size_h = roi_height / (float)PH;
  const float bin_size_w = roi_width / (float)PW;
  const int grid_h = (int)ceilf(roi_height / (float)PH);
  const int grid_w = (int)ceilf(roi_width / (float)PW);
  const float count = (float)(grid_h * grid_w);
  // one axis sample -> (low index, high index, weight of low, weight of high); weights 0 when the sample is skipped
  auto axis = [](float p, int n, int& lo, int& hi, float& wl, float& wh) {
    if (p < -1.0f || p > (float)n) { lo = hi = 0; wl = wh = 0.f; return; }
    if (p <= 0.f) p = 0.f;
    lo = (int)p;
    if (lo >= n - 1) { hi = lo = n - 1; p = (float)lo; } else { hi = lo + 1; }
    wh = p - (float)lo;
    wl = 1.f - wh;
  };
  // weight of index `idx` along one axis of one bin, and the touched index range
  auto table = [&](float start, float bin_size, int bin_index, int grid, int n, int slot, int& first, int& cnt) {
    int f = n, l = -1;
    for (int i = 0; i < grid; ++i) {
      int lo, hi; float wl, wh;
      axis(start + bin_index * bin_size + (i + .5f) * bin_size / (float)grid, n, lo, hi, wl, wh);
      if (wl + wh > 0.f) { f = min(f, lo); l = max(l, hi); }
    }
    float w = 0.f;
    for (int i = 0; i < grid; ++i) {
      int lo, hi; float wl, wh;
      axis(start + bin_index * bin_size + (i + .5f) * bin_size / (float)grid, n, lo, hi, wl, wh);
      if (wl + wh > 0.f) {
        if (lo - f == slot) w += wl;
        if (hi - f == slot) w += wh;
      }
    }
    if (l < f) { f = 0; l = -1; }           // every sample of the bin lies outside the map: an empty patch
    first = f;
    cnt = l - f + 1;
    return w;
  };
  const int tid = threadIdx.x;
  if (tid < PH * (RS_MAXP + 2)) {
    const int ph = tid / (RS_MAXP + 2), slot = tid - ph * (RS_MAXP + 2);
    int first, cnt;
    const float w = table(roi_start_h, bin_size_h, ph, grid_h, H, slot, first, cnt);
    s_wy[ph][slot] = slot < RS_MAXP ? w : 0.f;
    if (slot == 0) { s_y[ph][0] = first; s_y[ph][1] = cnt; }
  } else if (tid >= 128 && tid < 128 + PW * RS_MAXP) {
    const int pw = (tid - 128) / RS_MAXP, slot = (tid - 128) - pw * RS_MAXP;
    int first, cnt;
    s_wx[pw][slot] = table(roi_start_w, bin_size_w, pw, grid_w, W, slot, first, cnt);
    if (slot == 0) { s_x[pw][0] = first; s_x[pw][1] = cnt; }
  }
  __syncthreads();
  int NR = 0, NCm = 0;                       // block-uniform patch extents
  for (int i = 0; i < PH; ++i) NR = max(NR, s_y[i][1]);
  for (int i = 0; i < PW; ++i) NCm = max(NCm, s_x[i][1]);
  const float inv_count = 1.f / count;
  const T* fb = feat + (size_t)b * H * W * C;
  const int nitem = PH * PW * cvs;
  if (NR > RS_MAXP || NCm > RS_MAXP) {
    // a patch wider than the tables (a ROI larger than the map): sample by sample, any grid
    for (int item = tid; item < nitem; item += 256) {
      const int bin = item / cvs, ph = bin / PW, pw = bin - ph * PW;
      const int cv = slice * cvs + (item - bin * cvs);
      float acc[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) acc[e] = 0.f;
      for (int iy = 0; iy < grid_h; ++iy) {
        int ylo, yhi; float wyl, wyh;
        axis(roi_start_h + ph * bin_size_h + (iy + .5f) * bin_size_h / (float)grid_h, H, ylo, yhi, wyl, wyh);
        for (int ix = 0; ix < grid_w; ++ix) {
          int xlo, xhi; float wxl, wxh;
          axis(roi_start_w + pw * bin_size_w + (ix + .5f) * bin_size_w / (float)grid_w, W, xlo, xhi, wxl, wxh);
          if (wyl + wyh == 0.f || wxl + wxh == 0.f) continue;
          const int ys[2] = {ylo, yhi}, xs[2] = {xlo, xhi};
          const float wys[2] = {wyl, wyh}, wxs[2] = {wxl, wxh};
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const uint4 v = *reinterpret_cast<const uint4*>(fb + ((size_t)ys[a] * W + xs[c]) * C + (size_t)cv * VE);
              const T* ev = reinterpret_cast<const T*>(&v);
              const float wgt = wys[a] * wxs[c];
#pragma unroll
              for (int e = 0; e < VE; ++e) acc[e] = fmaf(wgt, Elem<T>::ld(ev + e), acc[e]);
            }
        }
      }
      store(k, bin, cv, acc, inv_count);
    }
    return;
  }
  for (int item = tid; item < nitem; item += 256) {          // (bin, channel vector of the slice)
    const int bin = item / cvs, ph = bin / PW, pw = bin - ph * PW;
    const int cv = slice * cvs + (item - bin * cvs);
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
    const T* base = fb + (size_t)cv * VE;
    const int y0 = s_y[ph][0], x0 = s_x[pw][0];
    if (NCm <= 3) roi_patch_accumulate<T, 3, 2>(base, W, H, C, y0, x0, NR, s_wy[ph], s_wx[pw], acc);
    else if (NCm <= 4) roi_patch_accumulate<T, 4, 2>(base, W, H, C, y0, x0, NR, s_wy[ph], s_wx[pw], acc);
    else if (NCm <= 6) roi_patch_accumulate<T, 6, 2>(base, W, H, C, y0, x0, NR, s_wy[ph], s_wx[pw], acc);
    else roi_patch_accumulate<T, RS_MAXP, 1>(base, W, H, C, y0, x0, NR, s_wy[ph], s_wx[pw], acc);
    store(k, bin, cv, acc, inv_count);
  }
}

// ------------------------------------------------------------------------------------ stem conv on the matrix cores
// bf16 path: the same 7x7/2 conv as a GEMM  out[pixel][n] = sum_k A[pixel][k] * W[n][k].  K is laid out as 21 (c, r)
// groups of 8 = the 7 taps of one kernel row + one zero-weight tap: k = (c*7 + r)*8 + s, 168 padded to 176 = 11 MFMA
// steps of 16.  A block owns an 8 x 32 tile of output pixels: its 3 x 21 x 69 input patch sits in LDS as bf16, and a
// lane's A fragment (one pixel, 8 consecutive k = one kernel row) is 16 contiguous, 4-byte-aligned bytes of the patch:
// four ds_read_b32 whose addresses differ between lanes only by the pixel's column (stride 4 B: conflict-free).
// W (bf16 [64][176], packed once on the host) is staged with a 368-B row stride (conflict-free ds_read_b128).
// 3 GF per frame: the kernel is bound by reading the f32 image and writing the 64-channel map, not by the MFMAs.
constexpr int SM_TY = 8, SM_TX = 32;
constexpr int SM_PH = 2 * SM_TY + 5, SM_PW = 2 * SM_TX + 5, SM_PS = 72;   // patch rows / cols / row stride (elements)
constexpr int SM_K = 176, SM_WS = 184;                                    // padded K, LDS weight row stride (elements)

__host__ __device__ constexpr int stem_group_off(int g) {   // LDS element offset of (c, r) group g relative to the pixel
  const int gg = g < 21 ? g : 20;                           // group 21 meets zero weights: any valid address will do
  return (gg / 7) * (SM_PH * SM_PS) + (gg % 7) * SM_PS;
}

// U8: the input is the uint8 HWC RGB frame batch itself and the preprocessing (ToTensor, to_bgr255, Normalize:
// mega_core/data/transforms/transforms.py:83-129 = frames.hip's preprocess_kernel) happens on the patch load:
// value = (float)byte - mean[c], the same f32 subtraction, then the same f32 -> bf16 conversion -- identical bits, a quarter
// of the input bytes, no f32 image in HBM and no preprocess launch.
template <bool U8>
__global__ __launch_bounds__(256, 4) void stem_mfma_kernel(const void* __restrict__ in_, const bf16_t* __restrict__ w,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                        int N, int H, int W, int Ho, int Wo, float m0, float m1, float m2,
                                                        int to_bgr) {
  const float* __restrict__ in = reinterpret_cast<const float*>(in_);
  const unsigned char* __restrict__ in8 = reinterpret_cast<const unsigned char*>(in_);
  // one buffer: [patch | weights] while the MFMAs run, then the block's 8 x 32 x 64 output tile (144-B pixel stride)
  constexpr int SM_PATCH_B = (3 * SM_PH * SM_PS * 2 + 15) / 16 * 16, SM_OUT_PS = 144;
  constexpr int SM_LDS_B = SM_TY * SM_TX * SM_OUT_PS > SM_PATCH_B + 64 * SM_WS * 2 ? SM_TY * SM_TX * SM_OUT_PS
                                                                                   : SM_PATCH_B + 64 * SM_WS * 2;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SM_LDS_B];
  unsigned short* patch = reinterpret_cast<unsigned short*>(smem);
  unsigned short* wl = reinterpret_cast<unsigned short*>(smem + SM_PATCH_B);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.z;
  const int oy0 = blockIdx.y * SM_TY, ox0 = blockIdx.x * SM_TX;
  const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
  // all of a thread's patch elements (17) and weight vectors (5) are REQUESTED before the first one is used: written
  // as load-then-store per element the loop serialised 17 global-memory round trips per block (0.49 ms per 20 frames,
  // 5 % of it MFMA time)
  // (the patch is filled over its whole 72-element row stride: the zero-weight 8th tap of a kernel row reads one
  //  column past the 69 real ones, and 0 x whatever-LDS-held is NaN when that happens to be a NaN pattern)
  constexpr int NPL = (3 * SM_PH * SM_PS + 255) / 256, NWL = (64 * (SM_K / 8) + 255) / 256;
  float pv[NPL];
  u32x4_t wv[NWL];      // (ext vector, not HIP's uint4 class: an array of those is placed in scratch memory)
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int e = tid + 256 * i;
    const int c = e / (SM_PH * SM_PS);
    const int rem = e - c * (SM_PH * SM_PS);
    const int y = rem / SM_PS, x = rem - y * SM_PS;
    const int iy = iy0 + y, ix = ix0 + x;
    pv[i] = 0.f;
    if (e < 3 * SM_PH * SM_PS && x < SM_PW && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
      if (U8) {
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
        pv[i] = (float)in8[(((size_t)n * H + iy) * W + ix) * 3 + (to_bgr ? 2 - c : c)] - mean;
      } else {
        pv[i] = in[((size_t)(n * 3 + c) * H + iy) * W + ix];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NWL; ++i) {
    const int e = tid + 256 * i;
    const int row = e / (SM_K / 8), v = e - row * (SM_K / 8);
    if (e < 64 * (SM_K / 8)) wv[i] = *reinterpret_cast<const u32x4_t*>(w + row * SM_K + v * 8);
  }
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int e = tid + 256 * i;
    if (e < 3 * SM_PH * SM_PS) patch[e] = f32_to_bf16(pv[i]);
  }
#pragma unroll
  for (int i = 0; i < NWL; ++i) {
    const int e = tid + 256 * i;
    const int row = e / (SM_K / 8), v = e - row * (SM_K / 8);
    if (e < 64 * (SM_K / 8)) *reinterpret_cast<u32x4_t*>(&wl[row * SM_WS + v * 8]) = wv[i];
  }
  __syncthreads();

  const int p = lane & 31, half = lane >> 5;
  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < SM_K / 16; ++ks) {
    uint4 a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned* base = reinterpret_cast<const unsigned*>(
          patch + (2 * (2 * wave + i)) * SM_PS + 2 * p + (half ? stem_group_off(2 * ks + 1) : stem_group_off(2 * ks)));
      a[i] = make_uint4(base[0], base[1], base[2], base[3]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
      b[j] = *reinterpret_cast<const uint4*>(&wl[(j * 32 + p) * SM_WS + ks * 16 + half * 8]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[i]),
                                                            __builtin_bit_cast(bf16x8_t, b[j]), acc[i][j], 0, 0, 0);
  }
  // lane owns channel (lane & 31) + 32 j and the pixels (r&3) + 8 (r>>2) + 4 half of output row 2*wave + i: written
  // straight to memory that is 2 bytes per lane, 64 contiguous bytes per half-wave (PMC: 2x write amplification).
  // The tile goes through LDS instead and leaves as whole 128-byte pixels, 4 KB contiguous per tile row.
  __syncthreads();                       // every wave is done with the patch / weights
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ch = j * 32 + p;
    const float sc = scale[ch], bi = bias[ch];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int px = (r & 3) + 8 * (r >> 2) + 4 * half;
        *reinterpret_cast<unsigned short*>(smem + ((2 * wave + i) * SM_TX + px) * SM_OUT_PS + ch * 2) =
            (bf16_t)(f32_to_bf16(fmaxf(acc[i][j][r] * sc + bi, 0.f)) & 0x7fffu);   // (& 0x7fff: fmaxf(-0, 0) may be -0; every ReLU of the path gives +0)
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < SM_TY * SM_TX * 8 / 256; ++i) {
    const int v = tid + 256 * i;
    const int pl = v >> 3, cv = v & 7;
    const int oy = oy0 + pl / SM_TX, ox = ox0 + pl % SM_TX;
    if (oy < Ho && ox < Wo)
      *reinterpret_cast<uint4*>(out + (((size_t)n * Ho + oy) * Wo + ox) * 64 + cv * 8) =
          *reinterpret_cast<const uint4*>(smem + pl * SM_OUT_PS + cv * 16);
  }
}

// ------------------------------------------------------------------------------- stem + 3x3/2 max-pool in one kernel (bf16)
// The stem's output (300 x 500 x 64 per frame, 19 MB) was written by stem_mfma_kernel only to be read back by the max-pool
// (resnet.py:361-366: F.max_pool2d(x, 3, 2, 1)).  Here a block owns a 4 x 16 tile of POOLED pixels: it computes the 9 x 33
// stem pixels their windows cover (297 GEMM rows in ten 32-row MFMA blocks, pixel p = row * 33 + col), stages them in LDS as
// bf16 with everything outside the stem map forced to 0 (every window holds at least one real pixel and all values are >= 0
// after the ReLU, so 0 is the identity of this max), and writes max over the 3 x 3 windows: the 19 MB never reach HBM.
// Same patch / weight layout, same MFMA and K order as stem_mfma_kernel; max of bf16 values is exact: bit-identical to
// stem + max-pool (tests/test_kernels_gpu.py::test_stem_pool_bit_equal).
constexpr int SP_PY = 4, SP_PX = 16;                       // pooled tile
constexpr int SP_TY = 2 * SP_PY + 1, SP_TX = 2 * SP_PX + 1; // stem pixels it needs: 9 x 33
constexpr int SP_NPIX = SP_TY * SP_TX;                     // 297
constexpr int SP_NBLK = (SP_NPIX + 31) / 32;               // 10 MFMA row blocks
constexpr int SP_PH = 2 * SP_TY + 5, SP_PW = 2 * SP_TX + 5, SP_PS = 72;   // input patch 23 x 71, row stride 72
static_assert(SP_PW + 1 <= SP_PS, "the zero-weight 8th tap reads one column past the patch");

__host__ __device__ constexpr int sp_group_off(int g) {
  const int gg = g < 21 ? g : 20;
  return (gg / 7) * (SP_PH * SP_PS) + (gg % 7) * SP_PS;
}

// HT: bf16_t or f16_t (IEEE half, round 6) = the type of the patch / weights in LDS and of the pooled output
template <bool U8, typename HT = bf16_t>
__global__ __launch_bounds__(256, 3) void stem_pool_kernel(const void* __restrict__ in_, const bf16_t* __restrict__ w,
                                                        const float* __restrict__ scale, const float* __restrict__ bias,
                                                        bf16_t* __restrict__ out, int N, int H, int W, int Ho, int Wo, int Hp,
                                                        int Wp, float m0, float m1, float m2, int to_bgr) {
  const float* __restrict__ in = reinterpret_cast<const float*>(in_);
  const unsigned char* __restrict__ in8 = reinterpret_cast<const unsigned char*>(in_);
  constexpr int PATCH_B = (3 * SP_PH * SP_PS * 2 + 15) / 16 * 16, OUT_PS = 144;
  constexpr int LDS_B = SP_NBLK * 32 * OUT_PS > PATCH_B + 64 * SM_WS * 2 ? SP_NBLK * 32 * OUT_PS : PATCH_B + 64 * SM_WS * 2;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_B];
  unsigned short* patch = reinterpret_cast<unsigned short*>(smem);
  unsigned short* wl = reinterpret_cast<unsigned short*>(smem + PATCH_B);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.z;
  const int py0 = blockIdx.y * SP_PY, px0 = blockIdx.x * SP_PX;
  const int oy0 = 2 * py0 - 1, ox0 = 2 * px0 - 1;          // first stem pixel of the tile (may be -1: outside)
  const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
  constexpr int NPL = (3 * SP_PH * SP_PS + 255) / 256, NWL = (64 * (SM_K / 8) + 255) / 256;
  float pv[NPL];
  u32x4_t wv[NWL];      // (ext vector, not HIP's uint4 class: an array of those is placed in scratch memory)
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int e = tid + 256 * i;
    const int c = e / (SP_PH * SP_PS);
    const int rem = e - c * (SP_PH * SP_PS);
    const int y = rem / SP_PS, x = rem - y * SP_PS;
    const int iy = iy0 + y, ix = ix0 + x;
    pv[i] = 0.f;
    if (e < 3 * SP_PH * SP_PS && x < SP_PW && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
      if (U8) {
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
        pv[i] = (float)in8[(((size_t)n * H + iy) * W + ix) * 3 + (to_bgr ? 2 - c : c)] - mean;
      } else {
        pv[i] = in[((size_t)(n * 3 + c) * H + iy) * W + ix];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NWL; ++i) {
    const int e = tid + 256 * i;
    const int row = e / (SM_K / 8), v = e - row * (SM_K / 8);
    if (e < 64 * (SM_K / 8)) wv[i] = *reinterpret_cast<const u32x4_t*>(w + row * SM_K + v * 8);
  }
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int e = tid + 256 * i;
    if (e < 3 * SP_PH * SP_PS) patch[e] = Half16<HT>::cvt(pv[i]);
  }
#pragma unroll
  for (int i = 0; i < NWL; ++i) {
    const int e = tid + 256 * i;
    const int row = e / (SM_K / 8), v = e - row * (SM_K / 8);
    if (e < 64 * (SM_K / 8)) *reinterpret_cast<u32x4_t*>(&wl[row * SM_WS + v * 8]) = wv[i];
  }
  __syncthreads();

  const int p = lane & 31, half = lane >> 5;
  // wave w owns MFMA row blocks w, w + 4, w + 8 (waves 0 and 1: three blocks, 2 and 3: two)
  constexpr int NB = 3;
  f32x16_t acc[NB][2];
  int pbase[NB];                       // patch element offset of this lane's pixel (row * 33 + col -> stem (r, c) -> patch (2 r, 2 c))
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int pix = min(32 * (wave + 4 * b) + p, SP_NPIX - 1);
    const int r = pix / SP_TX, c = pix - r * SP_TX;
    pbase[b] = 2 * r * SP_PS + 2 * c;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[b][j][q] = 0.f;
  }
#pragma unroll
  for (int ks = 0; ks < SM_K / 16; ++ks) {
    u32x4_t bfr[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bfr[j] = *reinterpret_cast<const u32x4_t*>(&wl[(j * 32 + p) * SM_WS + ks * 16 + half * 8]);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (wave + 4 * b < SP_NBLK) {    // (wave-uniform)
        const unsigned* base = reinterpret_cast<const unsigned*>(patch + pbase[b] + (half ? sp_group_off(2 * ks + 1) : sp_group_off(2 * ks)));
        const u32x4_t a = {base[0], base[1], base[2], base[3]};
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[b][j] = Half16<HT>::mfma32(a, bfr[j], acc[b][j]);
      }
    }
  }
  __syncthreads();                     // every wave is done with the patch / weights
  // ---- stem pixels -> LDS as bf16 [pixel][64 ch] (144-B stride); 0 outside the stem map
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ch = j * 32 + p;
    const float sc = scale[ch], bi = bias[ch];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (wave + 4 * b < SP_NBLK) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int pix = 32 * (wave + 4 * b) + (q & 3) + 8 * (q >> 2) + 4 * half;
          const int r = pix / SP_TX, c = pix - r * SP_TX;
          const int oy = oy0 + r, ox = ox0 + c;
          const bool ok = pix < SP_NPIX && (unsigned)oy < (unsigned)Ho && (unsigned)ox < (unsigned)Wo;
          const unsigned short hv = (unsigned short)(Half16<HT>::cvt(fmaxf(acc[b][j][q] * sc + bi, 0.f)) & 0x7fffu);   // +0, never -0: the pool below orders bit patterns
          *reinterpret_cast<unsigned short*>(smem + pix * OUT_PS + ch * 2) = ok ? hv : (unsigned short)0;
        }
      }
    }
  }
  __syncthreads();
  // ---- 3 x 3 / 2 max over the staged tile: pooled (y, x) of the tile = stem rows 2 y .. 2 y + 2, cols 2 x .. 2 x + 2 of the tile
#pragma unroll
  for (int i = 0; i < SP_PY * SP_PX * 8 / 256; ++i) {
    const int v = tid + 256 * i;
    const int pl = v >> 3, cv = v & 7;
    const int ty = pl / SP_PX, tx = pl - ty * SP_PX;
    const int oy = py0 + ty, ox = px0 + tx;
    typedef unsigned short u16x8_t __attribute__((ext_vector_type(8)));
    u16x8_t m = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const u16x8_t q = *reinterpret_cast<const u16x8_t*>(smem + ((2 * ty + dy) * SP_TX + 2 * tx + dx) * OUT_PS + cv * 16);
        m = __builtin_elementwise_max(m, q);      // (non-negative bf16 / f16 values order like their bit patterns)
      }
    if (oy < Hp && ox < Wp) *reinterpret_cast<u16x8_t*>(out + (((size_t)n * Hp + oy) * Wp + ox) * 64 + cv * 8) = m;
  }
}

}  // namespace

extern "C" int mega_stem_conv_bn_relu_bf16(const float* in, const void* w_n176_bf16, const float* scale,
                                           const float* bias, void* out, int N, int H, int W, void* stream) {
  mega_clear_error();
  if (!in || !w_n176_bf16 || !scale || !bias || !out || N <= 0 || H <= 0 || W <= 0) return MEGA_ERR_ARG;
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  dim3 grid(cdiv(Wo, SM_TX), cdiv(Ho, SM_TY), N);
  hipLaunchKernelGGL(stem_mfma_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const void*)in,
                     (const bf16_t*)w_n176_bf16, scale, bias, (bf16_t*)out, N, H, W, Ho, Wo, 0.f, 0.f, 0.f, 0);
  return mega_check_launch();
}

// The same layer fed by the uint8 frames [N][H][W][3] (RGB): preprocessing fused into the patch load (see the kernel).
// mean[c] is subtracted from OUTPUT channel c of the preprocessed image (to_bgr: channel 0 = B).  Same bits as
// mega_preprocess_frames + mega_stem_conv_bn_relu_bf16.
extern "C" int mega_stem_conv_bn_relu_bf16_u8(const void* frames_u8, const void* w_n176_bf16, const float* scale,
                                              const float* bias, void* out, int N, int H, int W, float mean0, float mean1,
                                              float mean2, int to_bgr, void* stream) {
  mega_clear_error();
  if (!frames_u8 || !w_n176_bf16 || !scale || !bias || !out || N <= 0 || H <= 0 || W <= 0) return MEGA_ERR_ARG;
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  dim3 grid(cdiv(Wo, SM_TX), cdiv(Ho, SM_TY), N);
  hipLaunchKernelGGL(stem_mfma_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, frames_u8, (const bf16_t*)w_n176_bf16,
                     scale, bias, (bf16_t*)out, N, H, W, Ho, Wo, mean0, mean1, mean2, to_bgr);
  return mega_check_launch();
}

extern "C" int mega_stem_conv_bn_relu(const float* in, const float* w_tap64, const float* scale, const float* bias,
                                      void* out, int N, int H, int W, int out_dtype, void* stream) {
  mega_clear_error();
  if (!in || !w_tap64 || !scale || !bias || !out || N <= 0 || H <= 0 || W <= 0) return MEGA_ERR_ARG;
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  dim3 grid(cdiv(Wo, ST_T), cdiv(Ho, ST_T), N);
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == MEGA_BF16)
    hipLaunchKernelGGL((stem_conv_kernel<bf16_t>), grid, dim3(256), 0, st, in, w_tap64, scale, bias, (bf16_t*)out, N, H,
                       W, Ho, Wo);
  else if (out_dtype == MEGA_F32)
    hipLaunchKernelGGL((stem_conv_kernel<float>), grid, dim3(256), 0, st, in, w_tap64, scale, bias, (float*)out, N, H, W,
                       Ho, Wo);
  else
    return MEGA_ERR_ARG;
  return mega_check_launch();
}

extern "C" int mega_maxpool3x3s2_nhwc(const void* in, void* out, int N, int H, int W, int C, int dtype, void* stream) {
  mega_clear_error();
  if (!in || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0) return MEGA_ERR_ARG;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MEGA_BF16) {
    if (C % 8) return MEGA_ERR_ARG;
    const size_t total = (size_t)N * Ho * Wo * (C / 8);
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL((maxpool_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)out, N, H,
                       W, C, Ho, Wo);
  } else if (dtype == MEGA_F16) {
    if (C % 8) return MEGA_ERR_ARG;
    const size_t total = (size_t)N * Ho * Wo * (C / 8);
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL((maxpool_kernel<f16_t>), dim3(blocks), dim3(256), 0, st, (const f16_t*)in, (f16_t*)out, N, H,
                       W, C, Ho, Wo);
  } else if (dtype == MEGA_F32) {
    if (C % 4) return MEGA_ERR_ARG;
    const size_t total = (size_t)N * Ho * Wo * (C / 4);
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL((maxpool_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)in, (float*)out, N, H, W,
                       C, Ho, Wo);
  } else {
    return MEGA_ERR_ARG;
  }
  return mega_check_launch();
}

extern "C" int mega_roi_align_fwd(const void* feat, const float* rois, void* out, int K, int C, int H, int W,
                                  float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio, int in_nhwc,
                                  int out_nhwc, int dtype, int out_dtype, void* stream) {
  mega_clear_error();
  if (K == 0) return MEGA_OK;
  if (!feat || !rois || !out || K < 0 || C <= 0 || H <= 0 || W <= 0 || pooled_h <= 0 || pooled_w <= 0)
    return MEGA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (in_nhwc && out_nhwc && dtype == out_dtype && C % (dtype != MEGA_F32 ? 8 : 4) == 0) {
    const int CV = C / (dtype != MEGA_F32 ? 8 : 4);
    const long long total = (long long)K * pooled_h * pooled_w * CV;
    static const bool no_slice = getenv("MEGA_ROI_NO_XCD_SLICE") != nullptr;       // A/B switch (experiments)
    const bool sliced = !no_slice && CV % 8 == 0 && CV / 8 >= 16 && total / 8 >= 256 * 64;
    long long nb = ((sliced ? total / 8 : total) + 255) / 256;
    if (nb > 131072) nb = 131072;
    dim3 vgrid((unsigned)(sliced ? nb * 8 : nb));
    static const bool no_sep = getenv("MEGA_ROI_NO_SEPARABLE") != nullptr;         // A/B switch (experiments)
    // the separable per-ROI form: adaptive grid only (sampling_ratio > 0 spreads a bin's samples over a sparse patch);
    // ROIs whose patch exceeds its tables fall back inside the kernel, so no bound on the boxes is assumed here
    if (dtype == MEGA_F16 && !no_sep && sampling_ratio <= 0 && pooled_w <= RS_MAXPW && pooled_h <= RS_MAXPW) {
      if (!no_slice && CV % 8 == 0 && CV / 8 >= 16)
        hipLaunchKernelGGL((roi_align_nhwc_sep_kernel<f16_t, 8>), dim3((unsigned)(K * 8)), dim3(256), 0, st,
                           (const f16_t*)feat, rois, (f16_t*)out, K, C, H, W, spatial_scale, pooled_h, pooled_w);
      else
        hipLaunchKernelGGL((roi_align_nhwc_sep_kernel<f16_t, 1>), dim3((unsigned)K), dim3(256), 0, st,
                           (const f16_t*)feat, rois, (f16_t*)out, K, C, H, W, spatial_scale, pooled_h, pooled_w);
      return mega_check_launch();
    }
    if (dtype == MEGA_BF16 && !no_sep && sampling_ratio <= 0 && pooled_w <= RS_MAXPW && pooled_h <= RS_MAXPW) {
      if (!no_slice && CV % 8 == 0 && CV / 8 >= 16)
        hipLaunchKernelGGL((roi_align_nhwc_sep_kernel<bf16_t, 8>), dim3((unsigned)(K * 8)), dim3(256), 0, st,
                           (const bf16_t*)feat, rois, (bf16_t*)out, K, C, H, W, spatial_scale, pooled_h, pooled_w);
      else
        hipLaunchKernelGGL((roi_align_nhwc_sep_kernel<bf16_t, 1>), dim3((unsigned)K), dim3(256), 0, st,
                           (const bf16_t*)feat, rois, (bf16_t*)out, K, C, H, W, spatial_scale, pooled_h, pooled_w);
      return mega_check_launch();
    }
    if (dtype == MEGA_BF16 && sliced)
      hipLaunchKernelGGL((roi_align_nhwc_vec_kernel<bf16_t, true>), vgrid, dim3(256), 0, st, (const bf16_t*)feat, rois,
                         (bf16_t*)out, K, C, H, W, spatial_scale, pooled_h, pooled_w, sampling_ratio);
    else if (dtype == MEGA_BF16)
      hipLaunchKernelGGL((roi_align_nhwc_vec_kernel<bf16_t, false>), vgrid, dim3(256), 0, st, (const bf16_t*)feat, rois,
                         (bf16_t*)out, K, C, H, W, spatial_scale, pooled_h, pooled_w, sampling_ratio);
    else if (dtype == MEGA_F16 && sliced)
      hipLaunchKernelGGL((roi_align_nhwc_vec_kernel<f16_t, true>), vgrid, dim3(256), 0, st, (const f16_t*)feat, rois,
                         (f16_t*)out, K, C, H, W, spatial_scale, pooled_h, pooled_w, sampling_ratio);
    else if (dtype == MEGA_F16)
      hipLaunchKernelGGL((roi_align_nhwc_vec_kernel<f16_t, false>), vgrid, dim3(256), 0, st, (const f16_t*)feat, rois,
                         (f16_t*)out, K, C, H, W, spatial_scale, pooled_h, pooled_w, sampling_ratio);
    else if (dtype == MEGA_F32 && sliced)
      hipLaunchKernelGGL((roi_align_nhwc_vec_kernel<float, true>), vgrid, dim3(256), 0, st, (const float*)feat, rois,
                         (float*)out, K, C, H, W, spatial_scale, pooled_h, pooled_w, sampling_ratio);
    else if (dtype == MEGA_F32)
      hipLaunchKernelGGL((roi_align_nhwc_vec_kernel<float, false>), vgrid, dim3(256), 0, st, (const float*)feat, rois,
                         (float*)out, K, C, H, W, spatial_scale, pooled_h, pooled_w, sampling_ratio);
    else
      return MEGA_ERR_ARG;
    return mega_check_launch();
  }
  dim3 grid((unsigned)(K * pooled_h * pooled_w));
  const int threads = C >= 256 ? 256 : (C >= 128 ? 128 : 64);
  if (dtype == MEGA_BF16 && out_dtype == MEGA_BF16)
    hipLaunchKernelGGL((roi_align_kernel<bf16_t, bf16_t>), grid, dim3(threads), 0, st, (const bf16_t*)feat, rois,
                       (bf16_t*)out, K, C, H, W, spatial_scale, pooled_h, pooled_w, sampling_ratio, in_nhwc, out_nhwc);
  else if (dtype == MEGA_F16 && out_dtype == MEGA_F16)
    hipLaunchKernelGGL((roi_align_kernel<f16_t, f16_t>), grid, dim3(threads), 0, st, (const f16_t*)feat, rois,
                       (f16_t*)out, K, C, H, W, spatial_scale, pooled_h, pooled_w, sampling_ratio, in_nhwc, out_nhwc);
  else if (dtype == MEGA_F32 && out_dtype == MEGA_F32)
    hipLaunchKernelGGL((roi_align_kernel<float, float>), grid, dim3(threads), 0, st, (const float*)feat, rois,
                       (float*)out, K, C, H, W, spatial_scale, pooled_h, pooled_w, sampling_ratio, in_nhwc, out_nhwc);
  else
    return MEGA_ERR_ARG;
  return mega_check_launch();
}

extern "C" int mega_avgpool2x2_ceil_nhwc(const void* in, void* out, int N, int H, int W, int C, int dtype,
                                         void* stream) {
  mega_clear_error();
  if (!in || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0) return MEGA_ERR_ARG;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  hipStream_t st = (hipStream_t)stream;
  const int ve = dtype == MEGA_BF16 ? 8 : 4;
  if (C % ve) return MEGA_ERR_ARG;
  const size_t total = (size_t)N * Ho * Wo * (C / ve);
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  if (dtype == MEGA_BF16)
    hipLaunchKernelGGL((avgpool2_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)out, N, H,
                       W, C, Ho, Wo);
  else if (dtype == MEGA_F32)
    hipLaunchKernelGGL((avgpool2_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)in, (float*)out, N, H, W,
                       C, Ho, Wo);
  else
    return MEGA_ERR_ARG;
  return mega_check_launch();
}

// 7x7/2 stem conv + FrozenBN + ReLU + 3x3/2 max-pool (pad 1) in one kernel, bf16: in = preprocessed f32 NCHW [N][3][H][W]
// (u8 = 0) or the uint8 frames [N][H][W][3] RGB with the preprocessing on the patch load (u8 = 1, as
// mega_stem_conv_bn_relu_bf16_u8); out: NHWC bf16 [N][Hp][Wp][64], Hp = (Ho - 1) / 2 + 1 with Ho = (H - 1) / 2 + 1.
// Bit-identical to mega_stem_conv_bn_relu_bf16[_u8] followed by mega_maxpool3x3s2_nhwc.
// dtype: MEGA_BF16, or MEGA_F16 (weights and output IEEE half).
extern "C" int mega_stem_pool_dt(const void* in, int u8, const void* w_n176, const float* scale, const float* bias,
                                 void* out, int N, int H, int W, float mean0, float mean1, float mean2, int to_bgr,
                                 int dtype, void* stream) {
  mega_clear_error();
  if (!in || !w_n176 || !scale || !bias || !out || N <= 0 || H <= 0 || W <= 0) return MEGA_ERR_ARG;
  if (dtype != MEGA_BF16 && dtype != MEGA_F16) return MEGA_ERR_ARG;
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const int Hp = (Ho + 2 - 3) / 2 + 1, Wp = (Wo + 2 - 3) / 2 + 1;
  dim3 grid(cdiv(Wp, SP_PX), cdiv(Hp, SP_PY), N);
  hipStream_t st = (hipStream_t)stream;
  const bf16_t* w = (const bf16_t*)w_n176;
  bf16_t* o = (bf16_t*)out;
  if (!u8) { mean0 = mean1 = mean2 = 0.f; to_bgr = 0; }
  if (dtype == MEGA_F16) {
    if (u8) hipLaunchKernelGGL((stem_pool_kernel<true, f16_t>), grid, dim3(256), 0, st, in, w, scale, bias, o, N, H, W, Ho, Wo, Hp, Wp, mean0, mean1, mean2, to_bgr);
    else hipLaunchKernelGGL((stem_pool_kernel<false, f16_t>), grid, dim3(256), 0, st, in, w, scale, bias, o, N, H, W, Ho, Wo, Hp, Wp, mean0, mean1, mean2, to_bgr);
  } else {
    if (u8) hipLaunchKernelGGL((stem_pool_kernel<true, bf16_t>), grid, dim3(256), 0, st, in, w, scale, bias, o, N, H, W, Ho, Wo, Hp, Wp, mean0, mean1, mean2, to_bgr);
    else hipLaunchKernelGGL((stem_pool_kernel<false, bf16_t>), grid, dim3(256), 0, st, in, w, scale, bias, o, N, H, W, Ho, Wo, Hp, Wp, mean0, mean1, mean2, to_bgr);
  }
  return mega_check_launch();
}

extern "C" int mega_stem_pool_bf16(const void* in, int u8, const void* w_n176_bf16, const float* scale, const float* bias,
                                   void* out, int N, int H, int W, float mean0, float mean1, float mean2, int to_bgr,
                                   void* stream) {
  return mega_stem_pool_dt(in, u8, w_n176_bf16, scale, bias, out, N, H, W, mean0, mean1, mean2, to_bgr, MEGA_BF16, stream);
}

// mega_roi_align_fwd for f32 NHWC features with the result as split-precision planes: out bf16 [K][2 PH PW C] =
// [hi | lo] of the f32 pooled row [PH PW C] (the same arithmetic, term by term, as the f32 kernel -- ROIAlign_cuda.cu:64-122 --
// then hi = bf16(v), lo = bf16(v - hi)).  C % 4 == 0.  The A operand of the split-precision fc0 (mega_conv2d_nhwc_sp).
extern "C" int mega_roi_align_fwd_planes_dt(const float* feat, const float* rois, void* out, int K, int C, int H, int W,
                                            float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio, int dtype,
                                            void* stream) {
  mega_clear_error();
  if (K == 0) return MEGA_OK;
  if (dtype != MEGA_BF16 && dtype != MEGA_F16) return MEGA_ERR_ARG;
  if (!feat || !rois || !out || K < 0 || C <= 0 || C % 4 || H <= 0 || W <= 0 || pooled_h <= 0 || pooled_w <= 0) return MEGA_ERR_ARG;
  const int CV = C / 4;
  const long long total = (long long)K * pooled_h * pooled_w * CV;
  const bool sliced = CV % 8 == 0 && CV / 8 >= 16 && total / 8 >= 256 * 64;
  long long nb = ((sliced ? total / 8 : total) + 255) / 256;
  if (nb > 131072) nb = 131072;
  dim3 vgrid((unsigned)(sliced ? nb * 8 : nb));
  hipStream_t st = (hipStream_t)stream;
  // the separable per-ROI form (adaptive grid, one block per ROI x XCD channel slice), as the 16-bit kernels: the sums differ
  // from the exact-term-order kernel at f32 round-off (~1e-7 relative); MEGA_ROI_NO_SEPARABLE=1 keeps the latter
  static const bool no_sep = getenv("MEGA_ROI_NO_SEPARABLE") != nullptr;
  if (!no_sep && sliced && sampling_ratio <= 0 && pooled_w <= RS_MAXPW && pooled_h <= RS_MAXPW) {
    if (dtype == MEGA_F16)
      hipLaunchKernelGGL((roi_align_nhwc_sep_kernel<float, 8, true, f16_t>), dim3((unsigned)(K * 8)), dim3(256), 0, st, feat, rois,
                         (float*)out, K, C, H, W, spatial_scale, pooled_h, pooled_w);
    else
      hipLaunchKernelGGL((roi_align_nhwc_sep_kernel<float, 8, true, bf16_t>), dim3((unsigned)(K * 8)), dim3(256), 0, st, feat, rois,
                         (float*)out, K, C, H, W, spatial_scale, pooled_h, pooled_w);
    return mega_check_launch();
  }
  if (dtype == MEGA_F16) {
    if (sliced)
      hipLaunchKernelGGL((roi_align_nhwc_vec_kernel<float, true, true, f16_t>), vgrid, dim3(256), 0, st, feat, rois, (float*)out, K, C,
                         H, W, spatial_scale, pooled_h, pooled_w, sampling_ratio);
    else
      hipLaunchKernelGGL((roi_align_nhwc_vec_kernel<float, false, true, f16_t>), vgrid, dim3(256), 0, st, feat, rois, (float*)out, K,
                         C, H, W, spatial_scale, pooled_h, pooled_w, sampling_ratio);
  } else if (sliced)
    hipLaunchKernelGGL((roi_align_nhwc_vec_kernel<float, true, true>), vgrid, dim3(256), 0, st, feat, rois, (float*)out, K, C, H,
                       W, spatial_scale, pooled_h, pooled_w, sampling_ratio);
  else
    hipLaunchKernelGGL((roi_align_nhwc_vec_kernel<float, false, true>), vgrid, dim3(256), 0, st, feat, rois, (float*)out, K, C, H,
                       W, spatial_scale, pooled_h, pooled_w, sampling_ratio);
  return mega_check_launch();
}

extern "C" int mega_roi_align_fwd_planes(const float* feat, const float* rois, void* out, int K, int C, int H, int W,
                                         float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio, void* stream) {
  return mega_roi_align_fwd_planes_dt(feat, rois, out, K, C, H, W, spatial_scale, pooled_h, pooled_w, sampling_ratio, MEGA_BF16,
                                      stream);
}
