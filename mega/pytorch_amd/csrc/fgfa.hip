// FGFA flow-guided feature aggregation (BASELINE configs[4]): the HBM-bound part of
// GeneralizedRCNNFGFA._forward_test (mega_core/modeling/detector/generalized_rcnn_fgfa.py):
//   :45-62  get_grid / resample : F.grid_sample(all_features, ((flow + grid) / ((W-1)/2, (H-1)/2) - 1), bilinear,
//                                               padding_mode="border")   (align_corners = False, torch default)
//   :64-76  compute_norm / compute_weight : cosine similarity of the 2048-d warped embeddings with the key frame's
//   :203-211 softmax over the T frames, weighted sum of the 1024-d warped features
// fused into ONE kernel: the [T,3072,H,W] warped tensor (559 MB at T=19, f32) is never materialised.
//
// Layout: feats NHWC [T][H][W][Cf+Ce] (first Cf channels = backbone features, next Ce = embeddings),
// flow [T][2][H][W] f32 (x displacement, y displacement, in feature-map pixels), out [H][W][Cf].
// One workgroup per output pixel; a wave reads 64 consecutive 16-byte channel vectors of one neighbour pixel
// (1 KiB, fully coalesced); the T warped feature vectors wait in LDS (T*Cf f32 <= 128 KiB) for the softmax.
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

template <typename T>
__global__ __launch_bounds__(256) void fgfa_kernel(const T* __restrict__ feats, const float* __restrict__ flow,
                                                   T* __restrict__ out, float* __restrict__ weights_out, int NT,
                                                   int H, int W, int Cf, int Ce, int key) {
  constexpr int VE = Elem<T>::VE;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* cur = lds;                 // [Ce]    key frame's warped embedding
  float* fbuf = lds + Ce;           // [NT][Cf] warped features of every frame
  __shared__ float red[4];
  __shared__ float wts[64];
  const int C = Cf + Ce;
  const int nvec = C / VE, fvec = Cf / VE;
  const int px = blockIdx.x % W, py = blockIdx.x / W;
  const int tid = threadIdx.x;

  // bilinear taps of frame t at this pixel (grid_sample, align_corners=False, border padding)
  auto taps = [&](int t, int& x0, int& y0, int& x1, int& y1, float& wx, float& wy) {
    const float fx = flow[((size_t)t * 2 + 0) * H * W + py * W + px];
    const float fy = flow[((size_t)t * 2 + 1) * H * W + py * W + px];
    const float gx = ((float)px + fx) / ((float)(W - 1) / 2.f) - 1.f;   // :55-58
    const float gy = ((float)py + fy) / ((float)(H - 1) / 2.f) - 1.f;
    float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f;                      // unnormalise, align_corners = False
    float iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));                          // padding_mode = border
    iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
    const float flx = floorf(ix), fly = floorf(iy);
    x0 = (int)flx; y0 = (int)fly;
    x1 = min(x0 + 1, W - 1); y1 = min(y0 + 1, H - 1);
    wx = ix - flx; wy = iy - fly;
  };
  auto gather = [&](int t, int v, float (&val)[VE]) {
    int x0, y0, x1, y1; float wx, wy;
    taps(t, x0, y0, x1, y1, wx, wy);
    const T* base = feats + (size_t)t * H * W * C + (size_t)v * VE;
    const uint4 a = *reinterpret_cast<const uint4*>(base + ((size_t)y0 * W + x0) * C);
    const uint4 b = *reinterpret_cast<const uint4*>(base + ((size_t)y0 * W + x1) * C);
    const uint4 c = *reinterpret_cast<const uint4*>(base + ((size_t)y1 * W + x0) * C);
    const uint4 d = *reinterpret_cast<const uint4*>(base + ((size_t)y1 * W + x1) * C);
    const T* ea = reinterpret_cast<const T*>(&a); const T* eb = reinterpret_cast<const T*>(&b);
    const T* ec = reinterpret_cast<const T*>(&c); const T* ed = reinterpret_cast<const T*>(&d);
    const float w00 = (1.f - wx) * (1.f - wy), w01 = wx * (1.f - wy), w10 = (1.f - wx) * wy, w11 = wx * wy;
#pragma unroll
    for (int e = 0; e < VE; ++e)
      val[e] = w00 * Elem<T>::ld(ea + e) + w01 * Elem<T>::ld(eb + e) + w10 * Elem<T>::ld(ec + e) + w11 * Elem<T>::ld(ed + e);
  };

  // ---- A. the key frame's warped embedding and its norm
  float n2 = 0.f;
  for (int v = fvec + tid; v < nvec; v += 256) {
    float val[VE];
    gather(key, v, val);
#pragma unroll
    for (int e = 0; e < VE; ++e) { cur[(v - fvec) * VE + e] = val[e]; n2 += val[e] * val[e]; }
  }
  const float cur_norm = sqrtf(block_sum_256(n2, red)) + 1e-10f;   // compute_norm :64-65

  // ---- B. every frame: cosine weight from the embedding channels, warped features parked in LDS
  for (int t = 0; t < NT; ++t) {
    float dot = 0.f, nn = 0.f;
    for (int v = tid; v < nvec; v += 256) {
      float val[VE];
      gather(t, v, val);
      if (v < fvec) {
#pragma unroll
        for (int e = 0; e < VE; ++e) fbuf[(size_t)t * Cf + v * VE + e] = val[e];
      } else {
#pragma unroll
        for (int e = 0; e < VE; ++e) { dot += val[e] * cur[(v - fvec) * VE + e]; nn += val[e] * val[e]; }
      }
    }
    const float d = block_sum_256(dot, red);
    const float n = sqrtf(block_sum_256(nn, red)) + 1e-10f;
    if (tid == 0) wts[t] = d / (n * cur_norm);                      // compute_weight :67-76
  }
  __syncthreads();
  // ---- C. softmax over frames (:209) and the weighted sum (:211)
  if (tid == 0) {
    float m = -INFINITY, s = 0.f;
    for (int t = 0; t < NT; ++t) m = fmaxf(m, wts[t]);
    for (int t = 0; t < NT; ++t) { wts[t] = expf(wts[t] - m); s += wts[t]; }
    for (int t = 0; t < NT; ++t) wts[t] /= s;
  }
  __syncthreads();
  if (weights_out && tid < NT) weights_out[((size_t)tid * H + py) * W + px] = wts[tid];
  for (int c = tid; c < Cf; c += 256) {
    float acc = 0.f;
    for (int t = 0; t < NT; ++t) acc += wts[t] * fbuf[(size_t)t * Cf + c];
    Elem<T>::st(out + ((size_t)py * W + px) * Cf + c, acc);
  }
}


// DFF (generalized_rcnn_dff.py:41-60,:132-135): out = grid_sample(key_feats, flow grid, bilinear, border) * scale_map.
// One thread per (pixel, 16-byte channel vector): pure gather + multiply, HBM/L2-bound.
template <typename T>
__global__ __launch_bounds__(256) void dff_warp_scale_kernel(const T* __restrict__ feats, const float* __restrict__ flow,
                                                             const T* __restrict__ scale, T* __restrict__ out, int H,
                                                             int W, int C) {
  constexpr int VE = Elem<T>::VE;
  const int nvec = C / VE;
  const size_t total = (size_t)H * W * nvec;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int v = (int)(i % nvec);
    const int pix = (int)(i / nvec);
    const int px = pix % W, py = pix / W;
    const float fx = flow[(size_t)pix], fy = flow[(size_t)H * W + pix];
    const float gx = ((float)px + fx) / ((float)(W - 1) / 2.f) - 1.f;
    const float gy = ((float)py + fy) / ((float)(H - 1) / 2.f) - 1.f;
    float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f;
    float iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
    iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
    const float flx = floorf(ix), fly = floorf(iy);
    const int x0 = (int)flx, y0 = (int)fly, x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
    const float wx = ix - flx, wy = iy - fly;
    const float w00 = (1.f - wx) * (1.f - wy), w01 = wx * (1.f - wy), w10 = (1.f - wx) * wy, w11 = wx * wy;
    const T* base = feats + (size_t)v * VE;
    const uint4 a = *reinterpret_cast<const uint4*>(base + ((size_t)y0 * W + x0) * C);
    const uint4 b = *reinterpret_cast<const uint4*>(base + ((size_t)y0 * W + x1) * C);
    const uint4 c = *reinterpret_cast<const uint4*>(base + ((size_t)y1 * W + x0) * C);
    const uint4 d = *reinterpret_cast<const uint4*>(base + ((size_t)y1 * W + x1) * C);
    const uint4 sc = *reinterpret_cast<const uint4*>(scale + (size_t)pix * C + (size_t)v * VE);
    const T* ea = reinterpret_cast<const T*>(&a); const T* eb = reinterpret_cast<const T*>(&b);
    const T* ec = reinterpret_cast<const T*>(&c); const T* ed = reinterpret_cast<const T*>(&d);
    const T* es = reinterpret_cast<const T*>(&sc);
    T* o = out + (size_t)pix * C + (size_t)v * VE;
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      const float w = w00 * Elem<T>::ld(ea + e) + w01 * Elem<T>::ld(eb + e) + w10 * Elem<T>::ld(ec + e) +
                      w11 * Elem<T>::ld(ed + e);
      Elem<T>::st(o + e, w * Elem<T>::ld(es + e));
    }
  }
}

}  // namespace

extern "C" int mega_dff_warp_scale(const void* feats, const float* flow, const void* scale, void* out, int H, int W,
                                   int C, int dtype, void* stream) {
  mega_clear_error();
  if (!feats || !flow || !scale || !out || H <= 1 || W <= 1 || C <= 0) return MEGA_ERR_ARG;
  const int ve = dtype == MEGA_BF16 ? 8 : 4;
  if (C % ve) return MEGA_ERR_ARG;
  const size_t total = (size_t)H * W * (C / ve);
  const dim3 grid((unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MEGA_BF16)
    hipLaunchKernelGGL((dff_warp_scale_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)feats, flow,
                       (const bf16_t*)scale, (bf16_t*)out, H, W, C);
  else if (dtype == MEGA_F32)
    hipLaunchKernelGGL((dff_warp_scale_kernel<float>), grid, dim3(256), 0, st, (const float*)feats, flow,
                       (const float*)scale, (float*)out, H, W, C);
  else
    return MEGA_ERR_ARG;
  return mega_check_launch();
}

extern "C" int mega_fgfa_warp_aggregate(const void* feats, const float* flow, void* out, float* weights_out, int T,
                                        int H, int W, int Cf, int Ce, int key, int dtype, void* stream) {
  mega_clear_error();
  if (!feats || !flow || !out || T <= 0 || T > 64 || H <= 1 || W <= 1 || Cf <= 0 || Ce <= 0 || key < 0 || key >= T)
    return MEGA_ERR_ARG;
  const int ve = dtype == MEGA_BF16 ? 8 : 4;
  if (Cf % ve || Ce % ve) return MEGA_ERR_ARG;
  const size_t smem = ((size_t)Ce + (size_t)T * Cf) * sizeof(float);
  if (smem > 150 * 1024) return MEGA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MEGA_BF16) {
    (void)hipFuncSetAttribute((const void*)fgfa_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((fgfa_kernel<bf16_t>), dim3(H * W), dim3(256), smem, st, (const bf16_t*)feats, flow,
                       (bf16_t*)out, weights_out, T, H, W, Cf, Ce, key);
  } else if (dtype == MEGA_F32) {
    (void)hipFuncSetAttribute((const void*)fgfa_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((fgfa_kernel<float>), dim3(H * W), dim3(256), smem, st, (const float*)feats, flow, (float*)out,
                       weights_out, T, H, W, Cf, Ce, key);
  } else {
    return MEGA_ERR_ARG;
  }
  return mega_check_launch();
}
