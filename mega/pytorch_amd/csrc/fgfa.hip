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
#include <cstdlib>

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


// Two-pass form of the same computation (round 3).  fgfa_kernel above walks the T frames one after the other with two
// block-wide reductions each: a chain of 21 dependent (gather -> barrier -> barrier) steps per pixel, one block per CU
// (94 KB of LDS): 588 GB/s.  Here nothing waits on a block barrier inside the frame loops:
//   pass 1  the 4 waves take the frames round-robin; a wave gathers one frame's warped EMBEDDING (Ce channels, 16 loads
//           per lane in flight), reduces dot / norm with shuffles and writes that frame's cosine weight;
//   pass 2  thread (frame group g, channel vector v) accumulates sum_t w_t * warp_t(features)[v] over its frames with
//           the taps of two frames in flight; the groups' partial sums meet in LDS.
// 12 KB of LDS -> 8 blocks per CU hide the gather latency; blocks are dealt to the XCDs in contiguous pixel bands so
// that an XCD's L2 only ever holds its own band (+ the flow-displaced halo) of the T feature maps.
template <typename T>
__global__ __launch_bounds__(256) void fgfa2_kernel(const T* __restrict__ feats, const float* __restrict__ flow,
                                                    T* __restrict__ out, float* __restrict__ weights_out, int NT,
                                                    int H, int W, int Cf, int Ce, int key,
                                                    const int* __restrict__ order, int flow_key_pos) {
  // order (optional, device): the T maps live in a RING -- order[0] = slot of the key frame, order[1 + t] = slot of the
  // frame at window position t.  Frames are visited in window order whatever their slots, so the sums have the bits of
  // the contiguous (deque-ordered) call; flow is indexed by slot like feats (flow_key_pos < 0) or by WINDOW POSITION
  // (flow_key_pos >= 0 = the key frame's position: flow [NT][2][H][W] holds exactly the window's pairs, in window order).
  // blockIdx.y = key frame of a batch (mega_fgfa_warp_aggregate_ring_pos with G > 1): its own index-table row, its own T flow
  // fields (window order), its own output map; the feature ring is shared
  if (gridDim.y > 1) {
    const size_t gb = blockIdx.y;
    order += gb * (size_t)(NT + 1);
    flow += gb * (size_t)NT * 2 * H * W;
    out += gb * (size_t)H * W * Cf;
    if (weights_out) weights_out += gb * (size_t)NT * H * W;
  }
  auto slot_of = [&](int t) { return order ? order[1 + t] : t; };
  auto flow_of = [&](int t, int slot) { return flow_key_pos >= 0 ? t : slot; };
  const int key_flow = (order && flow_key_pos >= 0) ? flow_key_pos : (order ? order[0] : key);
  if (order) key = order[0];
  constexpr int VE = Elem<T>::VE;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* cur = lds;                 // [Ce]  key frame's warped embedding
  float* part = lds + Ce;           // [G - 1][Cf] partial sums of frame groups 1..G-1
  __shared__ float red[4];
  __shared__ float wts[64];
  const int C = Cf + Ce;
  const int fvec = Cf / VE, evec = Ce / VE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int pix;
  {                                 // XCD-aware block -> pixel map (bijective): XCD x owns a contiguous band of pixels
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
    pix = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int px = pix % W, py = pix / W;

  struct Taps { int o00, o01, o10, o11; float w00, w01, w10, w11; };
  auto taps = [&](int t) {           // t = index of the frame's flow field (its slot, or its window position)
    const float fx = flow[((size_t)t * 2 + 0) * H * W + py * W + px];
    const float fy = flow[((size_t)t * 2 + 1) * H * W + py * W + px];
    const float gx = ((float)px + fx) / ((float)(W - 1) / 2.f) - 1.f;   // :55-58
    const float gy = ((float)py + fy) / ((float)(H - 1) / 2.f) - 1.f;
    float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f;                      // unnormalise, align_corners = False
    float iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));                          // padding_mode = border
    iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
    const float flx = floorf(ix), fly = floorf(iy);
    const int x0 = (int)flx, y0 = (int)fly, x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
    const float wx = ix - flx, wy = iy - fly;
    Taps tp;
    tp.o00 = (y0 * W + x0) * C; tp.o01 = (y0 * W + x1) * C; tp.o10 = (y1 * W + x0) * C; tp.o11 = (y1 * W + x1) * C;
    tp.w00 = (1.f - wx) * (1.f - wy); tp.w01 = wx * (1.f - wy); tp.w10 = (1.f - wx) * wy; tp.w11 = wx * wy;
    return tp;
  };
  auto blend = [&](const Taps& tp, const uint4& a, const uint4& b, const uint4& c, const uint4& d, float (&val)[VE]) {
    const T* ea = reinterpret_cast<const T*>(&a); const T* eb = reinterpret_cast<const T*>(&b);
    const T* ec = reinterpret_cast<const T*>(&c); const T* ed = reinterpret_cast<const T*>(&d);
#pragma unroll
    for (int e = 0; e < VE; ++e)
      val[e] = tp.w00 * Elem<T>::ld(ea + e) + tp.w01 * Elem<T>::ld(eb + e) + tp.w10 * Elem<T>::ld(ec + e) +
               tp.w11 * Elem<T>::ld(ed + e);
  };

  // ---- A. the key frame's warped embedding (all 256 threads) and its norm
  {
    const Taps tp = taps(key_flow);
    const T* base = feats + (size_t)key * H * W * C + Cf;
    float n2 = 0.f;
    for (int v = tid; v < evec; v += 256) {
      const T* p = base + (size_t)v * VE;
      const uint4 a = *reinterpret_cast<const uint4*>(p + tp.o00), b = *reinterpret_cast<const uint4*>(p + tp.o01);
      const uint4 c = *reinterpret_cast<const uint4*>(p + tp.o10), d = *reinterpret_cast<const uint4*>(p + tp.o11);
      float val[VE];
      blend(tp, a, b, c, d, val);
#pragma unroll
      for (int e = 0; e < VE; ++e) { cur[v * VE + e] = val[e]; n2 += val[e] * val[e]; }
    }
    const float tot = block_sum_256(n2, red);          // (ends with a barrier: cur[] is visible to every wave)
    if (tid == 0) wts[63] = sqrtf(tot) + 1e-10f;       // compute_norm :64-65
  }
  __syncthreads();
  const float cur_norm = wts[63];

  // ---- B. cosine weights: wave w takes frames w, w + 4, ...; no block barrier inside
  for (int t = wave; t < NT; t += 4) {
    const int st = slot_of(t);
    const Taps tp = taps(flow_of(t, st));
    const T* base = feats + (size_t)st * H * W * C + Cf;
    float dot = 0.f, nn = 0.f;
    for (int v0 = 0; v0 < evec; v0 += 256) {           // 4 vectors per lane per round: 16 loads in flight
      uint4 a[4], b[4], c[4], d[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int v = min(v0 + j * 64 + lane, evec - 1);
        const T* p = base + (size_t)v * VE;
        a[j] = *reinterpret_cast<const uint4*>(p + tp.o00); b[j] = *reinterpret_cast<const uint4*>(p + tp.o01);
        c[j] = *reinterpret_cast<const uint4*>(p + tp.o10); d[j] = *reinterpret_cast<const uint4*>(p + tp.o11);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int v = v0 + j * 64 + lane;
        if (v < evec) {
          float val[VE];
          blend(tp, a[j], b[j], c[j], d[j], val);
#pragma unroll
          for (int e = 0; e < VE; ++e) { dot += val[e] * cur[v * VE + e]; nn += val[e] * val[e]; }
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { dot += __shfl_xor(dot, o); nn += __shfl_xor(nn, o); }
    if (lane == 0) wts[t] = dot / ((sqrtf(nn) + 1e-10f) * cur_norm);       // compute_weight :67-76
  }
  __syncthreads();
  // ---- C. softmax over frames (:209)
  if (tid == 0) {
    float m = -INFINITY, s2 = 0.f;
    for (int t = 0; t < NT; ++t) m = fmaxf(m, wts[t]);
    for (int t = 0; t < NT; ++t) { wts[t] = expf(wts[t] - m); s2 += wts[t]; }
    for (int t = 0; t < NT; ++t) wts[t] /= s2;
  }
  __syncthreads();
  if (weights_out && tid < NT) weights_out[((size_t)tid * H + py) * W + px] = wts[tid];
  // ---- D. the weighted sum (:211): thread (group g, vector v) over frames g, g + G, ..., two frames in flight
  const int G = 256 / fvec;                            // host guarantees fvec <= 256 and 256 % fvec == 0
  const int g = tid / fvec, v = tid - g * fvec;
  float acc[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) acc[e] = 0.f;
  for (int t = g; t < NT; t += 2 * G) {
    const int t2 = t + G;
    const bool has2 = t2 < NT;
    const int s1 = slot_of(t), s2 = slot_of(has2 ? t2 : t);
    const Taps tp = taps(flow_of(t, s1)), tq = taps(flow_of(has2 ? t2 : t, s2));
    const T* p = feats + (size_t)s1 * H * W * C + (size_t)v * VE;
    const T* q2 = feats + (size_t)s2 * H * W * C + (size_t)v * VE;
    const uint4 a = *reinterpret_cast<const uint4*>(p + tp.o00), b = *reinterpret_cast<const uint4*>(p + tp.o01);
    const uint4 c = *reinterpret_cast<const uint4*>(p + tp.o10), d = *reinterpret_cast<const uint4*>(p + tp.o11);
    const uint4 a2 = *reinterpret_cast<const uint4*>(q2 + tq.o00), b2 = *reinterpret_cast<const uint4*>(q2 + tq.o01);
    const uint4 c2 = *reinterpret_cast<const uint4*>(q2 + tq.o10), d2 = *reinterpret_cast<const uint4*>(q2 + tq.o11);
    float val[VE];
    blend(tp, a, b, c, d, val);
    const float w1 = wts[t];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] += w1 * val[e];
    if (has2) {
      blend(tq, a2, b2, c2, d2, val);
      const float w2 = wts[t2];
#pragma unroll
      for (int e = 0; e < VE; ++e) acc[e] += w2 * val[e];
    }
  }
  if (g > 0) {
#pragma unroll
    for (int e = 0; e < VE; ++e) part[(size_t)(g - 1) * Cf + v * VE + e] = acc[e];
  }
  __syncthreads();
  if (g == 0) {
    for (int gg = 1; gg < G; ++gg)
#pragma unroll
      for (int e = 0; e < VE; ++e) acc[e] += part[(size_t)(gg - 1) * Cf + v * VE + e];
    uint4 o;
    T* oe = reinterpret_cast<T*>(&o);
#pragma unroll
    for (int e = 0; e < VE; ++e) Elem<T>::st(oe + e, acc[e]);
    *reinterpret_cast<uint4*>(out + ((size_t)py * W + px) * Cf + (size_t)v * VE) = o;
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


// ---- FlowNetS input (round 6): image pairs -> the first conv's operand in ONE kernel.
// generalized_rcnn_fgfa.py:196-198 builds cat([cur, ref], 1) / 255 for the T window frames and flownet.py:52,:56 average-pools
// it 2 x 2 before the 7 x 7 / stride-2 `flow_conv1` (6 -> 64 channels).  As separate steps on this path that was a 302 MB f32
// concatenation, a permute + cast, a pad to 8 channels, the pool, a pad to the GEMM's 64-channel K vector (403 MB) and a conv
// whose K = 49 x 64 carries 10.7x the real 49 x 6 products: ~1.5 ms of a 5.6 ms key frame.  Here one kernel reads the f32 NCHW
// frames once and writes, per pooled pixel (t, h, w), the SEVEN horizontal taps of the conv as one 128-byte row:
//   out[t][3 + h][w][s * 8 + c] = pool(pair)[t][h][w - 3 + s][c]   (c < 6: cur's 3 then the frame's 3 channels; 0 elsewhere),
// with three zero rows above and below, so that flow_conv1 is a 7 x 1 convolution (stride 2, pad 0) over 64 channels: K = 448.
// Arithmetic = the steps it replaces: each f32 pixel rounded to the 16-bit type, the in-bounds taps of a 2 x 2 window summed in
// f32 in (dy, dx) order and divided by their count (nn.AvgPool2d(2, 2, ceil_mode=True)), rounded once.
template <typename HT>
__global__ __launch_bounds__(256) void pair_taps_kernel(const float* __restrict__ ring, long long ring_stride,
                                                        const float* __restrict__ cur, long long cur_stride,
                                                        const int* __restrict__ order, HT* __restrict__ out, int H, int W,
                                                        int Hp, int Wp) {
  constexpr int SEG = 250;
  __shared__ uint4 px[SEG + 6];
  const int w0 = blockIdx.x * SEG, row = blockIdx.y, t = blockIdx.z;
  const int tid = threadIdx.x;
  const int npx = min(SEG, Wp - w0);
  uint4* orow = reinterpret_cast<uint4*>(out + (((size_t)t * (Hp + 6) + row) * Wp + w0) * 64);
  const int h = row - 3;
  const uint4 zero = make_uint4(0, 0, 0, 0);
  if (h < 0 || h >= Hp) {                                // the conv's vertical padding
    for (int i = tid; i < npx * 8; i += 256) orow[i] = zero;
    return;
  }
  const float* cb = cur ? cur + (size_t)t * cur_stride : ring + (size_t)order[0] * ring_stride;
  const float* rb = ring + (size_t)t * ring_stride;
  if (tid < SEG + 6) {
    const int w = w0 - 3 + tid;
    uint4 v = zero;
    if (w >= 0 && w < Wp) {
      float a[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const float* plane = (c < 3 ? cb + (size_t)c * H * W : rb + (size_t)(c - 3) * H * W);
        float acc = 0.f;
        int cnt = 0;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const int hi = 2 * h + dy;
          if (hi >= H) continue;
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const int wi = 2 * w + dx;
            if (wi >= W) continue;
            acc += Half16<HT>::one(Half16<HT>::cvt(plane[(size_t)hi * W + wi]));
            ++cnt;
          }
        }
        a[c] = acc / (float)cnt;
      }
      v = make_uint4(Half16<HT>::pack2(a[0], a[1]), Half16<HT>::pack2(a[2], a[3]), Half16<HT>::pack2(a[4], a[5]), 0u);
    }
    px[tid] = v;
  }
  __syncthreads();
  for (int i = tid; i < npx * 8; i += 256) {             // 16-byte chunk s of pixel p: tap s (chunk 7: the K pad)
    const int p_ = i >> 3, s_ = i & 7;
    orow[i] = s_ < 7 ? px[p_ + s_] : zero;
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

static int fgfa_impl(const void* feats, const float* flow, void* out, float* weights_out, int T, int H, int W, int Cf,
                     int Ce, int key, const int* order, int dtype, void* stream, int flow_key_pos = -1, int G = 1);

extern "C" int mega_fgfa_warp_aggregate(const void* feats, const float* flow, void* out, float* weights_out, int T,
                                        int H, int W, int Cf, int Ce, int key, int dtype, void* stream) {
  return fgfa_impl(feats, flow, out, weights_out, T, H, W, Cf, Ce, key, nullptr, dtype, stream);
}

// The same with the T maps (and their flow fields) held in a ring: order [1 + T] device ints, order[0] = slot of the key
// frame, order[1 + t] = slot of window position t.  Same bits as the contiguous call on the frames in window order.
extern "C" int mega_fgfa_warp_aggregate_ring(const void* feats, const float* flow, void* out, float* weights_out, int T,
                                             int H, int W, int Cf, int Ce, const int* order, int dtype, void* stream) {
  if (!order) return MEGA_ERR_ARG;
  return fgfa_impl(feats, flow, out, weights_out, T, H, W, Cf, Ce, 0, order, dtype, stream);
}

// The ring form with the flow fields in WINDOW order: flow [T][2][H][W] = the T pairs (key frame, window position t) and
// nothing else (the engine's FlowNetS pass over several key frames enumerates exactly each key frame's pairs); key_pos = the
// key frame's window position (cfg KEY_FRAME_LOCATION).  Same arithmetic, same order of the sums.
extern "C" int mega_fgfa_warp_aggregate_ring_pos(const void* feats, const float* flow, void* out, float* weights_out, int T,
                                                 int H, int W, int Cf, int Ce, const int* order, int key_pos, int dtype,
                                                 void* stream) {
  if (!order || key_pos < 0 || key_pos >= T) return MEGA_ERR_ARG;
  return fgfa_impl(feats, flow, out, weights_out, T, H, W, Cf, Ce, 0, order, dtype, stream, key_pos);
}

// ... for G key frames in ONE launch (the engine's group): order [G][1 + T], flow [G][T][2][H][W], out [G][H][W][Cf]
// (weights_out [G][T][H][W] or NULL); the feature ring is shared.  One key frame is 2394 blocks on 256 CUs x 8 resident
// blocks = 1.17 rounds; G of them in one grid run without the per-launch tail.  Same bits per key frame.
extern "C" int mega_fgfa_warp_aggregate_ring_pos_batched(const void* feats, const float* flow, void* out, float* weights_out,
                                                         int T, int H, int W, int Cf, int Ce, const int* order, int key_pos,
                                                         int G, int dtype, void* stream) {
  if (!order || key_pos < 0 || key_pos >= T || G < 1 || G > 65535) return MEGA_ERR_ARG;
  return fgfa_impl(feats, flow, out, weights_out, T, H, W, Cf, Ce, 0, order, dtype, stream, key_pos, G);
}

static int fgfa_impl(const void* feats, const float* flow, void* out, float* weights_out, int T, int H, int W, int Cf,
                     int Ce, int key, const int* order, int dtype, void* stream, int flow_key_pos, int G) {
  mega_clear_error();
  if (!feats || !flow || !out || T <= 0 || T > 64 || H <= 1 || W <= 1 || Cf <= 0 || Ce <= 0 || key < 0 || key >= T)
    return MEGA_ERR_ARG;
  const int ve = dtype == MEGA_BF16 ? 8 : 4;
  if (Cf % ve || Ce % ve) return MEGA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  static const bool legacy = getenv("MEGA_FGFA_LEGACY") != nullptr;      // A/B switch: the one-pass kernel of rounds 1-2
  const int fvec = Cf / ve;
  if (!legacy && fvec <= 256 && 256 % fvec == 0 && T < 63 && (size_t)H * W * (Cf + Ce) < 0x7FFFFFFFull) {
    const size_t smem2 = ((size_t)Ce + (size_t)(256 / fvec - 1) * Cf) * sizeof(float);
    if (dtype == MEGA_BF16)
      hipLaunchKernelGGL((fgfa2_kernel<bf16_t>), dim3(H * W, G), dim3(256), smem2, st, (const bf16_t*)feats, flow,
                         (bf16_t*)out, weights_out, T, H, W, Cf, Ce, key, order, flow_key_pos);
    else if (dtype == MEGA_F32)
      hipLaunchKernelGGL((fgfa2_kernel<float>), dim3(H * W, G), dim3(256), smem2, st, (const float*)feats, flow,
                         (float*)out, weights_out, T, H, W, Cf, Ce, key, order, flow_key_pos);
    else
      return MEGA_ERR_ARG;
    return mega_check_launch();
  }
  if (order || G != 1) return MEGA_ERR_ARG;          // the ring / batched forms exist for the two-pass kernel only
  const size_t smem = ((size_t)Ce + (size_t)T * Cf) * sizeof(float);
  if (smem > 150 * 1024) return MEGA_ERR_ARG;
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

/* see pair_taps_kernel.  ring: f32 [T][3][H][W] (ring_stride elements between frames); cur: the key frame(s), f32 [.][3][H][W]
 * with cur_stride elements between the T pairs' key frames (0: one key frame for all), or NULL: the key frame is ring slot
 * order[0] (device int).  out: [T][ceil(H/2) + 6][ceil(W/2)][64] of the 16-bit dtype. */
extern "C" int mega_fgfa_pair_taps(const float* ring, long long ring_stride, const float* cur, long long cur_stride,
                                   const int* order, void* out, int T, int H, int W, int dtype, void* stream) {
  mega_clear_error();
  if (!ring || !out || (!cur && !order) || T <= 0 || H <= 0 || W <= 0 || T > 65535) return MEGA_ERR_ARG;
  const int Hp = (H + 1) / 2, Wp = (W + 1) / 2;
  if (Hp + 6 > 65535 || (reinterpret_cast<size_t>(out) & 15)) return MEGA_ERR_ARG;
  const dim3 grid((unsigned)((Wp + 249) / 250), (unsigned)(Hp + 6), (unsigned)T);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MEGA_BF16)
    hipLaunchKernelGGL((pair_taps_kernel<bf16_t>), grid, dim3(256), 0, st, ring, ring_stride, cur, cur_stride, order, (bf16_t*)out, H, W, Hp, Wp);
  else if (dtype == MEGA_F16)
    hipLaunchKernelGGL((pair_taps_kernel<f16_t>), grid, dim3(256), 0, st, ring, ring_stride, cur, cur_stride, order, (f16_t*)out, H, W, Hp, Wp);
  else
    return MEGA_ERR_ARG;
  return mega_check_launch();
}

// ---- FlowNetS refinement level (flownet.py:94-111): concatN = cat([skip, crop(leaky(deconvN(x))), crop(upsample_flowN(flow))], 1)
// zero-padded to the GEMM's K vector.  The deconvolution's sub-pixel conv (mega_conv2d_nhwc_subpixel) writes its slice of the
// concatenation itself; this kernel writes the rest in one pass: the skip connection's channels [0, Cs), the 2-channel
// ConvTranspose2d(2, 2, 4, stride 2) of the coarse flow at [Cs + C, Cs + C + 2) -- 16 FMAs per pixel in f32 on the f32 weights:
// out pixel (Y, X) = (y + crop, x + crop) of the full map sees coarse pixels (Y / 2 - dy, X / 2 - dx) through taps
// (Y % 2 + 2 dy, X % 2 + 2 dx) -- and zeros up to the pixel stride.  Replaces a GEMM on 64-padded channels over a zero-stuffed
// map, two crops, a cat and a pad (five launches, ~0.15 ms per level of the 21-pair key frame).
namespace {
template <typename T>
__global__ __launch_bounds__(256) void flow_level_assemble_kernel(const T* __restrict__ skip, const T* __restrict__ flow,
                                                                  const float* __restrict__ wup, const float* __restrict__ bup,
                                                                  T* __restrict__ out, int N, int H2, int W2, int Cs, int C, int ldo,
                                                                  int h, int w, int crop) {
  constexpr int VE = Elem<T>::VE;
  const int vs = Cs / VE, vt = (ldo - Cs - C) / VE, vpp = vs + vt;
  const long long total = (long long)N * H2 * W2 * vpp;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long px = i / vpp;
    const int v = (int)(i - px * vpp);
    T* o = out + px * ldo;
    if (v < vs) {
      *reinterpret_cast<uint4*>(o + v * VE) = *reinterpret_cast<const uint4*>(skip + px * Cs + v * VE);
      continue;
    }
    uint4 z = make_uint4(0u, 0u, 0u, 0u);
    if (v == vs) {
      const int x = (int)(px % W2), y = (int)((px / W2) % H2), t = (int)(px / ((long long)W2 * H2));
      const int Y = y + crop, X = x + crop, a = Y & 1, b = X & 1, m = Y >> 1, n = X >> 1;
      float u0 = bup[0], u1 = bup[1];
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int mi = m - dy, ni = n - dx;
          if ((unsigned)mi >= (unsigned)h || (unsigned)ni >= (unsigned)w) continue;
          const T* f = flow + ((long long)(t * h + mi) * w + ni) * 2;
          const float f0 = Elem<T>::ld(f), f1 = Elem<T>::ld(f + 1);
          const int tap = (a + 2 * dy) * 4 + (b + 2 * dx);          // w[ci][co][kh][kw]
          u0 = fmaf(f0, wup[0 * 32 + 0 * 16 + tap], u0);
          u0 = fmaf(f1, wup[1 * 32 + 0 * 16 + tap], u0);
          u1 = fmaf(f0, wup[0 * 32 + 1 * 16 + tap], u1);
          u1 = fmaf(f1, wup[1 * 32 + 1 * 16 + tap], u1);
        }
      T* ze = reinterpret_cast<T*>(&z);
      T e0, e1;
      Elem<T>::st(&e0, u0);
      Elem<T>::st(&e1, u1);
      ze[0] = e0; ze[1] = e1;
    }
    *reinterpret_cast<uint4*>(o + Cs + C + (v - vs) * VE) = z;
  }
}
}  // namespace

/* see flow_level_assemble_kernel.  skip [N][H2][W2][Cs], flow [N][h][w][2], out [N][H2][W2][ldo] of `dtype` (MEGA_F32 / BF16 /
 * F16); w_up f32 [2][2][4][4] (ConvTranspose2d weight, [in][out][kh][kw]), b_up f32 [2]; Cs, C, ldo multiples of the 16-byte
 * vector, ldo >= Cs + C + 2; crop: rows / columns of the full (2h + 2) x (2w + 2) map dropped at the top / left. */
extern "C" int mega_flow_level_assemble(const void* skip, const void* flow, const float* w_up, const float* b_up, void* out,
                                        int N, int H2, int W2, int Cs, int C, int ldo, int h, int w, int crop, int dtype,
                                        void* stream) {
  mega_clear_error();
  if (!skip || !flow || !w_up || !b_up || !out || N <= 0 || H2 <= 0 || W2 <= 0 || h <= 0 || w <= 0 || crop < 0) return MEGA_ERR_ARG;
  const int ve = dtype == MEGA_F32 ? 4 : 8;
  if ((dtype != MEGA_F32 && dtype != MEGA_BF16 && dtype != MEGA_F16) || Cs <= 0 || C <= 0 || Cs % ve || C % ve || ldo % ve ||
      ldo < Cs + C + 2 || H2 + crop > 2 * h + 2 || W2 + crop > 2 * w + 2)
    return MEGA_ERR_ARG;
  if ((reinterpret_cast<size_t>(out) & 15) || (reinterpret_cast<size_t>(skip) & 15)) return MEGA_ERR_ARG;
  const long long total = (long long)N * H2 * W2 * ((Cs + (ldo - Cs - C)) / ve);
  const unsigned blocks = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MEGA_BF16)
    hipLaunchKernelGGL((flow_level_assemble_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)skip, (const bf16_t*)flow,
                       w_up, b_up, (bf16_t*)out, N, H2, W2, Cs, C, ldo, h, w, crop);
  else if (dtype == MEGA_F16)
    hipLaunchKernelGGL((flow_level_assemble_kernel<f16_t>), dim3(blocks), dim3(256), 0, st, (const f16_t*)skip, (const f16_t*)flow,
                       w_up, b_up, (f16_t*)out, N, H2, W2, Cs, C, ldo, h, w, crop);
  else
    hipLaunchKernelGGL((flow_level_assemble_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)skip, (const float*)flow,
                       w_up, b_up, (float*)out, N, H2, W2, Cs, C, ldo, h, w, crop);
  return mega_check_launch();
}

// ---- FlowNetS flow prediction (flownet.py:40-52 Convolution1..5: nn.Conv2d(Cin, 2, 3, padding=1)), second half.  A 3 x 3 conv
// with TWO output channels on 64-wide GEMM tiles spends 32x its products on padding columns; by linearity it is a 1 x 1 conv
// with 18 columns -- z[p][(r*3 + s)*2 + c] = sum_ci x[p][ci] w[c][ci][r][s], ONE pass over x, K = Cin instead of 9 Cin, f32
// output (mega_conv2d_nhwc) -- followed by this kernel: flow[t][y][x][c] = (sum_{r,s} z[t][y+r-1][x+s-1][(r*3+s)*2 + c]) * scale
// + bias[c], taps outside the map skipped (zero padding), summed in f32 in (r, s) order.
namespace {
template <typename OT>
__global__ __launch_bounds__(256) void flow_pred_finish_kernel(const float* __restrict__ z, int ldz, const float* __restrict__ bias,
                                                               float scale, OT* __restrict__ out, int N, int H, int W) {
  const long long total = (long long)N * H * W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int yy = y + r - 1, xx = x + s - 1;
        if ((unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W) continue;
        const float2 v = *reinterpret_cast<const float2*>(z + (i + (long long)(r - 1) * W + (s - 1)) * ldz + (r * 3 + s) * 2);
        a0 += v.x;
        a1 += v.y;
      }
    Elem<OT>::st(out + i * 2, fmaf(a0, scale, bias[0]));
    Elem<OT>::st(out + i * 2 + 1, fmaf(a1, scale, bias[1]));
  }
}
}  // namespace

/* see flow_pred_finish_kernel.  z f32 [N][H][W][ldz] (ldz >= 18, even), bias f32 [2] (already multiplied by `scale` if the
 * caller wants (conv + b) * scale), out [N][H][W][2] of out_dtype (MEGA_F32 / BF16 / F16). */
extern "C" int mega_flow_pred_finish(const float* z, int ldz, const float* bias, float scale, void* out, int N, int H, int W,
                                     int out_dtype, void* stream) {
  mega_clear_error();
  if (!z || !bias || !out || N <= 0 || H <= 0 || W <= 0 || ldz < 18 || (ldz & 1) || (reinterpret_cast<size_t>(z) & 7)) return MEGA_ERR_ARG;
  const long long total = (long long)N * H * W;
  const unsigned blocks = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == MEGA_F32) hipLaunchKernelGGL((flow_pred_finish_kernel<float>), dim3(blocks), dim3(256), 0, st, z, ldz, bias, scale, (float*)out, N, H, W);
  else if (out_dtype == MEGA_BF16) hipLaunchKernelGGL((flow_pred_finish_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, z, ldz, bias, scale, (bf16_t*)out, N, H, W);
  else if (out_dtype == MEGA_F16) hipLaunchKernelGGL((flow_pred_finish_kernel<f16_t>), dim3(blocks), dim3(256), 0, st, z, ldz, bias, scale, (f16_t*)out, N, H, W);
  else return MEGA_ERR_ARG;
  return mega_check_launch();
}

// ---- FlowNetS flow_conv1 per FRAME instead of per pair (round 6).  flow_conv1 (flownet.py:52,:56: Conv2d(6, 64, 7, stride 2) +
// LeakyReLU(0.1)) is linear before its activation, and its input is cat([key, frame_t]): conv(pair) = conv_key(key frame) +
// conv_ref(frame_t).  Both halves depend on ONE frame, so they are computed once when a frame enters the window (a conv with
// 128 output channels [A | B] over that frame's tap operand, f32 out) instead of 21 times per key frame; what is left per key
// frame is this kernel:  out[t][p][c] = leaky(A[key][p][c] + B[t][p][c] + bias[c]),  one 16-byte vector of 8 channels per thread.
// ab f32 [S][P][128] (A = channels 0..63, B = 64..127); key: slot of the key frame (order ? order[0] : key).
namespace {
template <typename HT>
__global__ __launch_bounds__(256) void flow_conv1_combine_kernel(const float* __restrict__ ab, const float* __restrict__ bias,
                                                                 const int* __restrict__ order, int key, unsigned short* __restrict__ out,
                                                                 int T, long long P, int nwin) {
  // nwin == 0: pair t = (key frame, the frame in slot t), key slot = order ? order[0] : key.
  // nwin > 0:  order = [G][1 + nwin] rows [key slot, slot of window position 0 .. nwin-1]; pair q = g nwin + t = (key frame of
  //            row g, window position t): exactly the pairs of G key frames, in window order (T = G nwin pairs).
  const long long total = (long long)T * P * 8;
  const int ks0 = order ? order[0] : key;
  const int v = threadIdx.x & 7;
  float bi[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bi[e] = bias[v * 8 + e];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long px = i >> 3;                 // q * P + p
    const long long p = px % P;
    int ks = ks0;
    long long bs = px - p;                       // (slot of the pair's second frame) * P
    if (nwin > 0) {
      const int q = (int)(px / P), g = q / nwin, t = q - g * nwin;
      ks = order[g * (nwin + 1)];
      bs = (long long)order[g * (nwin + 1) + 1 + t] * P;
    }
    const float4* a = reinterpret_cast<const float4*>(ab + ((long long)ks * P + p) * 128 + v * 8);
    const float4* b = reinterpret_cast<const float4*>(ab + (bs + p) * 128 + 64 + v * 8);
    const float4 a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
    float x[8] = {a0.x + b0.x, a0.y + b0.y, a0.z + b0.z, a0.w + b0.w, a1.x + b1.x, a1.y + b1.y, a1.z + b1.z, a1.w + b1.w};
    u32x4_t o;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      float y0 = x[2 * d] + bi[2 * d], y1 = x[2 * d + 1] + bi[2 * d + 1];
      y0 = y0 > 0.f ? y0 : y0 * 0.1f;
      y1 = y1 > 0.f ? y1 : y1 * 0.1f;
      o[d] = Half16<HT>::pack2(y0, y1);
    }
    *reinterpret_cast<u32x4_t*>(out + px * 64 + v * 8) = o;
  }
}
}  // namespace

/* see flow_conv1_combine_kernel.  ab f32 [S][P][128] with S >= T frames' [A | B] halves (mega_conv2d_nhwc, f32 output);
 * bias f32 [64]; order (device int, NULL: use `key`): order[0] = slot of the key frame; out [T][P][64] of dtype
 * (MEGA_BF16 / MEGA_F16): pair t = (key frame, frame in slot t).  nwin > 0: order = [G][1 + nwin] rows [key slot, slots of the
 * window positions], T = G nwin, pair g nwin + t = (key frame of row g, the frame at window position t). */
extern "C" int mega_flow_conv1_combine(const float* ab, const float* bias, const int* order, int key, void* out, int T,
                                       long long P, int nwin, int dtype, void* stream) {
  mega_clear_error();
  if (!ab || !bias || !out || T <= 0 || P <= 0 || (!order && key < 0) || (reinterpret_cast<size_t>(ab) & 15) ||
      (reinterpret_cast<size_t>(out) & 15) || nwin < 0 || (nwin > 0 && (!order || T % nwin != 0)))
    return MEGA_ERR_ARG;
  const long long total = (long long)T * P * 8;
  const unsigned blocks = (unsigned)((total + 255) / 256 > 32768 ? 32768 : (total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MEGA_BF16)
    hipLaunchKernelGGL((flow_conv1_combine_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, ab, bias, order, key, (unsigned short*)out, T, P, nwin);
  else if (dtype == MEGA_F16)
    hipLaunchKernelGGL((flow_conv1_combine_kernel<f16_t>), dim3(blocks), dim3(256), 0, st, ab, bias, order, key, (unsigned short*)out, T, P, nwin);
  else
    return MEGA_ERR_ARG;
  return mega_check_launch();
}
