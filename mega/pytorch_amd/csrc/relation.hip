// MEGA relation module (multi-head relation attention over proposal features).
//
// Reference: mega_core/modeling/roi_heads/box_head/roi_box_feature_extractors.py
//   :147-176 extract_position_matrix, :126-144 extract_position_embedding, :240-250 cal_position_embedding
//   :567-646 MEGAFeatureExtractor.attention_module_multi_head
//
//   aff[q,h,k]  = ((q_h + u_h) . k_h) / sqrt(64)            (:613-627; u folded into the Q projection bias)
//   w[q,h,k]    = log(relu(Wg_h . pe(q,k) + bg_h) + 1e-6)   (:593-597,:630)  -- "local"/"memory" only
//   P           = softmax_k(aff + w)                         (:633)
//   out[q, h*64+j] = sum_k P[q,h,k] * (V Wv_h^T)[k, j] + bv  (:638-644; the grouped 1x1 conv Wv is applied to
//                    V *before* the PV contraction: sum_k P = 1, so the result is identical up to f32
//                    reassociation and the [Nq*16, 1024] intermediate is never formed)
//
// Kernels:
//   pos_logits_kernel : boxes -> w[h][q][k] (f32), one thread per (q,k) pair, 64-d sin/cos embedding and the
//                       16x64 Wg contraction in registers (Wg staged in LDS, broadcast reads).
//   attn_kernel<T>    : flash-style streaming softmax.  One wave owns 32 query rows of one head; per 32-key tile
//                       S^T = K Q^T on MFMA (so a lane holds 16 keys of ONE query: row max / sum are register
//                       reductions + one cross-half shuffle), P feeds the PV MFMA straight from registers,
//                       V is consumed from a key-contiguous (transposed) LDS image.  bf16 -> mfma_32x32x16_bf16,
//                       f32 -> mfma_32x32x2_f32 (exact f32, parity mode).  The key range can be split over
//                       blockIdx.z (a call has only Nq/128 * 16 blocks for 256 CUs); partial (m, l, O) are then
//                       merged by attn_combine_kernel.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

// ----------------------------------------------------------------------------------------------- position logits
// sin/cos of a = 100 * log-ratio / 1000^(i/8), |a| up to a few hundred rad: two-constant Cody-Waite reduction to
// [-pi, pi] (exact to ~1e-7 rad for |n| < 2^12), then the hardware sin/cos (argument in revolutions).
__device__ __forceinline__ void sincos_reduced(float a, float* sn, float* cs) {
  const float INV2PI = 0.15915494309189535f;
  const float TWO_PI_HI = 6.28318548202514648f;      // float(2*pi)
  const float TWO_PI_LO = -1.7484555e-7f;            // 2*pi - float(2*pi)
  const float n = rintf(a * INV2PI);
  float r = fmaf(-n, TWO_PI_HI, a);
  r = fmaf(-n, TWO_PI_LO, r);
  const float rev = r * INV2PI;
  *sn = __builtin_amdgcn_sinf(rev);
  *cs = __builtin_amdgcn_cosf(rev);
}

// rois_q [Nq][4], rois_k [Nk][4], wgt [64][16] (Wg transposed), bg [16], dim_mat [8]
// out [16][Nq][ldp] f32
template <bool PRECISE>
__global__ __launch_bounds__(256) void pos_logits_kernel(const float4* __restrict__ rois_q,
                                                         const float4* __restrict__ rois_k,
                                                         const float* __restrict__ wgt, const float* __restrict__ bg,
                                                         const float* __restrict__ dim_mat, float* __restrict__ out,
                                                         int Nq, int Nk, int ldp) {
  __shared__ __attribute__((aligned(16))) float sw[64 * 16];
  __shared__ float sdim[8];
  for (int e = threadIdx.x; e < 1024; e += 256) sw[e] = wgt[e];
  if (threadIdx.x < 8) sdim[threadIdx.x] = dim_mat[threadIdx.x];
  __syncthreads();
  const int q = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= Nk) return;
  const float4 bq = rois_q[q];
  const float4 bk = rois_k[k];
  // :147-176 (bbox = query rois, ref_bbox = key rois)
  const float wq = bq.z - bq.x + 1.f, hq = bq.w - bq.y + 1.f;
  const float cxq = 0.5f * (bq.x + bq.z), cyq = 0.5f * (bq.y + bq.w);
  const float wk = bk.z - bk.x + 1.f, hk = bk.w - bk.y + 1.f;
  const float cxk = 0.5f * (bk.x + bk.z), cyk = 0.5f * (bk.y + bk.w);
  float pm[4];
  pm[0] = logf(fabsf((cxq - cxk) / wq) + 1e-3f);
  pm[1] = logf(fabsf((cyq - cyk) / hq) + 1e-3f);
  pm[2] = logf(wq / wk);
  pm[3] = logf(hq / hk);
  float acc[16];
#pragma unroll
  for (int h = 0; h < 16; ++h) acc[h] = bg[h];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const float pv = pm[d] * 100.0f;
#pragma unroll 2
    for (int i = 0; i < 8; ++i) {
      const float a = pv / sdim[i];
      float sn, cs;
      if (PRECISE) sincosf(a, &sn, &cs);
      else sincos_reduced(a, &sn, &cs);
      const float4* ws = reinterpret_cast<const float4*>(sw + (d * 16 + i) * 16);      // sin block: index d*16 + i
      const float4* wc = reinterpret_cast<const float4*>(sw + (d * 16 + 8 + i) * 16);  // cos block: d*16 + 8 + i
#pragma unroll
      for (int h4 = 0; h4 < 4; ++h4) {
        const float4 a4 = ws[h4], c4 = wc[h4];
        acc[4 * h4 + 0] = fmaf(sn, a4.x, acc[4 * h4 + 0]);
        acc[4 * h4 + 1] = fmaf(sn, a4.y, acc[4 * h4 + 1]);
        acc[4 * h4 + 2] = fmaf(sn, a4.z, acc[4 * h4 + 2]);
        acc[4 * h4 + 3] = fmaf(sn, a4.w, acc[4 * h4 + 3]);
        acc[4 * h4 + 0] = fmaf(cs, c4.x, acc[4 * h4 + 0]);
        acc[4 * h4 + 1] = fmaf(cs, c4.y, acc[4 * h4 + 1]);
        acc[4 * h4 + 2] = fmaf(cs, c4.z, acc[4 * h4 + 2]);
        acc[4 * h4 + 3] = fmaf(cs, c4.w, acc[4 * h4 + 3]);
      }
    }
  }
#pragma unroll
  for (int h = 0; h < 16; ++h)
    out[((size_t)h * Nq + q) * ldp + k] = logf(fmaxf(acc[h], 0.f) + 1e-6f);
}

// Fast (bf16-mode) variant with the 64 -> 16 Wg contraction on the matrix cores.  A wave handles tiles of 16 (q, k)
// pairs (one q, 16 consecutive k): v_mfma_f32_16x16x32_bf16 with A = the pairs' embedding (row = lane & 15,
// k = 8 (lane >> 4) + e), B = Wg^T (col = head), two K-steps for the 64-d embedding.  The four lanes that share a pair
// each build a quarter of its embedding (one delta's 8 sines or 8 cosines per K-step), so no value is computed twice
// and the 1024 FMAs per pair of the VALU version disappear.  D: col = head, rows = 4 consecutive pairs -> 16-B stores.
// sin(a + 2 pi shift_rev) after a two-constant Cody-Waite reduction: cos(x) = sin(x + 1/4 revolution), so a lane that
// needs only the sine OR the cosine of its arguments pays one transcendental per value.
__device__ __forceinline__ float sin_shifted(float a, float shift_rev) {
  const float INV2PI = 0.15915494309189535f;
  const float TWO_PI_HI = 6.28318548202514648f;
  const float TWO_PI_LO = -1.7484555e-7f;
  const float n = rintf(a * INV2PI);
  float r = fmaf(-n, TWO_PI_HI, a);
  r = fmaf(-n, TWO_PI_LO, r);
  return __builtin_amdgcn_sinf(fmaf(r, INV2PI, shift_rev));
}

// The same value with the argument already in REVOLUTIONS: sin(2 pi (x_rev + shift_rev)).  The position arguments are
// 100 * log-ratio / 1000^(i/8) with |log-ratio| <= 6.91 (the 1e-3 floor and a 1000-pixel image), i.e. |x_rev| <= 110:
// an f32 at that magnitude resolves 7.6e-6 revolutions = 4.8e-5 rad, the same as the radian argument the Cody-Waite
// form starts from (ulp of 690 rad = 6.1e-5), so v_fract + v_sin loses nothing -- and costs 3 VALU per value (fma,
// fract, sin) instead of 7.  (gfx9+ v_sin_f32 wants its argument pre-reduced to [0, 1).)
__device__ __forceinline__ float sin_rev(float x_rev) {
  return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x_rev));
}

// fast-mode (bf16 path) logarithm / division: v_log_f32 (1 ulp in log2) and v_rcp_f32 -- the position arguments are
// 100 x log(ratio): an absolute error of ~1e-6 in the log moves the phase by 1e-4 rad, far below the bf16 rounding
// of the embedding that follows.
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float fast_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

typedef float f32x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void pos_logits_mfma_kernel(const float4* __restrict__ rois_q,
                                                              const float4* __restrict__ rois_k,
                                                              const float* __restrict__ wgt,
                                                              const float* __restrict__ bg,
                                                              const float* __restrict__ dim_mat,
                                                              float* __restrict__ out, bf16_t* __restrict__ out_t,
                                                              int Nq, int Nk, int ldp) {
  // out_t != null: logits as bf16 in the attention kernel's own order, [16][ceil(Nk/32)][Nq][32] where the 32 keys of
  // a tile are stored (h2, rq, e) with key = 8 rq + 4 h2 + e -- the 16 keys one attention lane consumes per tile are
  // 32 contiguous bytes and a wave's read is one contiguous 2 KiB block.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = lane & 15, g = lane >> 4;          // pair within the tile / k-group (A, B);  head = row (B, D)
  const int q = blockIdx.y;
  // B fragments: Wg^T[k][head] for k = 32 s + 8 g + e
  bf16x8_t bw[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) bw[s][e] = (__bf16)wgt[(32 * s + 8 * g + e) * 16 + row];
  float crev[8];                                       // 100 / (2 pi dim[i]): log-ratio -> revolutions
#pragma unroll
  for (int i = 0; i < 8; ++i) crev[i] = (100.0f * 0.15915494309189535f) / dim_mat[i];
  const float bias = bg[row];
  const float4 bq = rois_q[q];
  const float wq = bq.z - bq.x + 1.f, hq = bq.w - bq.y + 1.f;
  const float cxq = 0.5f * (bq.x + bq.z), cyq = 0.5f * (bq.y + bq.w);
  const float shift = (g & 1) ? 0.25f : 0.f;          // odd k-groups hold cosines
#pragma unroll 1
  for (int t = 0; t < 4; ++t) {
    const int k0 = blockIdx.x * 256 + wave * 64 + t * 16;
    if (k0 >= Nk) break;
    const int k = min(k0 + row, Nk - 1);
    const float4 bk = rois_k[k];
    const float wk = bk.z - bk.x + 1.f, hk = bk.w - bk.y + 1.f;
    const float cxk = 0.5f * (bk.x + bk.z), cyk = 0.5f * (bk.y + bk.w);
    // K-step s covers deltas 2 s and 2 s + 1; this lane owns delta d = 2 s + (g >> 1)
    float pm[2];
    if (g >> 1) {
      pm[0] = fast_log(fabsf(fast_div(cyq - cyk, hq)) + 1e-3f);   // d = 1
      pm[1] = fast_log(fast_div(hq, hk));                           // d = 3
    } else {
      pm[0] = fast_log(fabsf(fast_div(cxq - cxk, wq)) + 1e-3f);   // d = 0
      pm[1] = fast_log(fast_div(wq, wk));                           // d = 2
    }
    f32x4_t acc = {bias, bias, bias, bias};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8_t a;
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = (__bf16)sin_rev(fmaf(pm[s], crev[i], shift));
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bw[s], acc, 0, 0, 0);
    }
    // D: head = lane & 15, pairs k0 + 4 g + r
    float4 o;
    o.x = fast_log(fmaxf(acc[0], 0.f) + 1e-6f); o.y = fast_log(fmaxf(acc[1], 0.f) + 1e-6f);
    o.z = fast_log(fmaxf(acc[2], 0.f) + 1e-6f); o.w = fast_log(fmaxf(acc[3], 0.f) + 1e-6f);
    if (out_t) {
      const int kf = k0 + 4 * g, kt = kf >> 5, kk = kf & 31;
      bf16_t* dst = out_t + (((size_t)row * ((Nk + 31) >> 5) + kt) * Nq + q) * 32 + ((kk >> 2) & 1) * 16 + (kk >> 3) * 4;
      uint2 pk;
      pk.x = (unsigned)f32_to_bf16(o.x) | ((unsigned)f32_to_bf16(o.y) << 16);
      pk.y = (unsigned)f32_to_bf16(o.z) | ((unsigned)f32_to_bf16(o.w) << 16);
      *reinterpret_cast<uint2*>(dst) = pk;
    } else {
      float* dst = out + ((size_t)row * Nq + q) * ldp + k0 + 4 * g;
      if (k0 + 4 * g + 3 < ldp) *reinterpret_cast<float4*>(dst) = o;    // ldp % 4 == 0: pad columns may be written
    }
  }
}

// Tiled-bf16 variant with coalesced output.  The tile-ordered logits [16][ceil(Nk/32)][Nq][32] are contiguous over q
// for a fixed (head, key tile): a block therefore owns 8 consecutive queries x 64 keys (two key tiles), computes its
// 512 pairs as 32 MFMA tiles (8 per wave, same operand layout as pos_logits_mfma_kernel), stages the 16 KiB of bf16
// logits in LDS in their final order and writes 32 runs of 512 contiguous bytes (16 B per lane) instead of 8-byte
// pieces scattered over 16 head planes.  Each lane needs only the sine OR the cosine of its 16 arguments:
// cos(x) = sin(x + 1/4 revolution), one transcendental per value instead of two.
// HT = bf16_t / f16_t (round 6: the head on IEEE-half operands): the embedding, Wg and the stored logits are HT
template <typename HT> struct PosMma;
template <> struct PosMma<bf16_t> {
  typedef bf16x8_t vec;
  typedef __bf16 elem;
  __device__ static __forceinline__ f32x4_t run(const vec& a, const vec& b, const f32x4_t& c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct PosMma<f16_t> {
  typedef f16x8_t vec;
  typedef _Float16 elem;
  __device__ static __forceinline__ f32x4_t run(const vec& a, const vec& b, const f32x4_t& c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

template <typename HT>
__device__ __forceinline__ void pos_logits_tiled_body(const float4* __restrict__ rois_q,
                                                      const float4* __restrict__ rois_k,
                                                      const float* __restrict__ wgt, const float* __restrict__ bg,
                                                      const float* __restrict__ dim_mat, unsigned short* __restrict__ out_t,
                                                      int Nq, int Nk) {
  typedef typename PosMma<HT>::vec hvec;
  typedef typename PosMma<HT>::elem helem;
  __shared__ __attribute__((aligned(16))) unsigned short stage[16 * 2 * 8 * 32];   // [head][key tile][q][tile order]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.y * 8, k00 = blockIdx.x * 64;
  hvec bw[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) bw[s][e] = (helem)wgt[(32 * s + 8 * g + e) * 16 + row];
  float crev[8];                                       // 100 / (2 pi dim[i]): log-ratio -> revolutions
#pragma unroll
  for (int i = 0; i < 8; ++i) crev[i] = (100.0f * 0.15915494309189535f) / dim_mat[i];
  const float bias = bg[row];
  const float shift = (g & 1) ? 0.25f : 0.f;        // odd k-groups hold cosines
  // The wave's 8 MFMA tiles use 2 queries (ql = 2 wave + {0, 1}) and 4 key groups of 16: their boxes are requested up
  // front, all six loads in flight at once (round 2 loaded them inside the tile loop: eight dependent memory round
  // trips per block made the kernel latency-bound -- 1.8 TB/s of logits written -- whatever the VALU count).
  float4 bqs[2], bks[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) bqs[j] = rois_q[min(q0 + 2 * wave + j, Nq - 1)];
#pragma unroll
  for (int j = 0; j < 4; ++j) bks[j] = rois_k[min(k00 + j * 16 + row, Nk - 1)];
#pragma unroll
  for (int tile = 0; tile < 8; ++tile) {
    const int ql = 2 * wave + (tile >> 2);
    const int kb = (tile & 3) * 16;
    const float4 bq = bqs[tile >> 2];
    const float4 bk = bks[tile & 3];
    const float wq = bq.z - bq.x + 1.f, hq = bq.w - bq.y + 1.f;
    const float cxq = 0.5f * (bq.x + bq.z), cyq = 0.5f * (bq.y + bq.w);
    const float wk = bk.z - bk.x + 1.f, hk = bk.w - bk.y + 1.f;
    const float cxk = 0.5f * (bk.x + bk.z), cyk = 0.5f * (bk.y + bk.w);
    float pm[2];
    if (g >> 1) {
      pm[0] = fast_log(fabsf(fast_div(cyq - cyk, hq)) + 1e-3f);   // d = 1
      pm[1] = fast_log(fast_div(hq, hk));                           // d = 3
    } else {
      pm[0] = fast_log(fabsf(fast_div(cxq - cxk, wq)) + 1e-3f);   // d = 0
      pm[1] = fast_log(fast_div(wq, wk));                           // d = 2
    }
    f32x4_t acc = {bias, bias, bias, bias};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      hvec a;
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = (helem)sin_rev(fmaf(pm[s], crev[i], shift));
      acc = PosMma<HT>::run(a, bw[s], acc);
    }
    // D: head = row, keys kb + 4 g + r of the block's 64
    const int kf = kb + 4 * g, ktl = kf >> 5, kk = kf & 31;
    uint2 pk;
    pk.x = Half16<HT>::pack2(fast_log(fmaxf(acc[0], 0.f) + 1e-6f), fast_log(fmaxf(acc[1], 0.f) + 1e-6f));
    pk.y = Half16<HT>::pack2(fast_log(fmaxf(acc[2], 0.f) + 1e-6f), fast_log(fmaxf(acc[3], 0.f) + 1e-6f));
    *reinterpret_cast<uint2*>(&stage[((row * 2 + ktl) * 8 + ql) * 32 + ((kk >> 2) & 1) * 16 + (kk >> 3) * 4]) = pk;
  }
  __syncthreads();
  const int ktiles = (Nk + 31) >> 5, kt0 = k00 >> 5;
  for (int idx = threadIdx.x; idx < 16 * 2 * 8 * 4; idx += 256) {
    const int chunk = idx & 3, ql = (idx >> 2) & 7, ktl = (idx >> 5) & 1, head = idx >> 6;
    if (q0 + ql < Nq && kt0 + ktl < ktiles)
      *reinterpret_cast<uint4*>(out_t + (((size_t)head * ktiles + kt0 + ktl) * Nq + q0 + ql) * 32 + chunk * 8) =
          *reinterpret_cast<const uint4*>(&stage[((head * 2 + ktl) * 8 + ql) * 32 + chunk * 8]);
  }
}

template <typename HT>
__global__ __launch_bounds__(256) void pos_logits_tiled_kernel(const float4* __restrict__ rois_q,
                                                               const float4* __restrict__ rois_k,
                                                               const float* __restrict__ wgt,
                                                               const float* __restrict__ bg,
                                                               const float* __restrict__ dim_mat,
                                                               unsigned short* __restrict__ out_t, int Nq, int Nk) {
  pos_logits_tiled_body<HT>(rois_q, rois_k, wgt, bg, dim_mat, out_t, Nq, Nk);
}

// the same for several (query boxes, key boxes) problems in one launch: blockIdx.z = problem
constexpr int POS_MAXB = 20;          // problems per launch: the key frames of the bench's 20-key-frame step-batch in ONE launch
struct PosBatch {
  struct { const float4* rq; const float4* rk; unsigned short* out; int Nq, Nk; } p[POS_MAXB];
};
template <typename HT>
__global__ __launch_bounds__(256) void pos_logits_tiled_batched_kernel(PosBatch b, const float* __restrict__ wgt,
                                                                       const float* __restrict__ bg,
                                                                       const float* __restrict__ dim_mat) {
  const auto& q = b.p[blockIdx.z];
  if ((int)blockIdx.x * 64 >= q.Nk || (int)blockIdx.y * 8 >= q.Nq) return;
  pos_logits_tiled_body<HT>(q.rq, q.rk, wgt, bg, dim_mat, q.out, q.Nq, q.Nk);
}

// ----------------------------------------------------------------------------------------------- attention
template <typename T> struct AttnCfg;
template <> struct AttnCfg<bf16_t> {
  static constexpr int KROW = 144;   // Ks row stride (64 bf16 = 128 B + 16)
  static constexpr int VROW = 72;    // Vs row stride (32 bf16 = 64 B + 8)
  static constexpr int NQF = 4;      // 16-B vectors per lane for a 64-wide head slice
  static constexpr int NPV = 2;      // P A-vectors per 32-key tile
  static constexpr int NLD = 1;      // 16-B loads per thread per tile (K and V each), 256 threads
};
template <> struct AttnCfg<f16_t> : AttnCfg<bf16_t> {};      // IEEE half (round 6): the bf16 geometry
template <> struct AttnCfg<float> {
  static constexpr int KROW = 272;   // 256 B + 16
  static constexpr int VROW = 144;   // 128 B + 16
  static constexpr int NQF = 8;
  static constexpr int NPV = 4;
  static constexpr int NLD = 2;
};

template <typename T> struct Mma32;
template <> struct Mma32<bf16_t> {
  __device__ static __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc,
                                                  0, 0, 0);
  }
};
template <> struct Mma32<f16_t> {
  __device__ static __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
  }
};
// sum of a packed, ROUNDED pair of P values (one dot-product instruction against (1, 1)); see the softmax below
template <typename T> struct PairSum;
template <> struct PairSum<bf16_t> {
  __device__ static __forceinline__ float run(unsigned pk, float acc) {
    typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_v, pk), __builtin_bit_cast(bf16x2_v, 0x3f803f80u), acc, false);
  }
};
template <> struct PairSum<f16_t> {
  __device__ static __forceinline__ float run(unsigned pk, float acc) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, pk), __builtin_bit_cast(f16x2_t, 0x3c003c00u), acc, false);
  }
};
template <> struct PairSum<float> {
  __device__ static __forceinline__ float run(unsigned, float acc) { return acc; }
};
template <> struct Mma32<float> {
  __device__ static __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};

// (pack_bf16x2 is common.h's: ONE v_cvt_pk_bf16_f32 per pair -- the element-wise form this file used until round 4
//  compiled to two conversions + shift + or)

struct AttnParams {
  const void* Q;              // [Nq][ldq], head h in columns h*64 .. h*64+63  (u already folded in)
  const void* K;              // [N1][ldk]   keys 0 .. N1-1
  const void* Vt;             // [G*64][ldv]  row h*64+dv, column = key (V already projected by Wv), keys 0 .. N1-1
  const void* K2;             // [Nk-N1][ldk]  keys N1 .. Nk-1 (second segment; unused when N1 == Nk)
  const void* Vt2;            // [G*64][ldv2]  keys N1 .. Nk-1; its columns may start at any 2-byte (bf16) / 4-byte (f32) address
  const float* pos;           // [G][Nq][ldp] additive logits or null
  const unsigned short* pos_t;   // or: 16-bit (T) logits in tile order [G][ceil(Nk/32)][Nq][32] (see pos_logits_mfma_kernel)
  const void* resid;          // [Nq][ldr] residual (feats_cur) or null
  const float* bias_v;        // [G*64] or null
  void* out;                  // [Nq][ldo]
  float* part_o;              // [nsplit][Nq][G*64] un-normalised partial outputs (nsplit > 1)
  float* part_ml;             // [nsplit][2][G][Nq]   running max / sum of each partial
  int ldq, ldk, ldv, ldv2, ldp, ldr, ldo;
  int Nq, Nk, N1, G;
  float scale;
  int nsplit, tiles_per_split;
  int vmask_always;           // experiments: mask the V^T tail in every tile (the pre-round-3 form) instead of the last one
  int io_f32;                 // resid / out are f32 rows whatever T is (the head's f32 activation stream in bf16 mode)
};

template <typename T, bool POS_TILED, bool SEG>
__device__ __forceinline__ void attn_body(const AttnParams& p, const int split) {
  using C = AttnCfg<T>;
  constexpr int VE = 16 / (int)sizeof(T);
  constexpr int KV_PER_ROW = 64 * (int)sizeof(T) / 16;  // 16-B vectors per K row
  constexpr int VV_PER_ROW = 32 * (int)sizeof(T) / 16;  // 16-B vectors per V^T row (32 keys)
  constexpr int NLD = C::NLD;
  __shared__ __attribute__((aligned(16))) unsigned char Ks[2][32 * C::KROW];
  __shared__ __attribute__((aligned(16))) unsigned char Vs[2][64 * C::VROW];
  __shared__ __attribute__((aligned(16))) float sRow[4][32];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h2 = lane >> 5, l31 = lane & 31;
  const int head = blockIdx.y;
  const int qw0 = blockIdx.x * 128 + wave * 32;  // first query row of this wave
  const T* __restrict__ Qp = (const T*)p.Q;
  // Keys 0 .. N1-1 live in (K, Vt), keys N1 .. Nk-1 in (K2, Vt2): the [local window ; memory snapshot] key set of a MEGA
  // stage is read where its two parts already are (the projections' output, the memory tape) instead of being copied into
  // one buffer per key frame.  N1 == Nk: one segment.  Nothing about the arithmetic changes: same keys, same order.
  const int N1 = SEG ? p.N1 : p.Nk;      // (SEG = false: the one-segment build, no seam code in the tile loop)
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.K), 0, (int)((unsigned)N1 * (unsigned)p.ldk * (unsigned)sizeof(T)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.Vt), 0, (int)((unsigned)(p.G * 64) * (unsigned)p.ldv * (unsigned)sizeof(T)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_k2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.K2), 0, (int)((unsigned)(p.Nk - N1) * (unsigned)p.ldk * (unsigned)sizeof(T)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.Vt2), 0, (int)((unsigned)(p.G * 64) * (unsigned)p.ldv2 * (unsigned)sizeof(T)), 0x00020000);

  // ---- Q fragments (B operand of S^T = K Q^T): row q = qw0 + l31, vector v at bytes v*32 + h2*16
  const int q_ld = min(qw0 + l31, p.Nq - 1);
  uint4 qf[C::NQF];
  {
    const unsigned char* qrow = (const unsigned char*)(Qp + (size_t)q_ld * p.ldq + head * 64);
#pragma unroll
    for (int v = 0; v < C::NQF; ++v) qf[v] = *reinterpret_cast<const uint4*>(qrow + v * 32 + h2 * 16);
  }

  // ---- cooperative tile loads (global -> registers), two register sets = two tiles in flight
  auto load_tiles = [&](int k0, uint4 (&kreg)[NLD], uint4 (&vreg)[NLD]) {
    // which segment(s) the 32 keys of this tile come from (k0 is wave-uniform: scalar branches)
    const bool in1 = !SEG || k0 + 32 <= N1, in2 = SEG && k0 >= N1;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + 256 * i;
      // branch-free buffer loads: lanes past the end use offset 0xFFFFFFFF, which the range check turns into zeros
      {  // K tile: 32 keys x 64 d
        const int row = idx / KV_PER_ROW, vec = idx - row * KV_PER_ROW;
        const int key = k0 + row;
        const unsigned col = (unsigned)(head * 64 + vec * VE);
        const unsigned off1 = ((unsigned)key * (unsigned)p.ldk + col) * (unsigned)sizeof(T);
        const unsigned off2 = ((unsigned)(key - N1) * (unsigned)p.ldk + col) * (unsigned)sizeof(T);
        u32x4_t v;
        if (in1) {
          v = __builtin_amdgcn_raw_buffer_load_b128(rs_k, key < N1 ? off1 : 0xFFFFFFFFu, 0, 0);
        } else if (!SEG) {
          v = u32x4_t{0, 0, 0, 0};
        } else if (in2) {
          v = __builtin_amdgcn_raw_buffer_load_b128(rs_k2, key < p.Nk ? off2 : 0xFFFFFFFFu, 0, 0);
        } else {          // the tile with the seam: a row is in exactly one segment, the other load returns zeros
          const u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(rs_k, key < N1 ? off1 : 0xFFFFFFFFu, 0, 0);
          const u32x4_t b = __builtin_amdgcn_raw_buffer_load_b128(rs_k2, key >= N1 && key < p.Nk ? off2 : 0xFFFFFFFFu, 0, 0);
          v = a | b;
        }
        kreg[i] = make_uint4(v.x, v.y, v.z, v.w);
      }
      {  // V^T tile: 64 dv x 32 keys
        const int row = idx / VV_PER_ROW, vec = idx - row * VV_PER_ROW;
        const int key = k0 + vec * VE;
        const unsigned off1 = ((unsigned)(head * 64 + row) * (unsigned)p.ldv + (unsigned)key) * (unsigned)sizeof(T);
        const unsigned off2 = ((unsigned)(head * 64 + row) * (unsigned)p.ldv2 + (unsigned)(key - N1)) * (unsigned)sizeof(T);
        u32x4_t r;
        if (in1) {
          r = __builtin_amdgcn_raw_buffer_load_b128(rs_v, key < N1 ? off1 : 0xFFFFFFFFu, 0, 0);
        } else if (!SEG) {
          r = u32x4_t{0, 0, 0, 0};
        } else if (in2) {       // (16-byte loads at element alignment: gfx950 serves them, tools/probes/unaligned_load.hip)
          r = __builtin_amdgcn_raw_buffer_load_b128(rs_v2, key < p.Nk ? off2 : 0xFFFFFFFFu, 0, 0);
        } else {
          // the tile with the seam: whole vectors from either side as above; the one vector per row that straddles it is
          // assembled element by element
          const u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(rs_v, key + VE <= N1 ? off1 : 0xFFFFFFFFu, 0, 0);
          const u32x4_t b = __builtin_amdgcn_raw_buffer_load_b128(rs_v2, key >= N1 && key < p.Nk ? off2 : 0xFFFFFFFFu, 0, 0);
          r = a | b;
          if (key < N1 && key + VE > N1) {
            T e[VE];
#pragma unroll
            for (int t = 0; t < VE; ++t) {
              const int kk = key + t;
              const T* src = kk < N1 ? (const T*)p.Vt + (size_t)(head * 64 + row) * p.ldv + kk
                                     : (const T*)p.Vt2 + (size_t)(head * 64 + row) * p.ldv2 + (kk - N1);
              e[t] = kk < p.Nk ? *src : (T)0;
            }
            r = *reinterpret_cast<const u32x4_t*>(e);
          }
        }
        uint4 v = make_uint4(r.x, r.y, r.z, r.w);
        if (p.vmask_always || k0 + 32 > p.Nk) {   // only a tile that reaches past Nk has tail keys (wave-uniform branch):
          T* e = reinterpret_cast<T*>(&v);         // zero them (pad columns of Vt are not guaranteed finite)
#pragma unroll
          for (int t = 0; t < VE; ++t) e[t] = (key + t < p.Nk) ? e[t] : (T)0;
        }
        vreg[i] = v;
      }
    }
  };
  auto store_tiles = [&](int buf, const uint4 (&kreg)[NLD], const uint4 (&vreg)[NLD]) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + 256 * i;
      {
        const int row = idx / KV_PER_ROW, vec = idx - row * KV_PER_ROW;
        *reinterpret_cast<uint4*>(&Ks[buf][row * C::KROW + vec * 16]) = kreg[i];
      }
      {
        const int row = idx / VV_PER_ROW, vec = idx - row * VV_PER_ROW;
        unsigned char* dst = &Vs[buf][row * C::VROW + vec * 16];
        if (sizeof(T) == 2) {  // 72-B rows: 8-B aligned only
          *reinterpret_cast<uint2*>(dst) = make_uint2(vreg[i].x, vreg[i].y);
          *reinterpret_cast<uint2*>(dst + 8) = make_uint2(vreg[i].z, vreg[i].w);
        } else {
          *reinterpret_cast<uint4*>(dst) = vreg[i];
        }
      }
    }
  };

  f32x16_t o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;
  const float* pos_row = p.pos ? p.pos + ((size_t)head * p.Nq + q_ld) * p.ldp : nullptr;
  // tiled bf16 logits: this lane's 16 keys of tile kt are 32 contiguous bytes; fetched one tile pair ahead
  const int ktiles = (p.Nk + 31) >> 5;
  auto load_pos = [&](int kt, uint4 (&pr)[2]) {
    if (POS_TILED) {
      const int kc = min(kt, ktiles - 1);
      const uint4* src = reinterpret_cast<const uint4*>(p.pos_t + (((size_t)head * ktiles + kc) * p.Nq + q_ld) * 32 + h2 * 16);
      pr[0] = src[0];
      pr[1] = src[1];
    }
  };

  auto compute = [&](int cur, int k0, const uint4 (&pr)[2]) {
    // ---- S^T[key][q] = sum_d K[key][d] * Q[q][d]
    f32x16_t st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
    {
      const unsigned char* kb = &Ks[cur][l31 * C::KROW + h2 * 16];
#pragma unroll
      for (int v = 0; v < C::NQF; ++v) {
        const uint4 kf = *reinterpret_cast<const uint4*>(kb + v * 32);
        Mma32<T>::run(st, kf, qf[v]);
      }
    }
    // lane: query q = qw0 + l31, keys k0 + (r&3) + 8*(r>>2) + 4*h2
    float s[16];
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      float pe[4] = {0.f, 0.f, 0.f, 0.f};
      if (POS_TILED) {
        const unsigned short* pb = reinterpret_cast<const unsigned short*>(&pr[0]);   // element 4 rq + e
#pragma unroll
        for (int e = 0; e < 4; ++e) pe[e] = Half16<typename std::conditional<sizeof(T) == 2, T, bf16_t>::type>::one(pb[4 * rq + e]);
      } else if (pos_row) {
        const float4 pw = *reinterpret_cast<const float4*>(pos_row + k0 + 8 * rq + 4 * h2);
        pe[0] = pw.x; pe[1] = pw.y; pe[2] = pw.z; pe[3] = pw.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) s[4 * rq + e] = st[4 * rq + e] * p.scale + pe[e];
    }
    if (k0 + 32 > p.Nk) {        // only the last tile of the key range has keys to mask (wave-uniform branch)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + 8 * (r >> 2) + 4 * h2 + (r & 3);
        s[r] = key < p.Nk ? s[r] : -INFINITY;
      }
    }
    float mt = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    const float m_new = fmaxf(m_run, mt);
    // bf16 mode: hardware exp2 (v_exp_f32) instead of libm's expf (range checks + a divergent branch per value)
    auto ex = [](float v) { return sizeof(T) == 2 ? __expf(v) : expf(v); };
    const float alpha = ex(m_run - m_new);
    float psum = 0.f;
    unsigned pk[8];              // bf16 mode: P packed for the PV MFMA (two keys per dword)
    if (sizeof(T) == 2) {        // e^(s - m) = 2^(s log2e - m log2e): one FMA + v_exp_f32 per score (was sub, mul, exp)
      const float kL2E = 1.44269504088896340736f, mneg = -m_new * kL2E;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], kL2E, mneg));
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = ex(s[r] - m_new);
    }
    if (sizeof(T) == 2) {
      // P reaches the PV MFMA rounded to bf16; the row sum is taken over the SAME rounded values, so that
      // out = sum p'_j v_j / sum p'_j is an exact weighted mean of the v_j with slightly perturbed weights.  With the sum
      // over the unrounded p_j the rounding errors times the COMMON part of the values (bias + the mean of the post-ReLU
      // features) did not cancel: 2.6e-4 of the 6.5e-4 median logit error of the bf16 head (tools/head_precision_cpu.py).
      // (PairSum: v_dot2c_f32_bf16 / v_dot2_f32_f16 against (1.0, 1.0) sums a rounded pair in one instruction)
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        pk[d] = Half16<typename std::conditional<sizeof(T) == 2, T, bf16_t>::type>::pack2(s[2 * d], s[2 * d + 1]);
        psum = PairSum<T>::run(pk[d], psum);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) psum += s[r];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;

    // ---- rescale O: alpha is per query (lane&31) but O rows are (r&3)+8*(r>>2)+4*h2 -> exchange through LDS.
    //      Skipped (exactly: every alpha is 1) when no query of this wave saw a larger maximum in this tile, which
    //      is the common case once the first few tiles have been seen (wave-uniform branch).
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
      if (h2 == 0) sRow[wave][l31] = alpha;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const float4 a4 = *reinterpret_cast<const float4*>(&sRow[wave][8 * rq + 4 * h2]);
        o0[4 * rq + 0] *= a4.x; o0[4 * rq + 1] *= a4.y; o0[4 * rq + 2] *= a4.z; o0[4 * rq + 3] *= a4.w;
        o1[4 * rq + 0] *= a4.x; o1[4 * rq + 1] *= a4.y; o1[4 * rq + 2] *= a4.z; o1[4 * rq + 3] *= a4.w;
      }
      __builtin_amdgcn_wave_barrier();
    }

    // ---- O[q][dv] += sum_key P[q][key] * V[key][dv]
    const unsigned char* vb0 = &Vs[cur][l31 * C::VROW];
    const unsigned char* vb1 = &Vs[cur][(32 + l31) * C::VROW];
#pragma unroll
    for (int u = 0; u < C::NPV; ++u) {
      uint4 pa, v0, v1;
      if (sizeof(T) == 2) {
        pa.x = pk[4 * u + 0];
        pa.y = pk[4 * u + 1];
        pa.z = pk[4 * u + 2];
        pa.w = pk[4 * u + 3];
        const int off = (16 * u + 4 * h2) * 2;
        const uint2 a0 = *reinterpret_cast<const uint2*>(vb0 + off);
        const uint2 a1 = *reinterpret_cast<const uint2*>(vb0 + off + 16);
        const uint2 b0 = *reinterpret_cast<const uint2*>(vb1 + off);
        const uint2 b1 = *reinterpret_cast<const uint2*>(vb1 + off + 16);
        v0 = make_uint4(a0.x, a0.y, a1.x, a1.y);
        v1 = make_uint4(b0.x, b0.y, b1.x, b1.y);
      } else {
        pa.x = __float_as_uint(s[4 * u + 0]);
        pa.y = __float_as_uint(s[4 * u + 1]);
        pa.z = __float_as_uint(s[4 * u + 2]);
        pa.w = __float_as_uint(s[4 * u + 3]);
        const int off = (8 * u + 4 * h2) * 4;
        v0 = *reinterpret_cast<const uint4*>(vb0 + off);
        v1 = *reinterpret_cast<const uint4*>(vb1 + off);
      }
      Mma32<T>::run(o0, pa, v0);
      Mma32<T>::run(o1, pa, v1);
    }
  };

  // ---- this block's key-tile range; loads run two tiles ahead of the MFMAs
  const int ntiles = (p.Nk + 31) / 32;
  const int t0 = split * p.tiles_per_split;
  const int n = min(ntiles, t0 + p.tiles_per_split) - t0;
  uint4 ka[NLD], va[NLD], kb2[NLD], vb2[NLD];
  uint4 pA[2] = {}, pB[2] = {};          // position logits of the next even / odd tile (POS_TILED)
  // (unconditional steady-state body, see igemm.hip: tiles past this split's range are loaded but never used)
  load_tiles(t0 * 32, ka, va);
  load_pos(t0, pA);
  store_tiles(0, ka, va);
  load_tiles((t0 + 1) * 32, ka, va);
  load_pos(t0 + 1, pB);
  __syncthreads();
  for (int i = 0; i + 1 < n; i += 2) {   // invariant: LDS buffer 0 holds tile t0+i, ka/va hold tile t0+i+1
    load_tiles((t0 + i + 2) * 32, kb2, vb2);
    compute(0, (t0 + i) * 32, pA);
    load_pos(t0 + i + 2, pA);
    store_tiles(1, ka, va);
    __syncthreads();
    load_tiles((t0 + i + 3) * 32, ka, va);
    compute(1, (t0 + i + 1) * 32, pB);
    load_pos(t0 + i + 3, pB);
    store_tiles(0, kb2, vb2);
    __syncthreads();
  }
  if (n & 1) compute(0, (t0 + n - 1) * 32, pA);

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  if (p.nsplit > 1) {
    // ---- partial result: un-normalised O, running max and sum (merged by attn_combine_kernel)
    if (h2 == 0 && qw0 + l31 < p.Nq) {
      float* ml = p.part_ml + (size_t)split * 2 * p.G * p.Nq;
      ml[(size_t)head * p.Nq + qw0 + l31] = m_run;
      ml[(size_t)(p.G + head) * p.Nq + qw0 + l31] = l_tot;
    }
    float* po = p.part_o + (size_t)split * p.Nq * p.G * 64;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int col = head * 64 + jj * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = qw0 + (r & 3) + 8 * (r >> 2) + 4 * h2;
        if (q < p.Nq) po[(size_t)q * p.G * 64 + col] = (jj == 0 ? o0[r] : o1[r]);
      }
    }
    return;
  }
  // ---- normalise and write: out[q'][head*64 + dv] = resid + O/l + bias_v
  if (h2 == 0) sRow[wave][l31] = 1.f / l_tot;
  __builtin_amdgcn_wave_barrier();
  const T* __restrict__ resid = (const T*)p.resid;
  T* __restrict__ out = (T*)p.out;
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int col = head * 64 + jj * 32 + l31;
    const float bv = p.bias_v ? p.bias_v[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ql = (r & 3) + 8 * (r >> 2) + 4 * h2;
      const int q = qw0 + ql;
      if (q < p.Nq) {
        float v = (jj == 0 ? o0[r] : o1[r]) * sRow[wave][ql] + bv;
        if (p.io_f32) {          // (block-uniform) the residual stream stays f32: nothing is rounded between stages
          if (resid) v += reinterpret_cast<const float*>(p.resid)[(size_t)q * p.ldr + col];
          reinterpret_cast<float*>(p.out)[(size_t)q * p.ldo + col] = v;
        } else {
          if (resid) v += Elem<T>::ld(resid + (size_t)q * p.ldr + col);
          Elem<T>::st(out + (size_t)q * p.ldo + col, v);
        }
      }
    }
  }
}

template <typename T, bool POS_TILED>
__global__ __launch_bounds__(256) void attn_kernel(AttnParams p) {
  attn_body<T, POS_TILED, false>(p, (int)blockIdx.z);
}

// Several independent attention problems (the key frames of one engine step-batch at the same stage) in ONE launch:
// blockIdx.z enumerates (problem, key-range split) pairs through a small table, so the chip is filled by all
// problems together and the host pays one launch per stage instead of one per key frame.  Every problem runs exactly
// the code (and the split count) of its single-problem launch: same bits.
constexpr int ATTN_MAXB = 20;         // (kernel arguments are limited to 4 KiB: see the static_assert below)
constexpr int ATTN_MAXZ = 256;
struct AttnBatch {
  int n, nz;
  unsigned char zprob[ATTN_MAXZ], zsplit[ATTN_MAXZ];
  AttnParams p[ATTN_MAXB];
};
static_assert(sizeof(AttnBatch) <= 4096, "AttnBatch travels as the kernel argument");

// MINB = blocks per CU the register allocation aims at: 2 or 3 (<= 168 VGPRs: three waves per SIMD; default since the
// second key segment, see mega_relation_attention_batched; MEGA_ATTN_OCC3=0 selects the 2-block bound).
template <typename T, bool POS_TILED, int MINB, bool SEG>
__global__ __launch_bounds__(256, MINB) void attn_batched_kernel(AttnBatch b) {
  const int z = blockIdx.z;
  const AttnParams& p = b.p[b.zprob[z]];
  if ((int)blockIdx.x * 128 >= p.Nq) return;          // (block-uniform: the grid is sized for the largest problem)
  attn_body<T, POS_TILED, SEG>(p, (int)b.zsplit[z]);
}

// out[q][c] = resid + bias + (sum_s e^{m_s - M} O_s[q][c]) / (sum_s e^{m_s - M} l_s),  M = max_s m_s  (head = c / 64)
template <typename T>
__device__ __forceinline__ void attn_combine_body(const AttnParams& p) {
  const int D = p.G * 64;
  const size_t total = (size_t)p.Nq * D;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(idx / D), c = (int)(idx - (size_t)q * D), head = c >> 6;
    float M = -INFINITY;
    for (int s = 0; s < p.nsplit; ++s)
      M = fmaxf(M, p.part_ml[((size_t)s * 2 * p.G + head) * p.Nq + q]);
    float num = 0.f, den = 0.f;
    for (int s = 0; s < p.nsplit; ++s) {
      const float* ml = p.part_ml + (size_t)s * 2 * p.G * p.Nq;
      const float w = expf(ml[(size_t)head * p.Nq + q] - M);
      den += w * ml[(size_t)(p.G + head) * p.Nq + q];
      num += w * p.part_o[((size_t)s * p.Nq + q) * D + c];
    }
    float v = num / den + (p.bias_v ? p.bias_v[c] : 0.f);
    if (p.io_f32) {
      if (p.resid) v += reinterpret_cast<const float*>(p.resid)[(size_t)q * p.ldr + c];
      reinterpret_cast<float*>(p.out)[(size_t)q * p.ldo + c] = v;
    } else {
      if (p.resid) v += Elem<T>::ld((const T*)p.resid + (size_t)q * p.ldr + c);
      Elem<T>::st((T*)p.out + (size_t)q * p.ldo + c, v);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_combine_kernel(AttnParams p) {
  attn_combine_body<T>(p);
}

template <typename T>
__global__ __launch_bounds__(256) void attn_combine_batched_kernel(AttnBatch b) {
  const AttnParams& p = b.p[blockIdx.y];
  if (p.nsplit > 1) attn_combine_body<T>(p);
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

// w[h][q][k] = log(relu(Wg_h . pe(q,k) + bg_h) + 1e-6);  out [16][Nq][ldp] f32, ldp >= Nk
// precise != 0: libm-accurate sincosf (parity mode); 0: Cody-Waite reduction + hardware sin/cos (~1e-6 abs).
extern "C" int mega_position_logits(const float* rois_q, const float* rois_k, const float* wg_t, const float* bg,
                                    const float* dim_mat, float* out, int Nq, int Nk, int ldp, int precise,
                                    void* stream) {
  mega_clear_error();
  if (Nq == 0 || Nk == 0) return MEGA_OK;
  if (!rois_q || !rois_k || !wg_t || !bg || !dim_mat || !out || Nq < 0 || Nk < 0 || ldp < Nk) return MEGA_ERR_ARG;
  dim3 grid(cdiv(Nk, 256), Nq);
  if (precise)
    hipLaunchKernelGGL((pos_logits_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, (const float4*)rois_q,
                       (const float4*)rois_k, wg_t, bg, dim_mat, out, Nq, Nk, ldp);
  else if (ldp % 4 == 0 && (reinterpret_cast<size_t>(out) & 15) == 0)
    hipLaunchKernelGGL(pos_logits_mfma_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float4*)rois_q,
                       (const float4*)rois_k, wg_t, bg, dim_mat, out, (bf16_t*)nullptr, Nq, Nk, ldp);
  else
    hipLaunchKernelGGL((pos_logits_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, (const float4*)rois_q,
                       (const float4*)rois_k, wg_t, bg, dim_mat, out, Nq, Nk, ldp);
  return mega_check_launch();
}

static int position_logits_tiled_impl(const float* rois_q, const float* rois_k, const float* wg_t, const float* bg,
                                      const float* dim_mat, void* out_bf16, int Nq, int Nk, int dtype, void* stream) {
  mega_clear_error();
  if (dtype != MEGA_BF16 && dtype != MEGA_F16) return MEGA_ERR_ARG;
  if (Nq == 0 || Nk == 0) return MEGA_OK;
  if (!rois_q || !rois_k || !wg_t || !bg || !dim_mat || !out_bf16 || Nq < 0 || Nk < 0 ||
      (reinterpret_cast<size_t>(out_bf16) & 15))
    return MEGA_ERR_ARG;
  static const bool legacy = getenv("MEGA_POS_LEGACY") != nullptr;     // A/B switch (experiments)
  if (dtype == MEGA_F16) {
    dim3 grid(cdiv(Nk, 64), cdiv(Nq, 8));
    hipLaunchKernelGGL(pos_logits_tiled_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const float4*)rois_q,
                       (const float4*)rois_k, wg_t, bg, dim_mat, (unsigned short*)out_bf16, Nq, Nk);
  } else if (legacy) {
    dim3 grid(cdiv(Nk, 256), Nq);
    hipLaunchKernelGGL(pos_logits_mfma_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float4*)rois_q,
                       (const float4*)rois_k, wg_t, bg, dim_mat, (float*)nullptr, (bf16_t*)out_bf16, Nq, Nk, 0);
  } else {
    dim3 grid(cdiv(Nk, 64), cdiv(Nq, 8));
    hipLaunchKernelGGL(pos_logits_tiled_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const float4*)rois_q,
                       (const float4*)rois_k, wg_t, bg, dim_mat, (unsigned short*)out_bf16, Nq, Nk);
  }
  return mega_check_launch();
}

extern "C" int mega_position_logits_tiled(const float* rois_q, const float* rois_k, const float* wg_t, const float* bg,
                                          const float* dim_mat, void* out_bf16, int Nq, int Nk, void* stream) {
  return position_logits_tiled_impl(rois_q, rois_k, wg_t, bg, dim_mat, out_bf16, Nq, Nk, MEGA_BF16, stream);
}

// round 6: the tile-ordered logits in the head's 16-bit operand type (dtype = MEGA_BF16 / MEGA_F16)
extern "C" int mega_position_logits_tiled_dt(const float* rois_q, const float* rois_k, const float* wg_t, const float* bg,
                                             const float* dim_mat, void* out16, int Nq, int Nk, int dtype, void* stream) {
  return position_logits_tiled_impl(rois_q, rois_k, wg_t, bg, dim_mat, out16, Nq, Nk, dtype, stream);
}

// Number of key-range splits the attention core uses for (Nq, Nk) -- a function of the problem alone, so a problem
// gets the same bits in a single launch and inside a batched launch.  The engine batches 10-20 problems per launch
// (>= 480 blocks without any split), so a problem only asks for MEGA_ATTN_BLOCKS = 48 blocks of its own: no MEGA shape
// (Nq >= 300: 3 x 16 blocks) splits any more, and the partial-sum round trip + combine launch (0.21 ms per 10 key
// frames, +1.3 % FPS measured with the splits off) disappears.  Round 1 used 768 for single launches, round 2 256.
extern "C" int mega_relation_attention_splits(int Nq, int Nk, int groups) {
  if (Nq <= 0 || Nk <= 0 || groups <= 0) return 1;
  static const int target = getenv("MEGA_ATTN_BLOCKS") ? atoi(getenv("MEGA_ATTN_BLOCKS")) : 48;
  const int blocks = cdiv(Nq, 128) * groups;
  const int ntiles = cdiv(Nk, 32);
  int s = cdiv(target, blocks);
  if (s > ntiles / 4) s = ntiles / 4;
  if (s > 16) s = 16;
  return s < 1 ? 1 : s;
}

extern "C" size_t mega_relation_attention_workspace_bytes(int Nq, int Nk, int groups) {
  const int s = mega_relation_attention_splits(Nq, Nk, groups);
  if (s <= 1) return 0;
  return align_up((size_t)s * Nq * groups * 64 * sizeof(float), 256) +
         align_up((size_t)s * 2 * groups * Nq * sizeof(float), 256);
}

// Multi-head relation attention core (groups heads x 64).  See header comment for the formula.
static int attn_fill(AttnParams& p, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldv,
                     const float* pos, int ldp, const void* pos_tiled, const void* resid, int ldr, const float* bias_v,
                     void* out, int ldo, int Nq, int Nk, int groups, float scale, int dtype, void* ws, size_t ws_bytes,
                     int io_f32 = 0, int nk1 = 0, const void* k2 = nullptr, const void* vt2 = nullptr, int ldv2 = 0) {
  if (!q || !k || !vt || !out || Nq <= 0 || Nk <= 0 || groups <= 0) return MEGA_ERR_ARG;
  // second key segment: keys nk1 .. Nk-1 from (k2, vt2); nk1 == 0 or nk1 == Nk: one segment
  if (nk1 < 0 || nk1 > Nk) return MEGA_ERR_ARG;
  if (nk1 == 0 || nk1 == Nk) { nk1 = Nk; k2 = k; vt2 = vt; ldv2 = ldv; }
  else if (!k2 || !vt2 || ldv2 < Nk - nk1) return MEGA_ERR_ARG;
  p.N1 = nk1; p.K2 = k2; p.Vt2 = vt2; p.ldv2 = ldv2;
  p.io_f32 = (io_f32 != 0 && dtype != MEGA_F32) ? 1 : 0;   // (f32 mode: T is float already)
  const int ve = dtype != MEGA_F32 ? 8 : 4;
  if (ldq % ve || ldk % ve || ldv % ve || (pos && (ldp % 32 || ldp < Nk))) return MEGA_ERR_ARG;
  if (ldv < (nk1 == Nk ? ((Nk + ve - 1) / ve) * ve : nk1)) return MEGA_ERR_ARG;
  if (pos_tiled && (pos || dtype == MEGA_F32 || (reinterpret_cast<size_t>(pos_tiled) & 15))) return MEGA_ERR_ARG;
  p.pos_t = (const unsigned short*)pos_tiled;
  p.Q = q; p.ldq = ldq; p.K = k; p.ldk = ldk; p.Vt = vt; p.ldv = ldv; p.pos = pos; p.ldp = ldp;
  p.resid = resid; p.ldr = ldr; p.bias_v = bias_v; p.out = out; p.ldo = ldo; p.Nq = Nq; p.Nk = Nk; p.G = groups;
  p.scale = scale;
  const int want = mega_relation_attention_splits(Nq, Nk, groups);
  int nsplit = want;
  if (nsplit > 1 && (!ws || ws_bytes < mega_relation_attention_workspace_bytes(Nq, Nk, groups))) nsplit = 1;
  const int ntiles = cdiv(Nk, 32);
  p.tiles_per_split = cdiv(ntiles, nsplit);
  static const int vmask_always = (getenv("MEGA_ATTN_VMASK_ALWAYS") && getenv("MEGA_ATTN_VMASK_ALWAYS")[0] == '1') ? 1 : 0;
  p.vmask_always = vmask_always;      // (experiments; read once: not on the per-launch path)
  nsplit = cdiv(ntiles, p.tiles_per_split);   // no empty splits
  p.nsplit = nsplit;
  p.part_o = (float*)ws;
  p.part_ml = nsplit > 1 ? (float*)((unsigned char*)ws + align_up((size_t)want * Nq * groups * 64 * sizeof(float), 256))
                         : nullptr;
  return MEGA_OK;
}

static int relation_attention_impl(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldv,
                                   const float* pos, int ldp, const void* pos_tiled, const void* resid, int ldr,
                                   const float* bias_v, void* out, int ldo, int Nq, int Nk, int groups, float scale,
                                   int dtype, void* ws, size_t ws_bytes, void* stream) {
  mega_clear_error();
  if (Nq == 0) return MEGA_OK;
  AttnParams p;
  const int rc = attn_fill(p, q, ldq, k, ldk, vt, ldv, pos, ldp, pos_tiled, resid, ldr, bias_v, out, ldo, Nq, Nk, groups,
                           scale, dtype, ws, ws_bytes);
  if (rc != MEGA_OK) return rc;
  const int nsplit = p.nsplit;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(cdiv(Nq, 128), groups, nsplit);
  if (dtype == MEGA_BF16 && pos_tiled) hipLaunchKernelGGL((attn_kernel<bf16_t, true>), grid, dim3(256), 0, st, p);
  else if (dtype == MEGA_BF16) hipLaunchKernelGGL((attn_kernel<bf16_t, false>), grid, dim3(256), 0, st, p);
  else if (dtype == MEGA_F16 && pos_tiled) hipLaunchKernelGGL((attn_kernel<f16_t, true>), grid, dim3(256), 0, st, p);
  else if (dtype == MEGA_F16) hipLaunchKernelGGL((attn_kernel<f16_t, false>), grid, dim3(256), 0, st, p);
  else if (dtype == MEGA_F32) hipLaunchKernelGGL((attn_kernel<float, false>), grid, dim3(256), 0, st, p);
  else return MEGA_ERR_ARG;
  if (nsplit > 1) {
    const size_t total = (size_t)Nq * groups * 64;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (dtype == MEGA_BF16) hipLaunchKernelGGL((attn_combine_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, p);
    else if (dtype == MEGA_F16) hipLaunchKernelGGL((attn_combine_kernel<f16_t>), dim3(blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attn_combine_kernel<float>), dim3(blocks), dim3(256), 0, st, p);
  }
  return mega_check_launch();
}

// n <= 16 independent problems in one launch (+ one combine launch when any of them splits its key range).  All
// problems share groups / scale / dtype and the kind of position term (none, f32 rows, or tile-ordered bf16).
struct MegaAttnDescC {
  const void* q; const void* k; const void* vt; const float* pos; const void* pos_tiled; const void* resid;
  const float* bias_v; void* out; void* ws; size_t ws_bytes;
  int ldq, ldk, ldv, ldp, ldr, ldo, Nq, Nk;
  int io_f32, nk1;
  const void* k2; const void* vt2; int ldv2, reserved;
};

extern "C" int mega_relation_attention_batched(const void* descs, int n, int groups, float scale, int dtype,
                                               void* stream) {
  mega_clear_error();
  if (n == 0) return MEGA_OK;
  if (!descs || n < 0 || n > ATTN_MAXB) return MEGA_ERR_ARG;
  const MegaAttnDescC* d = (const MegaAttnDescC*)descs;
  AttnBatch b;                 // by-value kernel argument (~2.5 KB)
  b.n = n;
  int nz = 0, max_q = 0, any_split = 0, any_seg = 0;
  size_t max_total = 0;
  const bool tiled = d[0].pos_tiled != nullptr;
  for (int i = 0; i < n; ++i) {
    if ((d[i].pos_tiled != nullptr) != tiled) return MEGA_ERR_ARG;
    const int rc = attn_fill(b.p[i], d[i].q, d[i].ldq, d[i].k, d[i].ldk, d[i].vt, d[i].ldv, d[i].pos, d[i].ldp,
                             d[i].pos_tiled, d[i].resid, d[i].ldr, d[i].bias_v, d[i].out, d[i].ldo, d[i].Nq, d[i].Nk,
                             groups, scale, dtype, d[i].ws, d[i].ws_bytes, d[i].io_f32, d[i].nk1, d[i].k2, d[i].vt2,
                             d[i].ldv2);
    if (rc != MEGA_OK) return rc;
    if (nz + b.p[i].nsplit > ATTN_MAXZ) return MEGA_ERR_ARG;
    for (int s2 = 0; s2 < b.p[i].nsplit; ++s2) { b.zprob[nz] = (unsigned char)i; b.zsplit[nz] = (unsigned char)s2; ++nz; }
    max_q = d[i].Nq > max_q ? d[i].Nq : max_q;
    any_split |= b.p[i].nsplit > 1;
    any_seg |= b.p[i].N1 != b.p[i].Nk;
    const size_t total = (size_t)d[i].Nq * groups * 64;
    max_total = total > max_total ? total : max_total;
  }
  b.nz = nz;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(cdiv(max_q, 128), groups, nz);
  // (three blocks per CU: with the second key segment the tiled-position variant needs 170 VGPRs under a 2-block bound -- two
  //  over the 168 that still fit three waves per SIMD -- and exactly 168, without spills, under a 3-block bound)
  // Builds: SEG = some problem of the launch has a second key segment (the seam code costs the one-segment launches 5 %:
  // they keep their own build); MINB = blocks per CU the register allocation aims at -- the two-segment tiled-position
  // variant needs 170 VGPRs under a 2-block bound, two over the 168 that still fit three waves per SIMD, and exactly 168,
  // without spills, under a 3-block bound (886 against 1020 us at stage 0); every other variant fits 166 under the 2-block
  // bound and is 4 % faster that way.  MEGA_ATTN_OCC3 = 0 / 1 forces the bound for all variants (A/B).
  static const int occ_env = getenv("MEGA_ATTN_OCC3") == nullptr ? -1 : (getenv("MEGA_ATTN_OCC3")[0] == '1' ? 1 : 0);
  const bool occ3 = occ_env >= 0 ? occ_env == 1 : (any_seg && tiled);
#define MEGA_ATTN_LAUNCH(T, TILED)                                                                                   \
  do {                                                                                                               \
    if (any_seg && occ3) hipLaunchKernelGGL((attn_batched_kernel<T, TILED, 3, true>), grid, dim3(256), 0, st, b);    \
    else if (any_seg) hipLaunchKernelGGL((attn_batched_kernel<T, TILED, 2, true>), grid, dim3(256), 0, st, b);       \
    else if (occ3) hipLaunchKernelGGL((attn_batched_kernel<T, TILED, 3, false>), grid, dim3(256), 0, st, b);         \
    else hipLaunchKernelGGL((attn_batched_kernel<T, TILED, 2, false>), grid, dim3(256), 0, st, b);                   \
  } while (0)
  if (dtype == MEGA_BF16 && tiled) MEGA_ATTN_LAUNCH(bf16_t, true);
  else if (dtype == MEGA_BF16) MEGA_ATTN_LAUNCH(bf16_t, false);
  else if (dtype == MEGA_F16 && tiled) MEGA_ATTN_LAUNCH(f16_t, true);
  else if (dtype == MEGA_F16) MEGA_ATTN_LAUNCH(f16_t, false);
  else if (dtype == MEGA_F32 && any_seg) hipLaunchKernelGGL((attn_batched_kernel<float, false, 2, true>), grid, dim3(256), 0, st, b);
  else if (dtype == MEGA_F32) hipLaunchKernelGGL((attn_batched_kernel<float, false, 2, false>), grid, dim3(256), 0, st, b);
#undef MEGA_ATTN_LAUNCH      // (f32: 192 / 211 VGPRs, two blocks per CU either way; a 3-block bound would spill)
  else return MEGA_ERR_ARG;
  if (any_split) {
    const int blocks = (int)((max_total + 255) / 256 > 4096 ? 4096 : (max_total + 255) / 256);
    if (dtype == MEGA_BF16) hipLaunchKernelGGL((attn_combine_batched_kernel<bf16_t>), dim3(blocks, n), dim3(256), 0, st, b);
    else if (dtype == MEGA_F16) hipLaunchKernelGGL((attn_combine_batched_kernel<f16_t>), dim3(blocks, n), dim3(256), 0, st, b);
    else hipLaunchKernelGGL((attn_combine_batched_kernel<float>), dim3(blocks, n), dim3(256), 0, st, b);
  }
  return mega_check_launch();
}

struct MegaPosDescC { const float* rois_q; const float* rois_k; void* out_bf16; int Nq, Nk; };

static int position_logits_tiled_batched_impl(const void* descs, int n, const float* wg_t, const float* bg,
                                              const float* dim_mat, int dtype, void* stream) {
  mega_clear_error();
  if (dtype != MEGA_BF16 && dtype != MEGA_F16) return MEGA_ERR_ARG;
  if (n == 0) return MEGA_OK;
  if (!descs || n < 0 || n > POS_MAXB || !wg_t || !bg || !dim_mat) return MEGA_ERR_ARG;
  const MegaPosDescC* d = (const MegaPosDescC*)descs;
  PosBatch b;
  int max_q = 0, max_k = 0;
  for (int i = 0; i < n; ++i) {
    if (!d[i].rois_q || !d[i].rois_k || !d[i].out_bf16 || d[i].Nq <= 0 || d[i].Nk <= 0 ||
        (reinterpret_cast<size_t>(d[i].out_bf16) & 15))
      return MEGA_ERR_ARG;
    b.p[i].rq = (const float4*)d[i].rois_q; b.p[i].rk = (const float4*)d[i].rois_k; b.p[i].out = (unsigned short*)d[i].out_bf16;
    b.p[i].Nq = d[i].Nq; b.p[i].Nk = d[i].Nk;
    max_q = d[i].Nq > max_q ? d[i].Nq : max_q;
    max_k = d[i].Nk > max_k ? d[i].Nk : max_k;
  }
  dim3 grid(cdiv(max_k, 64), cdiv(max_q, 8), n);
  if (dtype == MEGA_F16) hipLaunchKernelGGL(pos_logits_tiled_batched_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)stream, b, wg_t, bg, dim_mat);
  else hipLaunchKernelGGL(pos_logits_tiled_batched_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, b, wg_t, bg, dim_mat);
  return mega_check_launch();
}

extern "C" int mega_position_logits_tiled_batched(const void* descs, int n, const float* wg_t, const float* bg,
                                                  const float* dim_mat, void* stream) {
  return position_logits_tiled_batched_impl(descs, n, wg_t, bg, dim_mat, MEGA_BF16, stream);
}

extern "C" int mega_position_logits_tiled_batched_dt(const void* descs, int n, const float* wg_t, const float* bg,
                                                     const float* dim_mat, int dtype, void* stream) {
  return position_logits_tiled_batched_impl(descs, n, wg_t, bg, dim_mat, dtype, stream);
}

extern "C" int mega_relation_attention(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldv,
                                       const float* pos, int ldp, const void* resid, int ldr, const float* bias_v,
                                       void* out, int ldo, int Nq, int Nk, int groups, float scale, int dtype,
                                       void* ws, size_t ws_bytes, void* stream) {
  return relation_attention_impl(q, ldq, k, ldk, vt, ldv, pos, ldp, nullptr, resid, ldr, bias_v, out, ldo, Nq, Nk,
                                 groups, scale, dtype, ws, ws_bytes, stream);
}

extern "C" int mega_relation_attention_tiled_pos(const void* q, int ldq, const void* k, int ldk, const void* vt,
                                                 int ldv, const void* pos_tiled_bf16, const void* resid, int ldr,
                                                 const float* bias_v, void* out, int ldo, int Nq, int Nk, int groups,
                                                 float scale, void* ws, size_t ws_bytes, void* stream) {
  return relation_attention_impl(q, ldq, k, ldk, vt, ldv, nullptr, 0, pos_tiled_bf16, resid, ldr, bias_v, out, ldo, Nq,
                                 Nk, groups, scale, MEGA_BF16, ws, ws_bytes, stream);
}

extern "C" int mega_relation_attention_tiled_pos_dt(const void* q, int ldq, const void* k, int ldk, const void* vt,
                                                    int ldv, const void* pos_tiled16, const void* resid, int ldr,
                                                    const float* bias_v, void* out, int ldo, int Nq, int Nk, int groups,
                                                    float scale, int dtype, void* ws, size_t ws_bytes, void* stream) {
  if (dtype != MEGA_BF16 && dtype != MEGA_F16) return MEGA_ERR_ARG;
  return relation_attention_impl(q, ldq, k, ldk, vt, ldv, nullptr, 0, pos_tiled16, resid, ldr, bias_v, out, ldo, Nq,
                                 Nk, groups, scale, dtype, ws, ws_bytes, stream);
}
