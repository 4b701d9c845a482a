// Box-level (HBM/latency-bound, integer + f32) kernels of the Faster R-CNN heads, all on device,
// no D2H round trip:
//   * greedy NMS        : mega_core/csrc/cuda/nms.cu:13-131 (+1 area convention, IoU > thr) and
//                         csrc/cpu/nms_cpu.cpp:5-65 (IoU >= thr) -- `strict_gt` selects the comparison
//   * RPN proposal path : mega_core/modeling/rpn/inference.py:76-123 (sigmoid, top-k sorted, decode, clip,
//                         small-box filter, NMS, keep first post_nms_top_n), box_coder.py:52-95,
//                         rpn/anchor_generator.py:73-95, structures/bounding_box.py:214-219
//   * box-head post-processing : roi_heads/box_head/inference.py:45-149 (softmax, per-class decode, clip,
//                         score threshold, per-class NMS, detections-per-image k-th value cut)
// This translation unit is compiled with -ffp-contract=off: the reference evaluates these formulas as
// separate f32 mul/add ops (torch elementwise kernels), so no FMA contraction here either.
//
// Ordering rule (documented deviation-free refinement): wherever the reference sorts scores with an
// unspecified tie order (torch.topk / sort, nms.cu:74), these kernels sort by (score desc, index asc).
#include <stdlib.h>

#include "common.h"

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ unsigned f32_sortable(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float sortable_f32(unsigned s) {
  unsigned u = (s & 0x80000000u) ? (s & 0x7fffffffu) : ~s;
  return __uint_as_float(u);
}

// Block-wide bitonic sort, descending, of n (power of two) u64 keys in LDS.
__device__ void bitonic_sort_desc(u64* s, int n) {
  for (int k2 = 2; k2 <= n; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const u64 a = s[i], b = s[ixj];
          const bool desc = (i & k2) == 0;
          if (desc ? (a < b) : (a > b)) { s[i] = b; s[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
}

// nms.cu:13-21 devIoU, term by term.
__device__ __forceinline__ float dev_iou(const float4 a, const float4 b) {
  const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
  const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
  const float width = fmaxf(right - left + 1.f, 0.f), height = fmaxf(bottom - top + 1.f, 0.f);
  const float interS = width * height;
  const float Sa = (a.z - a.x + 1.f) * (a.w - a.y + 1.f);
  const float Sb = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);
  return interS / (Sa + Sb - interS);
}

// The decision `dev_iou(a, b) > thr` (strict_gt) or `>= thr` WITHOUT the division in all but borderline cases:
// iou <> thr  <=>  interS <> thr * union; when the two sides differ by more than 1e-5 relative -- 40x the worst
// rounding error of the quotient (<= 2.5 ulp for the device's f32 division) plus that of the product -- the answer
// cannot depend on how the quotient rounds, otherwise the quotient is evaluated exactly as dev_iou does.
__device__ __forceinline__ bool dev_suppresses_areas(const float4 a, const float Sa, const float4 b, const float Sb,
                                                     float thr, int strict_gt) {
  const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
  const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
  const float width = fmaxf(right - left + 1.f, 0.f), height = fmaxf(bottom - top + 1.f, 0.f);
  const float interS = width * height;
  const float uni = Sa + Sb - interS;
  if (thr > 0.f && uni > 0.f) {
    const float rhs = thr * uni;
    if (interS > rhs * 1.00001f) return true;
    if (interS < rhs * 0.99999f) return false;
  }
  const float iou = interS / uni;
  return strict_gt ? (iou > thr) : (iou >= thr);
}
__device__ __forceinline__ bool dev_suppresses(const float4 a, const float4 b, float thr, int strict_gt) {
  return dev_suppresses_areas(a, (a.z - a.x + 1.f) * (a.w - a.y + 1.f), b, (b.z - b.x + 1.f) * (b.w - b.y + 1.f), thr,
                              strict_gt);
}

// ---------------------------------------------------------------------------------------------- NMS mask
// boxes: [P][nmax][4] score-sorted; counts[P]; mask: [P][nmax][cbmax] u64 (only col-block >= row-block written)
__global__ __launch_bounds__(64) void nms_mask_kernel(const float4* __restrict__ boxes, const int* __restrict__ counts,
                                                      u64* __restrict__ mask, int nmax, int cbmax, float thr,
                                                      int strict_gt) {
  const int p = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;
  const int n = min(counts[p], nmax);
  if (rb * 64 >= n || cb * 64 >= n) return;
  __shared__ float4 cbox[64];
  const int lane = threadIdx.x;
  const float4* bx = boxes + (size_t)p * nmax;
  const int col_size = min(n - cb * 64, 64);
  if (lane < col_size) cbox[lane] = bx[cb * 64 + lane];
  __syncthreads();
  const int i = rb * 64 + lane;
  if (i < n) {
    const float4 cur = bx[i];
    u64 t = 0;
    const int start = (rb == cb) ? lane + 1 : 0;
    for (int c = start; c < col_size; ++c) {
      if (dev_suppresses(cur, cbox[c], thr, strict_gt)) t |= 1ULL << c;
    }
    mask[((size_t)p * nmax + i) * cbmax + cb] = t;
  }
}

// ---------------------------------------------------------------------------------------------- NMS scan
// One wavefront per problem: the 64-bit suppression words of nms.cu map 1:1 onto wave64 lanes/ballots.
// keep_pos[P][max_keep] : kept positions (in score-sorted order), ascending = score-descending
// flags[P][nmax]        : optional, flags[order[pos]] = 1 for kept boxes (order may be null -> identity)
constexpr int SCAN_MAXCB = 128;  // up to 8192 boxes per problem

__global__ __launch_bounds__(64) void nms_scan_kernel(const u64* __restrict__ mask, const int* __restrict__ counts,
                                                      const unsigned char* __restrict__ valid,
                                                      const int* __restrict__ order, int nmax, int cbmax,
                                                      int max_keep, int* __restrict__ keep_pos,
                                                      int* __restrict__ keep_cnt, unsigned char* __restrict__ flags) {
  const int p = blockIdx.x;
  const int lane = threadIdx.x;
  const int n = min(counts[p], nmax);
  const int ncb = (n + 63) >> 6;
  const u64* mk = mask + (size_t)p * nmax * cbmax;
  u64 remv0 = 0, remv1 = 0;  // lane l holds removal words l and l+64
  int nkeep = 0;
  for (int blk = 0; blk < ncb && nkeep < max_keep; ++blk) {
    const int i = blk * 64 + lane;
    const bool ok = i < n && (!valid || valid[(size_t)p * nmax + i]);
    const u64 okmask = __ballot(ok);
    u64 diag = 0;
    if (i < n) diag = mk[(size_t)i * cbmax + blk];
    const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
    const u64 mine = blk < 64 ? remv0 : remv1;
    const int src = blk & 63;
    // NB: the builtin returns a signed int -- go through unsigned or bit 31 sign-extends into the high word
    u64 cur = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mine >> 32), src) << 32) |
              (u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)mine, src);
    cur |= ~okmask;
    u64 kept = 0;
    for (int t = 0; t < 64; ++t) {
      if (!((cur >> t) & 1ULL)) {
        kept |= 1ULL << t;
        cur |= ((u64)(unsigned)__builtin_amdgcn_readlane((int)dhi, t) << 32) |
               (u64)(unsigned)__builtin_amdgcn_readlane((int)dlo, t);
      }
    }
    // record kept boxes (ascending position)
    if ((kept >> lane) & 1ULL) {
      const int pos = nkeep + __popcll(kept & ((1ULL << lane) - 1ULL));
      if (pos < max_keep) {
        keep_pos[(size_t)p * max_keep + pos] = i;
        if (flags) flags[(size_t)p * nmax + (order ? order[(size_t)p * nmax + i] : i)] = 1;
      }
    }
    nkeep += __popcll(kept);
    if (nkeep >= max_keep) break;
    // OR the rows of the kept boxes into the removal words of later column blocks
    u64 k2 = kept;
    const bool w0 = lane > blk && lane < ncb;
    const bool w1 = (lane + 64) > blk && (lane + 64) < ncb;
    while (k2) {
      u64 acc0[4] = {0, 0, 0, 0}, acc1[4] = {0, 0, 0, 0};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (k2) {
          const int t = __ffsll((long long)k2) - 1;
          k2 &= k2 - 1;
          const u64* row = mk + (size_t)(blk * 64 + t) * cbmax;
          if (w0) acc0[u] = row[lane];
          if (w1) acc1[u] = row[lane + 64];
        }
      }
      remv0 |= acc0[0] | acc0[1] | acc0[2] | acc0[3];
      remv1 |= acc1[0] | acc1[1] | acc1[2] | acc1[3];
    }
  }
  if (lane == 0) keep_cnt[p] = nkeep < max_keep ? nkeep : max_keep;
}

// ---------------------------------------------------------------------------------------------- NMS, lazy form
// The mask form above evaluates all n^2/2 pairs although greedy NMS only ever needs the rows of the KEPT boxes, and
// the RPN stops after post_nms_top_n (300) of its 6000 candidates.  One 1024-thread block per problem keeps the
// score-sorted boxes in LDS (16 B x 8192 max) and walks them in windows of 1024 boxes (one box per thread, in
// registers), each window in 64-box blocks:
//   window entry   : every box of the window is tested against the boxes kept so far (boxes of windows the walk
//                    never reaches are never tested at all)
//   A  all threads : the 64 x 64 suppression bits inside the block (4 pairs per thread)
//   B  wave 0      : the serial scan of nms.cu:97-113 over those bits -> this block's kept boxes
//   C  all threads : the rest of the window against the block's kept boxes
// and stops as soon as max_keep boxes are kept.  Same IoU arithmetic, same comparison, same visiting order: the kept
// set is the mask form's, bit for bit (tests/test_kernels_gpu.py::test_nms_golden_and_random, test_rpn_select).
__global__ __launch_bounds__(1024) void nms_lazy_kernel(const float4* __restrict__ boxes, const int* __restrict__ counts,
                                                        const unsigned char* __restrict__ valid,
                                                        const int* __restrict__ order, int nmax, float thr,
                                                        int strict_gt, int max_keep, int* __restrict__ keep_pos,
                                                        int* __restrict__ keep_cnt, unsigned char* __restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lz_smem[];
  float4* sb = reinterpret_cast<float4*>(lz_smem);                       // [nmax] boxes
  unsigned char* rem = lz_smem + (size_t)nmax * 16;                       // [nmax] 1 = suppressed / invalid
  unsigned short* klist = reinterpret_cast<unsigned short*>(rem + (((size_t)nmax + 15) & ~(size_t)15));   // kept, in order
  __shared__ float4 kb[64];
  __shared__ unsigned diag_lo[2][64], diag_hi[2][64];
  __shared__ int s_nk, s_total;
  const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = min(counts[p], nmax);
  const float4* bx = boxes + (size_t)p * nmax;
  for (int j = tid; j < n; j += 1024) {
    sb[j] = bx[j];
    rem[j] = (valid && !valid[(size_t)p * nmax + j]) ? 1 : 0;
  }
  if (tid == 0) s_total = 0;
  __syncthreads();
  const int nblk = (n + 63) >> 6;
  // A(blk): bit c of row r = box b0+r suppresses box b0+c (c > r); double-buffered so that A(blk+1) shares an
  // epoch (one barrier interval) with C(blk)
  auto phase_a = [&](int blk) {
    const int b0 = blk * 64, bsz = min(64, n - b0);
    const int r = tid >> 4, c0 = (tid & 15) * 4;
    unsigned bits = 0;
    if (r < bsz) {
      const float4 me = sb[b0 + r];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = c0 + e;
        if (c > r && c < bsz) {
          if (dev_suppresses(me, sb[b0 + c], thr, strict_gt)) bits |= 1u << (c & 31);
        }
      }
    }
    // the 16 threads of a row are consecutive lanes: OR their 4-bit groups, one writer per row
    unsigned lo = c0 < 32 ? bits : 0u, hi = c0 < 32 ? 0u : bits;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      lo |= __shfl_xor(lo, d);
      hi |= __shfl_xor(hi, d);
    }
    if ((tid & 15) == 0) { diag_lo[blk & 1][r] = lo; diag_hi[blk & 1][r] = hi; }
  };
  if (nblk > 0) phase_a(0);
  __syncthreads();
  bool done = false;
  for (int ws = 0; ws < n && !done; ws += 1024) {
    // ---- window entry: this thread's box against everything kept so far
    const int j = ws + tid;
    float4 mine = make_float4(0.f, 0.f, 0.f, 0.f);
    bool alive = false;
    if (j < n) {
      mine = sb[j];
      alive = !rem[j];
    }
    const float my_area = (mine.z - mine.x + 1.f) * (mine.w - mine.y + 1.f);
    if (ws > 0) {
      const int K = s_total;
      for (int k = 0; k < K; ++k) {
        if (!__ballot(alive)) break;                      // the whole wavefront is dead
        const float4 kbx = sb[klist[k]];
        const float ka = (kbx.z - kbx.x + 1.f) * (kbx.w - kbx.y + 1.f);
        if (alive && dev_suppresses_areas(kbx, ka, mine, my_area, thr, strict_gt)) alive = false;
      }
      if (j < n && !alive) rem[j] = 1;
      __syncthreads();
    }
    const int blk_end = min(nblk, (ws >> 6) + 16);
    for (int blk = ws >> 6; blk < blk_end; ++blk) {
      const int b0 = blk * 64, bsz = min(64, n - b0);
      // ---- B: serial scan (one wavefront; the 64-bit words of nms.cu map onto wave64 ballots); one step per KEPT box
      if (wave == 0) {
        const int i = b0 + lane;
        const bool ok = lane < bsz && !rem[i];
        u64 cand = __ballot(ok);
        const unsigned dlo = diag_lo[blk & 1][lane], dhi = diag_hi[blk & 1][lane];
        u64 kept = 0;
        while (cand) {
          const int t = __ffsll((long long)cand) - 1;
          kept |= 1ULL << t;
          const u64 sup = ((u64)(unsigned)__builtin_amdgcn_readlane((int)dhi, t) << 32) |
                          (u64)(unsigned)__builtin_amdgcn_readlane((int)dlo, t);
          cand &= ~(sup | (1ULL << t));
        }
        const int base = s_total;
        if ((kept >> lane) & 1ULL) {
          const int k = __popcll(kept & ((1ULL << lane) - 1ULL));
          kb[k] = sb[i];
          klist[base + k] = (unsigned short)i;            // (global results are written once, after the walk: a store
        }                                                  //  here would put a memory round trip into every step)
        if (lane == 0) {
          s_nk = __popcll(kept);
          s_total = base + __popcll(kept);
        }
      }
      __syncthreads();
      const int nk = s_nk;
      if (s_total >= max_keep) { done = true; break; }
      // ---- C: the kept boxes of this block against the later boxes of the window;  A of the next block
      if (__ballot(alive && j >= b0 + 64)) {
        for (int k = 0; k < nk; ++k) {
          const float4 kbx = kb[k];
          const float ka = (kbx.z - kbx.x + 1.f) * (kbx.w - kbx.y + 1.f);
          if (alive && j >= b0 + 64 && dev_suppresses_areas(kbx, ka, mine, my_area, thr, strict_gt)) {
            alive = false;
            rem[j] = 1;
          }
        }
      }
      if (blk + 1 < nblk) phase_a(blk + 1);
      __syncthreads();
    }
  }
  __syncthreads();
  // ---- results: kept positions (ascending = score-descending), flags
  const int total = min(s_total, max_keep);
  for (int k = tid; k < total; k += 1024) {
    const int i = klist[k];
    keep_pos[(size_t)p * max_keep + k] = i;
    if (flags) flags[(size_t)p * nmax + (order ? order[(size_t)p * nmax + i] : i)] = 1;
  }
  if (tid == 0) keep_cnt[p] = total;
}

// ---------------------------------------------------------------------------------------------- generic sort
// Sort one problem's scores descending (ties: lower index first), gather boxes, emit order[].
// scores [P][nmax], boxes [P][nmax][4], counts[P] (null -> nmax).  n <= 8192.
constexpr int SORT_MAX = 8192;
__global__ __launch_bounds__(1024) void sort_boxes_kernel(const float* __restrict__ scores,
                                                          const float4* __restrict__ boxes,
                                                          const int* __restrict__ counts, int nmax,
                                                          float4* __restrict__ sboxes, float* __restrict__ sscores,
                                                          int* __restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u64* s = reinterpret_cast<u64*>(smem_raw);
  const int p = blockIdx.x;
  const int n = counts ? min(counts[p], nmax) : nmax;
  int ns = 64;
  while (ns < n) ns <<= 1;
  for (int i = threadIdx.x; i < ns; i += blockDim.x) {
    u64 key = 0;
    if (i < n) key = ((u64)f32_sortable(scores[(size_t)p * nmax + i]) << 32) | (u64)(0xFFFFFFFFu - (unsigned)i);
    s[i] = key;
  }
  __syncthreads();
  bitonic_sort_desc(s, ns);
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const u64 key = s[i];
    const int idx = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu));
    order[(size_t)p * nmax + i] = idx;
    sboxes[(size_t)p * nmax + i] = boxes[(size_t)p * nmax + idx];
    if (sscores) sscores[(size_t)p * nmax + i] = scores[(size_t)p * nmax + idx];
  }
}

// flags[P][nmax] -> ascending index list + count (single block per problem)
__global__ __launch_bounds__(1024) void compact_flags_kernel(const unsigned char* __restrict__ flags, int nmax,
                                                             long long* __restrict__ out_idx,
                                                             int* __restrict__ out_cnt) {
  __shared__ int wsum[16];
  __shared__ int base;
  const int p = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int start = 0; start < nmax; start += 1024) {
    const int i = start + tid;
    const bool f = i < nmax && flags[(size_t)p * nmax + i];
    const u64 bal = __ballot(f);
    if (lane == 0) wsum[wv] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wv; ++w) off += wsum[w];
    if (f) out_idx[(size_t)p * nmax + off + __popcll(bal & ((1ULL << lane) - 1ULL))] = i;
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += wsum[w];
      base += tot;
    }
    __syncthreads();
  }
  if (tid == 0) out_cnt[p] = base;
}

// ---------------------------------------------------------------------------------------------- RPN top-k + decode
struct RpnParams {
  const float* rpn_out;   // [B][Hf*Wf][ldc]: channel a = objectness logit of anchor a, A + a*4 + j = delta j
  const float* cell_anchors;  // [A][4]
  float4* boxes;          // [B][kmax][4]  decoded, clipped, score-sorted
  float* scores;          // [B][kmax]     sigmoid(logit)
  unsigned char* valid;   // [B][kmax]     remove_small_boxes result
  int* anchor_idx;        // [B][kmax] or null: flat anchor index (y * Wf + x) * A + a of each sorted candidate
  int* counts;            // [B]           = k
  int B, Hf, Wf, A, ldc, stride;
  int k;                  // pre_nms_top_n (already min'ed with the anchor count)
  int kmax;               // row capacity of boxes/scores
  float im_w, im_h, min_size, clip;
};

__global__ __launch_bounds__(1024) void rpn_topk_decode_kernel(RpnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u64* skeys = reinterpret_cast<u64*>(smem_raw);  // [ns]
  __shared__ unsigned hist[256];
  __shared__ unsigned sh_prefix, sh_k;
  __shared__ int sh_cnt, sh_eqbase;
  __shared__ int wsum[16];

  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int NA = p.Hf * p.Wf * p.A;
  const float* lg = p.rpn_out + (size_t)b * p.Hf * p.Wf * p.ldc;
  auto key_of = [&](int e) -> unsigned {
    const int pix = e / p.A, a = e - pix * p.A;
    return f32_sortable(lg[(size_t)pix * p.ldc + a]);
  };
  int ns = 64;
  while (ns < p.k) ns <<= 1;

  // ---- radix select: exact key T of the k-th largest logit and how many ties at T to take
  unsigned prefix = 0, kk = (unsigned)p.k;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned hmask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int e = tid; e < NA; e += 1024) {
      const unsigned key = key_of(e);
      if ((key & hmask) == (prefix & hmask)) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (wv == 0) {
      const unsigned c0 = hist[255 - 4 * lane], c1 = hist[254 - 4 * lane], c2 = hist[253 - 4 * lane],
                     c3 = hist[252 - 4 * lane];
      const unsigned s = c0 + c1 + c2 + c3;
      unsigned incl = s;
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
      }
      const unsigned excl = incl - s;
      if (excl < kk && kk <= incl) {
        unsigned r = kk - excl;
        int bin;
        if (r <= c0) { bin = 255 - 4 * lane; }
        else if (r <= c0 + c1) { bin = 254 - 4 * lane; r -= c0; }
        else if (r <= c0 + c1 + c2) { bin = 253 - 4 * lane; r -= c0 + c1; }
        else { bin = 252 - 4 * lane; r -= c0 + c1 + c2; }
        sh_prefix = prefix | ((unsigned)bin << shift);
        sh_k = r;
      }
    }
    __syncthreads();
    prefix = sh_prefix;
    kk = sh_k;
    __syncthreads();
  }
  const unsigned T = prefix;      // key of the k-th largest
  const int take_eq = (int)kk;    // ties at T to take, lowest anchor index first

  // ---- compaction into the sort buffer
  if (tid == 0) { sh_cnt = 0; sh_eqbase = 0; }
  for (int i = tid; i < ns; i += 1024) skeys[i] = 0;
  __syncthreads();
  for (int e = tid; e < NA; e += 1024) {
    const unsigned key = key_of(e);
    if (key > T) {
      const int pos = atomicAdd(&sh_cnt, 1);
      skeys[pos] = ((u64)key << 32) | (u64)(0xFFFFFFFFu - (unsigned)e);
    }
  }
  __syncthreads();
  const int n_gt = sh_cnt;  // == k - take_eq
  // ties: ordered by index -> chunked scan over the whole array
  for (int start = 0; start < NA; start += 1024) {
    const int e = start + tid;
    const bool f = e < NA && key_of(e) == T;
    const u64 bal = __ballot(f);
    if (lane == 0) wsum[wv] = __popcll(bal);
    __syncthreads();
    int off = sh_eqbase;
    for (int w = 0; w < wv; ++w) off += wsum[w];
    const int rank = off + __popcll(bal & ((1ULL << lane) - 1ULL));
    if (f && rank < take_eq) skeys[n_gt + rank] = ((u64)T << 32) | (u64)(0xFFFFFFFFu - (unsigned)e);
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += wsum[w];
      sh_eqbase += tot;
    }
    __syncthreads();
    if (sh_eqbase >= take_eq) break;
  }
  __syncthreads();
  bitonic_sort_desc(skeys, ns);

  // ---- decode + clip (box_coder.py:52-95 with weights (1,1,1,1); bounding_box.py:214-219)
  for (int i = tid; i < p.k; i += 1024) {
    const u64 key = skeys[i];
    const int e = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu));
    const float logit = sortable_f32((unsigned)(key >> 32));
    const int pix = e / p.A, a = e - pix * p.A;
    const int gy = pix / p.Wf, gx = pix - gy * p.Wf;
    const float sx = (float)(gx * p.stride), sy = (float)(gy * p.stride);
    const float ax1 = p.cell_anchors[a * 4 + 0] + sx, ay1 = p.cell_anchors[a * 4 + 1] + sy;
    const float ax2 = p.cell_anchors[a * 4 + 2] + sx, ay2 = p.cell_anchors[a * 4 + 3] + sy;
    const float* d = lg + (size_t)pix * p.ldc + p.A + a * 4;
    const float widths = ax2 - ax1 + 1.f, heights = ay2 - ay1 + 1.f;
    const float ctr_x = ax1 + 0.5f * widths, ctr_y = ay1 + 0.5f * heights;
    const float dx = d[0] / 1.f, dy = d[1] / 1.f;
    const float dw = fminf(d[2] / 1.f, p.clip), dh = fminf(d[3] / 1.f, p.clip);
    const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
    const float pw = expf(dw) * widths, ph = expf(dh) * heights;
    float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph;
    float x2 = pcx + 0.5f * pw - 1.f, y2 = pcy + 0.5f * ph - 1.f;
    x1 = fminf(fmaxf(x1, 0.f), p.im_w - 1.f);
    y1 = fminf(fmaxf(y1, 0.f), p.im_h - 1.f);
    x2 = fminf(fmaxf(x2, 0.f), p.im_w - 1.f);
    y2 = fminf(fmaxf(y2, 0.f), p.im_h - 1.f);
    const size_t o = (size_t)b * p.kmax + i;
    p.boxes[o] = make_float4(x1, y1, x2, y2);
    p.scores[o] = 1.f / (1.f + expf(-logit));
    // remove_small_boxes (boxlist_ops.py:34-50): xywh widths with TO_REMOVE = 1
    p.valid[o] = ((x2 - x1 + 1.f) >= p.min_size) && ((y2 - y1 + 1.f) >= p.min_size);
    if (p.anchor_idx) p.anchor_idx[o] = e;
  }
  if (tid == 0) p.counts[b] = p.k;
}

// proposals[b][r] = boxes[b][keep_pos[b][r]] for r < keep_cnt[b], zero rows after
__global__ void gather_kept_kernel(const float4* __restrict__ boxes, const float* __restrict__ scores,
                                   const int* __restrict__ keep_pos, const int* __restrict__ keep_cnt, int nmax,
                                   int max_keep, float4* __restrict__ out_boxes, float* __restrict__ out_scores,
                                   const int* __restrict__ idx_in, int* __restrict__ idx_out) {
  const int b = blockIdx.x;
  const int cnt = keep_cnt[b];
  for (int r = threadIdx.x; r < max_keep; r += blockDim.x) {
    float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
    float sc = 0.f;
    int ix = -1;
    if (r < cnt) {
      const int pos = keep_pos[(size_t)b * max_keep + r];
      bx = boxes[(size_t)b * nmax + pos];
      sc = scores[(size_t)b * nmax + pos];
      if (idx_in) ix = idx_in[(size_t)b * nmax + pos];
    }
    out_boxes[(size_t)b * max_keep + r] = bx;
    out_scores[(size_t)b * max_keep + r] = sc;
    if (idx_out) idx_out[(size_t)b * max_keep + r] = ix;
  }
}

// ---------------------------------------------------------------------------------------------- box-head post-processing
// P1: softmax + per-class decode (weights wx,wy,ww,wh) + clip + score threshold.
// Image b = blockIdx.y of a batch (every image R rows):
// logits [R][NC], deltas [R][NC*4], props [R][4]  ->  cboxes [NC-1][R][4], cscores [NC-1][R] (-1 when below thr)
__global__ void post_prepare_kernel(const float* __restrict__ logits, const float* __restrict__ deltas,
                                    const float4* __restrict__ props, const int* __restrict__ nprop_ptr, int R, int NC,
                                    float wx, float wy, float ww, float wh, float clip, float im_w, float im_h,
                                    float score_thresh, float4* __restrict__ cboxes, float* __restrict__ cscores,
                                    float* __restrict__ probs_out, unsigned char* __restrict__ flags) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int b = blockIdx.y;
  logits += (size_t)b * R * NC;
  deltas += (size_t)b * R * NC * 4;
  props += (size_t)b * R;
  cboxes += (size_t)b * (NC - 1) * R;
  cscores += (size_t)b * (NC - 1) * R;
  flags += (size_t)b * (NC - 1) * R;     // the kept flags of this call start from zero (written here, not by a memset node)
  if (probs_out) probs_out += (size_t)b * R * NC;
  const int nprop = nprop_ptr ? min(nprop_ptr[b], R) : R;
  const bool live = r < nprop;
  const float* lg = logits + (size_t)r * NC;
  float mx = -INFINITY;
  for (int j = 0; j < NC; ++j) mx = fmaxf(mx, lg[j]);
  float sum = 0.f;
  for (int j = 0; j < NC; ++j) sum += expf(lg[j] - mx);
  const float4 pb = props[r];
  const float widths = pb.z - pb.x + 1.f, heights = pb.w - pb.y + 1.f;
  const float ctr_x = pb.x + 0.5f * widths, ctr_y = pb.y + 0.5f * heights;
  for (int j = 0; j < NC; ++j) {
    const float pr = expf(lg[j] - mx) / sum;
    if (probs_out) probs_out[(size_t)r * NC + j] = pr;
    if (j == 0) continue;
    const float* d = deltas + (size_t)r * NC * 4 + j * 4;
    const float dx = d[0] / wx, dy = d[1] / wy;
    const float dw = fminf(d[2] / ww, clip), dh = fminf(d[3] / wh, clip);
    const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
    const float pw = expf(dw) * widths, ph = expf(dh) * heights;
    float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph;
    float x2 = pcx + 0.5f * pw - 1.f, y2 = pcy + 0.5f * ph - 1.f;
    x1 = fminf(fmaxf(x1, 0.f), im_w - 1.f);
    y1 = fminf(fmaxf(y1, 0.f), im_h - 1.f);
    x2 = fminf(fmaxf(x2, 0.f), im_w - 1.f);
    y2 = fminf(fmaxf(y2, 0.f), im_h - 1.f);
    const size_t o = (size_t)(j - 1) * R + r;
    cboxes[o] = make_float4(x1, y1, x2, y2);
    cscores[o] = (live && pr > score_thresh) ? pr : -1.f;
    flags[o] = 0;
  }
}

// P2: per-class sort of the candidates (score desc, proposal index asc); candidates with score < 0 dropped.
__global__ __launch_bounds__(256) void post_sort_kernel(const float4* __restrict__ cboxes,
                                                        const float* __restrict__ cscores, int R,
                                                        float4* __restrict__ sboxes, int* __restrict__ order,
                                                        int* __restrict__ counts) {
  __shared__ u64 s[1024];
  __shared__ int cnt;
  const int c = blockIdx.x;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  int ns = 64;
  while (ns < R) ns <<= 1;
  int local = 0;
  for (int i = threadIdx.x; i < ns; i += blockDim.x) {
    u64 key = 0;
    if (i < R) {
      const float sc = cscores[(size_t)c * R + i];
      if (sc >= 0.f) { key = ((u64)f32_sortable(sc) << 32) | (u64)(0xFFFFFFFFu - (unsigned)i); ++local; }
    }
    s[i] = key;
  }
  if (local) atomicAdd(&cnt, local);
  __syncthreads();
  bitonic_sort_desc(s, ns);
  const int n = cnt;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int idx = (int)(0xFFFFFFFFu - (unsigned)(s[i] & 0xFFFFFFFFu));
    order[(size_t)c * R + i] = idx;
    sboxes[(size_t)c * R + i] = cboxes[(size_t)c * R + idx];
  }
  if (threadIdx.x == 0) counts[c] = n;
}

// P4: class-major / proposal-ascending compaction of the kept detections, then the
// detections_per_img cut: keep score >= (D - max_det + 1)-th smallest score (inference.py:139-148).
// flags [NCm1][R]; outputs capacity NCm1*R.  One block per image (blockIdx.x) of a batch.
__global__ __launch_bounds__(1024) void post_finalize_kernel(const unsigned char* __restrict__ flags,
                                                             const float4* __restrict__ cboxes,
                                                             const float* __restrict__ cscores, int NCm1, int R,
                                                             int max_det, float4* __restrict__ out_boxes,
                                                             float* __restrict__ out_scores,
                                                             long long* __restrict__ out_labels,
                                                             int* __restrict__ out_cnt, int* __restrict__ tmp_idx) {
  __shared__ int wsum[16];
  __shared__ int base;
  __shared__ unsigned hist[256];
  __shared__ unsigned sh_prefix, sh_k;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int total = NCm1 * R;
  {
    const size_t img = (size_t)blockIdx.x * total;
    flags += img; cboxes += img; cscores += img; tmp_idx += img;
    out_boxes += img; out_scores += img; out_labels += img; out_cnt += blockIdx.x;
  }
  if (tid == 0) base = 0;
  __syncthreads();
  // pass 1: compaction of kept (class, proposal) pairs into tmp_idx (flat index c*R + r), order preserved
  for (int start = 0; start < total; start += 1024) {
    const int i = start + tid;
    const bool f = i < total && flags[i];
    const u64 bal = __ballot(f);
    if (lane == 0) wsum[wv] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wv; ++w) off += wsum[w];
    if (f) tmp_idx[off + __popcll(bal & ((1ULL << lane) - 1ULL))] = i;
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += wsum[w];
      base += tot;
    }
    __syncthreads();
  }
  const int D = base;
  __syncthreads();
  unsigned T = 0;  // sortable key threshold; keep key >= T
  if (max_det > 0 && D > max_det) {
    // k-th largest with k = max_det  (== (D - max_det + 1)-th smallest)
    unsigned prefix = 0, kk = (unsigned)max_det;
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      const unsigned hmask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
      for (int e = tid; e < D; e += 1024) {
        const unsigned key = f32_sortable(cscores[tmp_idx[e]]);
        if ((key & hmask) == (prefix & hmask)) atomicAdd(&hist[(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (wv == 0) {
        const unsigned c0 = hist[255 - 4 * lane], c1 = hist[254 - 4 * lane], c2 = hist[253 - 4 * lane],
                       c3 = hist[252 - 4 * lane];
        const unsigned s = c0 + c1 + c2 + c3;
        unsigned incl = s;
        for (int d = 1; d < 64; d <<= 1) {
          const unsigned t = __shfl_up(incl, d);
          if (lane >= d) incl += t;
        }
        const unsigned excl = incl - s;
        if (excl < kk && kk <= incl) {
          unsigned r = kk - excl;
          int bin;
          if (r <= c0) { bin = 255 - 4 * lane; }
          else if (r <= c0 + c1) { bin = 254 - 4 * lane; r -= c0; }
          else if (r <= c0 + c1 + c2) { bin = 253 - 4 * lane; r -= c0 + c1; }
          else { bin = 252 - 4 * lane; r -= c0 + c1 + c2; }
          sh_prefix = prefix | ((unsigned)bin << shift);
          sh_k = r;
        }
      }
      __syncthreads();
      prefix = sh_prefix;
      kk = sh_k;
      __syncthreads();
    }
    T = prefix;
  }
  // pass 2: final ordered compaction with the score cut
  if (tid == 0) base = 0;
  __syncthreads();
  for (int start = 0; start < D; start += 1024) {
    const int e = start + tid;
    int flat = 0;
    bool f = false;
    if (e < D) {
      flat = tmp_idx[e];
      f = f32_sortable(cscores[flat]) >= T;
    }
    const u64 bal = __ballot(f);
    if (lane == 0) wsum[wv] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wv; ++w) off += wsum[w];
    if (f) {
      const int o = off + __popcll(bal & ((1ULL << lane) - 1ULL));
      out_boxes[o] = cboxes[flat];
      out_scores[o] = cscores[flat];
      out_labels[o] = (long long)(flat / R + 1);
    }
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += wsum[w];
      base += tot;
    }
    __syncthreads();
  }
  if (tid == 0) *out_cnt = base;
}

// *p = v and zero[0 .. nzero) = 0  (grid covers nzero bytes, 4 per thread)
__global__ void set_int_kernel(int* p, int v, unsigned char* zero, int nzero) {
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i == 0) *p = v;
  for (int e = i; e < min(i + 4, nzero); ++e) zero[e] = 0;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

// ================================================================================================ C ABI
// Workspace size for mega_nms_sorted / rpn / post paths (bytes).
extern "C" size_t mega_nms_workspace_bytes(int P, int nmax) {
  const size_t cb = (size_t)cdiv(nmax, 64);
  return align_up((size_t)P * nmax * cb * sizeof(u64), 256) + 256;
}

// Greedy NMS over P independent, already score-sorted problems.
//   boxes [P][nmax][4] f32, counts[P] (device), valid [P][nmax] u8 or null
//   keep_pos [P][max_keep] i32 (positions in sorted order, ascending), keep_cnt [P]
//   order/flags optional: flags[p][order[p][pos]] = 1 for every kept box (flags must be pre-zeroed)
extern "C" int mega_nms_sorted(const float* boxes, const int* counts, const unsigned char* valid, const int* order,
                               int P, int nmax, float thr, int strict_gt, int max_keep, int* keep_pos, int* keep_cnt,
                               unsigned char* flags, void* ws, size_t ws_bytes, void* stream) {
  mega_clear_error();
  if (P == 0 || nmax == 0) return MEGA_OK;
  if (!boxes || !counts || !keep_pos || !keep_cnt || !ws || P < 0 || nmax < 0 || max_keep <= 0) return MEGA_ERR_ARG;
  if (nmax > SCAN_MAXCB * 64) return MEGA_ERR_ARG;
  if (ws_bytes < mega_nms_workspace_bytes(P, nmax)) return MEGA_ERR_WS;
  hipStream_t st = (hipStream_t)stream;
  const int cb = cdiv(nmax, 64);
  static const int lazy_min = getenv("MEGA_NMS_LAZY_MIN") ? atoi(getenv("MEGA_NMS_LAZY_MIN")) : 1024;
  // Lazy form: one block per problem walks the sorted list and evaluates only the kept boxes' rows, stopping at
  // max_keep -- it pays when few boxes are kept out of many (the RPN: 300 of 6000) or the list is long; many small
  // problems that keep everything (post-processing with R = 1024: 30 classes x B images, max_keep = R) are better
  // served by the mask + scan pair, whose work is spread over nmax^2 / 4096 blocks per problem.
  if (nmax >= lazy_min && (4 * max_keep <= nmax || nmax > 2048)) {
    const size_t lds = (((size_t)nmax * 17 + 15) & ~(size_t)15) + (size_t)nmax * 2 + 16;
    if (hipFuncSetAttribute((const void*)nms_lazy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return MEGA_ERR_LAUNCH;
    hipLaunchKernelGGL(nms_lazy_kernel, dim3(P), dim3(1024), lds, st, (const float4*)boxes, counts, valid, order, nmax,
                       thr, strict_gt, max_keep, keep_pos, keep_cnt, flags);
    return mega_check_launch();
  }
  u64* mask = (u64*)ws;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, cb, P), dim3(64), 0, st, (const float4*)boxes, counts, mask, nmax, cb,
                     thr, strict_gt);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(P), dim3(64), 0, st, mask, counts, valid, order, nmax, cb, max_keep,
                     keep_pos, keep_cnt, flags);
  return mega_check_launch();
}

// Drop-in core of mega_core._C.nms (csrc/nms.h:10-28): unsorted dets/scores -> kept ORIGINAL indices,
// ascending (nms.cu:127-130 / nms_cpu.cpp:64), int64, plus the count (device).
//   ws layout: sorted boxes | sorted order | keep_pos | flags | counts | mask
extern "C" size_t mega_nms_full_workspace_bytes(int n) {
  return align_up((size_t)n * 16, 256) + align_up((size_t)n * 4, 256) * 2 + align_up((size_t)n, 256) + 256 +
         mega_nms_workspace_bytes(1, n);
}

extern "C" int mega_nms(const float* dets, const float* scores, int n, float thr, int strict_gt, long long* keep_out,
                        int* keep_cnt, void* ws, size_t ws_bytes, void* stream) {
  mega_clear_error();
  if (n == 0) return MEGA_OK;
  if (!dets || !scores || !keep_out || !keep_cnt || !ws || n < 0) return MEGA_ERR_ARG;
  if (n > SORT_MAX) return MEGA_ERR_ARG;
  if (ws_bytes < mega_nms_full_workspace_bytes(n)) return MEGA_ERR_WS;
  hipStream_t st = (hipStream_t)stream;
  unsigned char* w = (unsigned char*)ws;
  float4* sboxes = (float4*)w; w += align_up((size_t)n * 16, 256);
  int* order = (int*)w; w += align_up((size_t)n * 4, 256);
  int* keep_pos = (int*)w; w += align_up((size_t)n * 4, 256);
  unsigned char* flags = w; w += align_up((size_t)n, 256);
  int* counts = (int*)w; w += 256;
  void* mws = w;
  int ns = 64;
  while (ns < n) ns <<= 1;
  // (a kernel, not hipMemsetAsync: captured into a hipGraph a memset NODE was seen to run out of order, see NOTES)
  hipLaunchKernelGGL(set_int_kernel, dim3(cdiv(n, 1024)), dim3(256), 0, st, counts, n, flags, n);
  hipLaunchKernelGGL(sort_boxes_kernel, dim3(1), dim3(1024), (size_t)ns * sizeof(u64), st, scores, (const float4*)dets,
                     (const int*)nullptr, n, sboxes, (float*)nullptr, order);
  int rc = mega_nms_sorted((const float*)sboxes, counts, nullptr, order, 1, n, thr, strict_gt, n, keep_pos, keep_cnt,
                           flags, mws, mega_nms_workspace_bytes(1, n), stream);
  if (rc != MEGA_OK) return rc;
  hipLaunchKernelGGL(compact_flags_kernel, dim3(1), dim3(1024), 0, st, flags, n, keep_out, keep_cnt);
  return mega_check_launch();
}

// RPN proposal selection for B frames (rpn/inference.py:76-123).
//   rpn_out [B][Hf*Wf][ldc] f32; cell_anchors [A][4]; outputs proposals [B][post_nms][4], prop_scores, prop_cnt[B]
extern "C" size_t mega_rpn_select_workspace_bytes(int B, int pre_nms) {
  return align_up((size_t)B * pre_nms * 16, 256) + align_up((size_t)B * pre_nms * 4, 256) * 3 +
         align_up((size_t)B * pre_nms, 256) + align_up((size_t)B * 4, 256) + mega_nms_workspace_bytes(B, pre_nms);
}

// prop_index [B][post_nms_top_n] (optional): the flat anchor index (y * Wf + x) * A + a of every kept proposal, -1 in
// the unused rows -- the "proposal indices after NMS" in the reference's (N, H, W, A) flattening (rpn/utils.py:10-14).
extern "C" int mega_rpn_select_idx(const float* rpn_out, const float* cell_anchors, int B, int Hf, int Wf, int A, int ldc,
                                   int anchor_stride, int pre_nms_top_n, int post_nms_top_n, float nms_thresh,
                                   int strict_gt, float min_size, float im_w, float im_h, float* proposals,
                                   float* prop_scores, int* prop_cnt, int* prop_index, void* ws, size_t ws_bytes,
                                   void* stream) {
  mega_clear_error();
  if (!rpn_out || !cell_anchors || !proposals || !prop_scores || !prop_cnt || !ws || B <= 0 || Hf <= 0 || Wf <= 0 ||
      A <= 0 || ldc < 5 * A || pre_nms_top_n <= 0 || post_nms_top_n <= 0)
    return MEGA_ERR_ARG;
  const int NA = Hf * Wf * A;
  const int k = pre_nms_top_n < NA ? pre_nms_top_n : NA;
  if (k > SORT_MAX) return MEGA_ERR_ARG;
  if (ws_bytes < mega_rpn_select_workspace_bytes(B, k)) return MEGA_ERR_WS;
  hipStream_t st = (hipStream_t)stream;
  unsigned char* w = (unsigned char*)ws;
  float4* sboxes = (float4*)w; w += align_up((size_t)B * k * 16, 256);
  float* sscores = (float*)w; w += align_up((size_t)B * k * 4, 256);
  int* keep_pos = (int*)w; w += align_up((size_t)B * k * 4, 256);
  int* aidx = (int*)w; w += align_up((size_t)B * k * 4, 256);
  unsigned char* valid = w; w += align_up((size_t)B * k, 256);
  int* counts = (int*)w; w += align_up((size_t)B * 4, 256);
  void* mws = w;
  RpnParams p;
  p.anchor_idx = prop_index ? aidx : nullptr;
  p.rpn_out = rpn_out; p.cell_anchors = cell_anchors; p.boxes = sboxes; p.scores = sscores; p.valid = valid;
  p.counts = counts; p.B = B; p.Hf = Hf; p.Wf = Wf; p.A = A; p.ldc = ldc; p.stride = anchor_stride; p.k = k;
  p.kmax = k; p.im_w = im_w; p.im_h = im_h; p.min_size = min_size; p.clip = logf(1000.f / 16.f);
  int ns = 64;
  while (ns < k) ns <<= 1;
  // set on every call: a per-process flag would miss the second device of a multi-GPU process
  (void)hipFuncSetAttribute((const void*)rpn_topk_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            SORT_MAX * (int)sizeof(u64));
  hipLaunchKernelGGL(rpn_topk_decode_kernel, dim3(B), dim3(1024), (size_t)ns * sizeof(u64), st, p);
  int rc = mega_nms_sorted((const float*)sboxes, counts, valid, nullptr, B, k, nms_thresh, strict_gt, post_nms_top_n,
                           keep_pos, prop_cnt, nullptr, mws, mega_nms_workspace_bytes(B, k), stream);
  if (rc != MEGA_OK) return rc;
  // keep_pos rows are post_nms_top_n wide
  hipLaunchKernelGGL(gather_kept_kernel, dim3(B), dim3(256), 0, st, sboxes, sscores, keep_pos, prop_cnt, k,
                     post_nms_top_n, (float4*)proposals, prop_scores, (const int*)p.anchor_idx, prop_index);
  return mega_check_launch();
}

extern "C" int mega_rpn_select(const float* rpn_out, const float* cell_anchors, int B, int Hf, int Wf, int A, int ldc,
                               int anchor_stride, int pre_nms_top_n, int post_nms_top_n, float nms_thresh,
                               int strict_gt, float min_size, float im_w, float im_h, float* proposals,
                               float* prop_scores, int* prop_cnt, void* ws, size_t ws_bytes, void* stream) {
  return mega_rpn_select_idx(rpn_out, cell_anchors, B, Hf, Wf, A, ldc, anchor_stride, pre_nms_top_n, post_nms_top_n,
                             nms_thresh, strict_gt, min_size, im_w, im_h, proposals, prop_scores, prop_cnt, nullptr, ws,
                             ws_bytes, stream);
}

// Box-head post-processor for one image (roi_heads/box_head/inference.py:45-149).
//   logits [R][NC], deltas [R][NC*4], props [R][4], nprop (device int, may be null -> R)
//   outputs (capacity (NC-1)*R rows): out_boxes [.][4], out_scores, out_labels (i64), out_cnt (device int)
extern "C" size_t mega_postprocess_batched_workspace_bytes(int B, int R, int NC) {
  const size_t m = (size_t)B * (NC - 1) * R;
  return 2 * align_up(m * 16, 256) + 4 * align_up(m * 4, 256) + align_up(m, 256) +
         2 * align_up((size_t)B * (NC - 1) * 4, 256) + mega_nms_workspace_bytes(B * (NC - 1), R);
}

extern "C" size_t mega_postprocess_workspace_bytes(int R, int NC) {
  return mega_postprocess_batched_workspace_bytes(1, R, NC);
}

extern "C" int mega_postprocess_batched(const float* logits, const float* deltas, const float* props, const int* nprop,
                                        int B, int R, int NC, float wx, float wy, float ww, float wh, float im_w,
                                        float im_h, float score_thresh, float nms_thresh, int strict_gt, int max_det,
                                        float* out_boxes, float* out_scores, long long* out_labels, int* out_cnt,
                                        float* probs_out, void* ws, size_t ws_bytes, void* stream) {
  mega_clear_error();
  if (!logits || !deltas || !props || !out_boxes || !out_scores || !out_labels || !out_cnt || !ws || R <= 0 || NC < 2 ||
      B <= 0 || B > 65535)
    return MEGA_ERR_ARG;
  if (R > 1024) return MEGA_ERR_ARG;
  if (ws_bytes < mega_postprocess_batched_workspace_bytes(B, R, NC)) return MEGA_ERR_WS;
  hipStream_t st = (hipStream_t)stream;
  const int C1 = NC - 1;
  const int P = B * C1;                      // (image, class) problems, image-major: the layout of every array below
  const size_t m = (size_t)P * R;
  unsigned char* w = (unsigned char*)ws;
  float4* cboxes = (float4*)w; w += align_up(m * 16, 256);
  float4* sboxes = (float4*)w; w += align_up(m * 16, 256);
  float* cscores = (float*)w; w += align_up(m * 4, 256);
  int* order = (int*)w; w += align_up(m * 4, 256);
  int* keep_pos = (int*)w; w += align_up(m * 4, 256);
  int* tmp_idx = (int*)w; w += align_up(m * 4, 256);
  unsigned char* flags = w; w += align_up(m, 256);
  int* counts = (int*)w; w += align_up((size_t)P * 4, 256);
  int* keep_cnt = (int*)w; w += align_up((size_t)P * 4, 256);
  void* mws = w;
  // (the kept flags are zeroed by post_prepare_kernel, one store per candidate it writes anyway: a hipMemsetAsync here
  // becomes a memset NODE when the call is captured into a hipGraph, and a captured FGFA step whose graph holds that node
  // produced history-dependent kept sets after a restart -- see NOTES, traps)
  hipLaunchKernelGGL(post_prepare_kernel, dim3(cdiv(R, 64), B), dim3(64), 0, st, logits, deltas, (const float4*)props,
                     nprop, R, NC, wx, wy, ww, wh, logf(1000.f / 16.f), im_w, im_h, score_thresh, cboxes, cscores,
                     probs_out, flags);
  hipLaunchKernelGGL(post_sort_kernel, dim3(P), dim3(256), 0, st, cboxes, cscores, R, sboxes, order, counts);
  int rc = mega_nms_sorted((const float*)sboxes, counts, nullptr, order, P, R, nms_thresh, strict_gt, R, keep_pos,
                           keep_cnt, flags, mws, mega_nms_workspace_bytes(P, R), stream);
  if (rc != MEGA_OK) return rc;
  hipLaunchKernelGGL(post_finalize_kernel, dim3(B), dim3(1024), 0, st, flags, cboxes, cscores, C1, R, max_det,
                     (float4*)out_boxes, out_scores, out_labels, out_cnt, tmp_idx);
  return mega_check_launch();
}

extern "C" int mega_postprocess(const float* logits, const float* deltas, const float* props, const int* nprop, int R,
                                int NC, float wx, float wy, float ww, float wh, float im_w, float im_h,
                                float score_thresh, float nms_thresh, int strict_gt, int max_det, float* out_boxes,
                                float* out_scores, long long* out_labels, int* out_cnt, float* probs_out, void* ws,
                                size_t ws_bytes, void* stream) {
  return mega_postprocess_batched(logits, deltas, props, nprop, 1, R, NC, wx, wy, ww, wh, im_w, im_h, score_thresh,
                                  nms_thresh, strict_gt, max_det, out_boxes, out_scores, out_labels, out_cnt, probs_out,
                                  ws, ws_bytes, stream);
}
