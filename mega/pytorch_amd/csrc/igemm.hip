// Implicit-GEMM convolution / linear layer on the gfx950 matrix cores.
//
// Replaces (on the MEGA inference path) every dense contraction the reference hands to
// cuDNN/cuBLAS through torch:  Conv2d + FrozenBatchNorm2d + ReLU (+ residual add) of the
// ResNet bottlenecks (mega_core/modeling/backbone/resnet.py:324-344, layers/batch_norm.py:19-31),
// the RPN head convs (modeling/rpn/rpn.py:99-106), the res5 head (resnet.py:155-204), and all
// nn.Linear layers of the box head (roi_box_feature_extractors.py:894,:907,:826; roi_box_predictors.py:50-57).
//
// Layout: activations NHWC  [N][H][W][Cin], weights OHWI [Cout][R][S][Cin] -> both GEMM operands
// are K-contiguous rows:   C[m][n] = sum_k A[m][k] * Bt[n][k],  m = (n_img, ho, wo), k = (r, s, c).
// A linear layer is the R=S=1, H=W=1 case.  Epilogue: y = acc*scale[n] + bias[n] (+ residual) (ReLU).
//
// Tiling (wave64): 256 threads = 4 waves in a 2x2 grid; block tile BM x BN, K-tile = 128 bytes of K
// per row (64 bf16 / 32 f32).  Global -> registers (16-B vectors, prefetched one K-tile ahead) ->
// LDS (row stride 144 B: every 16-row ds_read_b128 group is bank-conflict free) -> MFMA
// v_mfma_f32_32x32x16_bf16 (bf16) or v_mfma_f32_32x32x2_f32 (exact f32).  Two LDS buffers, one
// barrier per K-tile.  blockIdx -> tile map is XCD-aware (blocks of one XCD share A row panels in
// that XCD's L2).
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "igemm_params.h"

namespace {

constexpr int KTB = 128;          // bytes of K per LDS row per K-tile
constexpr int LDS_STRIDE = 144;   // bytes, 128 + 16 pad
constexpr int NTHREADS = 256;

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  // a, b: 16 bytes = 8 bf16 along K for row (lane&31), K-half (lane>>5)
  __device__ static __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
    bf16x8_t av = __builtin_bit_cast(bf16x8_t, a);
    bf16x8_t bv = __builtin_bit_cast(bf16x8_t, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
  }
};
template <> struct Mma<f16_t> {
  __device__ static __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // a, b: 4 f32 along K; MFMA e consumes component e (K order is permuted identically for A and B)
  __device__ static __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};

template <typename T, typename OT, int BM, int BN, bool PS = false>
__global__ __launch_bounds__(NTHREADS, (BM * BN <= 128 * 128 ? 2 : 1)) void igemm_kernel(ConvParams p) {
  constexpr int KE = KTB / (int)sizeof(T);   // K elements per tile
  constexpr int VE = 16 / (int)sizeof(T);    // elements per 16-B vector
  constexpr int AV = BM / 32;                // A vectors per thread per tile
  constexpr int BV = BN / 32;
  constexpr int WTM = BM / 2, WTN = BN / 2;  // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;                               // [2][BM][LDS_STRIDE]
  unsigned char* Bs = smem + 2 * BM * LDS_STRIDE;         // [2][BN][LDS_STRIDE]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- XCD-aware block -> tile map (bijective for any grid size)
  const int ntn = (p.Cout + BN - 1) / BN;
  int lid;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tile_m = lid / ntn, tile_n = lid - tile_m * ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // Buffer descriptors: loads are branch-free `buffer_load_dwordx4 ... offen`; a lane that must read zero
  // (conv padding, M / Cout tails) gets the offset 0xFFFFFFFF, which the hardware range check turns into 0.
  // No exec-mask branches around the loads => the compiler keeps COUNTED vmcnt waits and two K-tiles stay in flight.
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);

  // ---- per-thread load descriptors
  const int vec = tid & 7;
  const int lrow = tid >> 3;  // 0..31
  int a_hi0[AV], a_wi0[AV];
  unsigned a_base[AV];
  bool a_ok[AV];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < AV; ++i) {
    const int m = m0 + lrow + 32 * i;
    a_ok[i] = m < p.M;
    const int mm = a_ok[i] ? m : 0;
    const int nimg = mm / HoWo;
    const int rem = mm - nimg * HoWo;
    const int ho = rem / p.Wo;
    const int wo = rem - ho * p.Wo;
    a_hi0[i] = ho * p.stride - p.pad;
    a_wi0[i] = wo * p.stride - p.pad;
    a_base[i] = (unsigned)nimg * (unsigned)(p.H * p.W);
  }
  unsigned b_off[BV];
  bool b_ok[BV];
#pragma unroll
  for (int j = 0; j < BV; ++j) {
    const int n = n0 + lrow + 32 * j;
    b_ok[j] = n < p.Cout;
    b_off[j] = b_ok[j] ? ((unsigned)n * (unsigned)p.K + vec * VE) * (unsigned)sizeof(T) : 0xFFFFFFFFu;
  }

  // split-K: this block's K-tile range (the whole K when ksplit == 1)
  const int nkt_all = p.K / KE;
  const int kt_per = (nkt_all + p.ksplit - 1) / p.ksplit;
  const int kt0 = (int)blockIdx.z * kt_per;
  const int nkt = min(kt_per, nkt_all - kt0);
  int kk = kt0 * KE;           // k offset of the tile about to be loaded
  int kc = kk % p.Cin;         // and its (r, s, c0)
  int ks = (kk / p.Cin) % p.S, kr = (kk / p.Cin) / p.S;

  auto load_tile = [&](uint4 (&areg)[AV], uint4 (&breg)[BV]) {
    const int dh = kr * p.dil, dw = ks * p.dil;
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int hi = a_hi0[i] + dh, wi = a_wi0[i] + dw;
      const bool ok = a_ok[i] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      const unsigned off = ((a_base[i] + (unsigned)(hi * p.W + wi)) * (unsigned)p.Cin + (unsigned)(kc + vec * VE)) *
                           (unsigned)sizeof(T);
      const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? off : 0xFFFFFFFFu, 0, 0);
      areg[i] = make_uint4(v.x, v.y, v.z, v.w);
    }
#pragma unroll
    for (int j = 0; j < BV; ++j) {
      const unsigned off = b_ok[j] ? b_off[j] + (unsigned)kk * (unsigned)sizeof(T) : 0xFFFFFFFFu;
      const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0);
      breg[j] = make_uint4(v.x, v.y, v.z, v.w);
    }
    // advance (r, s, c0)
    kk += KE;
    kc += KE;
    if (kc >= p.Cin) {
      kc = 0;
      if (++ks == p.S) { ks = 0; ++kr; }
    }
  };
  auto store_tile = [&](int buf, const uint4 (&areg)[AV], const uint4 (&breg)[BV]) {
    unsigned char* a = As + buf * BM * LDS_STRIDE;
    unsigned char* b = Bs + buf * BN * LDS_STRIDE;
#pragma unroll
    for (int i = 0; i < AV; ++i)
      *reinterpret_cast<uint4*>(a + (lrow + 32 * i) * LDS_STRIDE + vec * 16) = areg[i];
#pragma unroll
    for (int j = 0; j < BV; ++j)
      *reinterpret_cast<uint4*>(b + (lrow + 32 * j) * LDS_STRIDE + vec * 16) = breg[j];
  };

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag_off = (lane & 31) * LDS_STRIDE + (lane >> 5) * 16;
  auto compute = [&](int cur) {
    const unsigned char* a = As + cur * BM * LDS_STRIDE + (wm * WTM) * LDS_STRIDE + frag_off;
    const unsigned char* b = Bs + cur * BN * LDS_STRIDE + (wn * WTN) * LDS_STRIDE + frag_off;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      uint4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const uint4*>(a + i * 32 * LDS_STRIDE + s4 * 32);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const uint4*>(b + j * 32 * LDS_STRIDE + s4 * 32);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<T>::run(acc[i][j], af[i], bf[j]);
    }
  };

  // ---- main loop: global loads run TWO K-tiles ahead of the MFMAs (two register sets, two LDS buffers,
  //      one barrier per K-tile); the store of tile t+1 waits only for the older register set.
  uint4 ra[AV], rb[BV], sa[AV], sb[BV];
  //      The steady-state body has NO conditionals (hipcc merges wait counts conservatively at control-flow
  //      joins: one conditional load collapses the counted vmcnt into vmcnt(0)).  Tiles past the end of K are
  //      still "loaded" -- the buffer range check makes that harmless -- and never fed to an MFMA.
  load_tile(ra, rb);                // tile 0
  store_tile(0, ra, rb);
  load_tile(ra, rb);                // tile 1 in flight
  __syncthreads();
  for (int it = 0; it < (nkt >> 1); ++it) {   // invariant: LDS buffer 0 holds tile 2*it, ra/rb hold tile 2*it+1
    load_tile(sa, sb);              // tile 2*it+2
    compute(0);                     // tile 2*it
    store_tile(1, ra, rb);          // tile 2*it+1
    __syncthreads();
    load_tile(ra, rb);              // tile 2*it+3
    compute(1);                     // tile 2*it+1
    store_tile(0, sa, sb);          // tile 2*it+2
    __syncthreads();
  }
  if (nkt & 1) {                    // odd tile count: the last tile sits in buffer 0
    compute(0);
    __syncthreads();                // the epilogue below re-uses the LDS buffers other waves may still read
  }

  // ---- epilogue.  Accumulators (lane owns column n = lane&31, rows (r&3)+8*(r>>2)+4*(lane>>5)) are scaled /
  //      biased and staged through LDS as f32, 64 tile rows at a time (the i-th 32-row slab of both wave rows:
  //      [2][32][BN+4] floats, so the staging area never exceeds the main loop's LDS even for 256-wide tiles),
  //      then written as whole 16-byte vectors along n: coalesced row segments instead of 2-byte-per-lane stores;
  //      the residual is read the same way.
  constexpr int CST = BN + 4;  // f32 row stride of the staged slab
  static_assert(64 * CST * 4 <= 2 * (BM + BN) * LDS_STRIDE, "staging slab must fit in the main-loop LDS");
  float* cs = reinterpret_cast<float*>(smem);
  OT* __restrict__ out = (OT*)p.out;
  const T* __restrict__ res = (const T*)p.res;
  constexpr int OVE = 16 / (int)sizeof(OT);   // output elements per 16-byte vector
  constexpr int VPR = BN / OVE;               // vectors per tile row
  // activation: relu = 0 none, 1 ReLU, 2 LeakyReLU(0.1) (FlowNetS, mega_core/modeling/backbone/flownet.py:50)
  const float neg_slope = p.relu == 1 ? 0.f : (p.relu == 2 ? 0.1f : 1.f);
  auto act = [&](float x) { return x > 0.f ? x : x * neg_slope; };
  const bool vec_ok = (p.ldo % OVE == 0) && (!res || (sizeof(T) == sizeof(OT) && p.ldr % OVE == 0));
  float sc[TN], bi[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WTN + j * 32 + (lane & 31);
    sc[j] = (p.ksplit == 1 && p.scale && n < p.Cout) ? p.scale[n] : 1.f;
    bi[j] = (p.ksplit == 1 && p.bias && n < p.Cout) ? p.bias[n] : 0.f;
  }
  // ---- fast path (Cout a multiple of the tile width, output below 2 GiB -- every layer of the frame stage): buffer
  //      loads / stores with the hardware range check instead of per-vector bounds branches.  With branches in the
  //      read-out loop hipcc merges the wait counts at every join into vmcnt(0), so each 16-byte store waited for
  //      the previous one to be acknowledged and each residual vector was an exposed round trip; straight-line code
  //      keeps the counts exact, the residual rows are a rolling (two-slab) prefetch and the stores drain behind the
  //      next slab's staging (barriers wait for LDS traffic only).
  constexpr bool RES_FAST = 64 * VPR / NTHREADS <= 4;  // (256-wide tiles: 8-16 residual vectors per slab in flight twice would spill)
  // PS (sub-pixel output, igemm_params.h): the launcher only admits shapes this path serves
  const size_t out_elems = PS ? (size_t)p.N * p.ps_H * p.ps_W * p.ldo : (size_t)(p.M - 1) * p.ldo + p.Cout;
  const bool fast = vec_ok && p.ksplit == 1 && n0 + BN <= p.Cout &&
                    out_elems * sizeof(OT) < 0x7FF00000ull &&
                    (res == nullptr || (RES_FAST && ((size_t)(p.M - 1) * p.ldr + p.Cout) * sizeof(T) < 0x7FF00000ull));
  if (fast) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        p.out, 0, (int)(out_elems * sizeof(OT)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.res ? p.res : p.out), 0, res ? (int)(((size_t)(p.M - 1) * p.ldr + p.Cout) * sizeof(T)) : 0,
        0x00020000);
    constexpr int NIT = 64 * VPR / NTHREADS;           // 16-byte vectors per thread per slab
    constexpr int RSTEP = NTHREADS / VPR;              // slab rows between a thread's consecutive vectors
    static_assert(NIT >= 1 && NIT * NTHREADS == 64 * VPR, "slab vectors must divide evenly");
    const int row0 = tid / VPR, cvo = (tid % VPR) * OVE, ncol = n0 + cvo;
    auto slab_m = [&](int i, int it) {      // (row0 < RSTEP, RSTEP divides 32: per-thread part + compile-time part)
      return (m0 + row0) + (((it * RSTEP) >> 5) * WTM + i * 32 + ((it * RSTEP) & 31));
    };
    // byte offset of the 16-byte output vector of GEMM row m (this thread's columns ncol ..): plain rows, or the sub-pixel
    // scatter of a transposed conv's phases (a vector never straddles two phases: ps_C % OVE == 0)
    const int ps_q = PS ? ncol / p.ps_C : 0;
    const int ps_col = PS ? p.ps_coff + (ncol - ps_q * p.ps_C) : 0;
    auto out_off = [&](int m) -> unsigned {
      if constexpr (PS) {
        const int t = m / HoWo, rem = m - t * HoWo, mh = rem / p.Wo, mw = rem - mh * p.Wo;
        const int y = 2 * mh + (ps_q >> 1) - p.ps_crop, x = 2 * mw + (ps_q & 1) - p.ps_crop;
        const bool ok = m < p.M && (unsigned)y < (unsigned)p.ps_H && (unsigned)x < (unsigned)p.ps_W;
        return ok ? (unsigned)(((t * p.ps_H + y) * p.ps_W + x) * p.ldo + ps_col) * (unsigned)sizeof(OT) : 0xFFFFFFFFu;
      } else {
        return (unsigned)(m * p.ldo + ncol) * (unsigned)sizeof(OT);
      }
    };
    // RL = 1: ReLU on the ROUNDED value (bf16: one v_pk_max_i16 per pair on the packed bits; f32: v_max) -- equal to
    // the generic x > 0 ? x : x * slope bit for bit except that negative inputs give +0 instead of -0 (see igemm8.hip)
    auto run = [&](auto HR, auto RL) {
      constexpr bool HAS_RES = decltype(HR)::value;
      constexpr bool RELU = decltype(RL)::value;
      constexpr int PD = 2;                          // slabs of residual rows in flight
      u32x4_t rr[PD][HAS_RES ? NIT : 1];
      auto ldres = [&](int i, u32x4_t (&dst)[HAS_RES ? NIT : 1]) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
          dst[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, (unsigned)(slab_m(i, it) * p.ldr + ncol) * (unsigned)sizeof(T), 0, 0);
      };
      if (HAS_RES) {
        ldres(0, rr[0]);
        if (PD > 1 && TM > 1) ldres(1, rr[PD - 1]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (i > 0) {                                   // the previous slab has been read out
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int nl = wn * WTN + j * 32 + (lane & 31);
          const int rb = wm * 32 + 4 * (lane >> 5);
#pragma unroll
          for (int r = 0; r < 16; ++r) cs[(rb + (r & 3) + 8 * (r >> 2)) * CST + nl] = acc[i][j][r] * sc[j] + bi[j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int row = row0 + it * RSTEP;
          float v[OVE];
#pragma unroll
          for (int t = 0; t < OVE; t += 4) {
            const float4 f = *reinterpret_cast<const float4*>(cs + row * CST + cvo + t);
            v[t] = f.x; v[t + 1] = f.y; v[t + 2] = f.z; v[t + 3] = f.w;
          }
          if (HAS_RES) {
            const T* re = reinterpret_cast<const T*>(&rr[i % PD][it]);
#pragma unroll
            for (int t = 0; t < OVE; ++t) v[t] += Elem<T>::ld(re + t);
          }
          u32x4_t o;                                 // packed explicitly (no type-punned stores into o)
          if constexpr (sizeof(OT) == 2) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              if constexpr (RELU) {
                const s16x2_t z = {0, 0};
                o[d] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, Half16<OT>::pack2(v[2 * d], v[2 * d + 1])), z));
              } else {
                o[d] = Half16<OT>::pack2(act(v[2 * d]), act(v[2 * d + 1]));
              }
            }
          } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = __float_as_uint(RELU ? fmaxf(v[t], 0.f) : act(v[t]));
          }
          __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, out_off(slab_m(i, it)), 0, 0);
        }
        if (HAS_RES && i + PD < TM) ldres(i + PD, rr[i % PD]);   // slab i + PD into the registers slab i just freed
      }
    };
    if (res) {
      if (p.relu == 1) run(std::integral_constant<bool, RES_FAST>{}, std::true_type{});
      else run(std::integral_constant<bool, RES_FAST>{}, std::false_type{});
    } else {
      if (p.relu == 1) run(std::false_type{}, std::true_type{}); else run(std::false_type{}, std::false_type{});
    }
    return;
  }
  if (PS && p.ksplit == 1) return;            // (the sub-pixel launcher refuses what the fast path does not serve)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (i > 0) __syncthreads();               // the previous slab has been read out
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nl = wn * WTN + j * 32 + (lane & 31);
      const int rb = wm * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) cs[(rb + (r & 3) + 8 * (r >> 2)) * CST + nl] = acc[i][j][r] * sc[j] + bi[j];
    }
    __syncthreads();
    if (p.ksplit > 1) {     // raw partial sums (sc = 1, bi = 0 were forced above); splitk_finalize_kernel finishes
      float* part = p.partial + (size_t)blockIdx.z * p.M * p.Cout;
      for (int e = tid; e < 64 * (BN / 4); e += NTHREADS) {
        const int row = e / (BN / 4), cv = e - row * (BN / 4);
        const int m = m0 + (row >> 5) * WTM + i * 32 + (row & 31), n = n0 + cv * 4;
        if (m >= p.M || n >= p.Cout) continue;
        const float4 f = *reinterpret_cast<const float4*>(cs + row * CST + cv * 4);
        if (n + 4 <= p.Cout && p.Cout % 4 == 0) {
          *reinterpret_cast<float4*>(part + (size_t)m * p.Cout + n) = f;
        } else {
          const float v[4] = {f.x, f.y, f.z, f.w};
          for (int t = 0; t < 4 && n + t < p.Cout; ++t) part[(size_t)m * p.Cout + n + t] = v[t];
        }
      }
      continue;
    }
    for (int e = tid; e < 64 * VPR; e += NTHREADS) {
      const int row = e / VPR, cv = e - row * VPR;
      const int m = m0 + (row >> 5) * WTM + i * 32 + (row & 31), n = n0 + cv * OVE;
      if (m >= p.M || n >= p.Cout) continue;
      float v[OVE];
#pragma unroll
      for (int t = 0; t < OVE; t += 4) {
        const float4 f = *reinterpret_cast<const float4*>(cs + row * CST + cv * OVE + t);
        v[t] = f.x; v[t + 1] = f.y; v[t + 2] = f.z; v[t + 3] = f.w;
      }
      if (vec_ok && n + OVE <= p.Cout) {
        if (res) {
          const uint4 rr = *reinterpret_cast<const uint4*>(res + (size_t)m * p.ldr + n);
          const T* re = reinterpret_cast<const T*>(&rr);
#pragma unroll
          for (int t = 0; t < OVE; ++t) v[t] += Elem<T>::ld(re + t);
        }
        uint4 o;
        OT* oe = reinterpret_cast<OT*>(&o);
#pragma unroll
        for (int t = 0; t < OVE; ++t) Elem<OT>::st(oe + t, act(v[t]));
        *reinterpret_cast<uint4*>(out + (size_t)m * p.ldo + n) = o;
      } else {
        for (int t = 0; t < OVE && n + t < p.Cout; ++t) {
          float x = v[t];
          if (res) x += Elem<T>::ld(res + (size_t)m * p.ldr + n + t);
          x = act(x);
          Elem<OT>::st(out + (size_t)m * p.ldo + n + t, x);
        }
      }
    }
  }
}

template <typename T, typename OT, int BM, int BN, bool PS = false>
int launch(const ConvParams& p, hipStream_t st) {
  const int ntm = cdiv(p.M, BM), ntn = cdiv(p.Cout, BN);
  const size_t smem = 2 * (BM + BN) * LDS_STRIDE;
  // set on every launch: a per-process flag would miss the second device of a multi-GPU process (cheap host call)
  (void)hipFuncSetAttribute((const void*)igemm_kernel<T, OT, BM, BN, PS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL((igemm_kernel<T, OT, BM, BN, PS>), dim3(ntm * ntn, 1, p.ksplit), dim3(NTHREADS), smem, st, p);
  return mega_check_launch();
}

// out[m][n] = act((sum_z partial[z][m][n]) * scale[n] + bias[n] + res[m][n]); fixed summation order -> deterministic
template <typename T, typename OT>
__global__ __launch_bounds__(256) void splitk_finalize_kernel(ConvParams p) {
  const size_t total = (size_t)p.M * p.Cout;
  const float neg_slope = p.relu == 1 ? 0.f : (p.relu == 2 ? 0.1f : 1.f);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / p.Cout), n = (int)(i - (size_t)m * p.Cout);
    float v = 0.f;
    for (int z = 0; z < p.ksplit; ++z) v += p.partial[(size_t)z * total + i];
    v = v * (p.scale ? p.scale[n] : 1.f) + (p.bias ? p.bias[n] : 0.f);
    if (p.res) v += Elem<T>::ld((const T*)p.res + (size_t)m * p.ldr + n);
    v = v > 0.f ? v : v * neg_slope;
    if (p.ps) {                                   // sub-pixel output (igemm_params.h)
      const int HoWo = p.Ho * p.Wo, t = m / HoWo, rem = m - t * HoWo, mh = rem / p.Wo, mw = rem - mh * p.Wo;
      const int q = n / p.ps_C, y = 2 * mh + (q >> 1) - p.ps_crop, x = 2 * mw + (q & 1) - p.ps_crop;
      if ((unsigned)y < (unsigned)p.ps_H && (unsigned)x < (unsigned)p.ps_W)
        Elem<OT>::st((OT*)p.out + ((size_t)(t * p.ps_H + y) * p.ps_W + x) * p.ldo + p.ps_coff + (n - q * p.ps_C), v);
      continue;
    }
    Elem<OT>::st((OT*)p.out + (size_t)m * p.ldo + n, v);
  }
}

template <typename T, typename OT>
int launch_finalize(const ConvParams& p, hipStream_t st) {
  const size_t total = (size_t)p.M * p.Cout;
  const unsigned blocks = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL((splitk_finalize_kernel<T, OT>), dim3(blocks), dim3(256), 0, st, p);
  return mega_check_launch();
}

// Tile choice, from measurements on MI355X (tools/bench_kernels.py --tiles ..., profiles/README.md): 128x128 at two
// blocks per CU wins every shape of the path except the very long-K 1024->1024 3x3 RPN conv, where 256x256 at one
// block per CU is ~10 % faster; 256x128 never wins (one block of 4 waves per CU hides too little latency).  Small
// grids shrink the tile to keep >= ~1.5 rounds of blocks over the 256 CUs.
// Split-K depends on K ALONE (never on M): a row of a layer is then summed in the same order whatever batch it is part
// of, which keeps results batch-invariant (tests: reference call convention == batched engine, bit for bit).  Only
// the box head's first FC (K = 100352 on R-101) qualifies: three ranges of 523 K-tiles (240 blocks on 256 CUs at 3750 rows).
inline int choose_ksplit(int K) { return K >= 32768 ? 3 : 1; }   // (3 x 20 x 4 = 240 blocks of 192 rows for the first FC of a 20-frame batch)

// kind 0: igemm_kernel<.., bm, bn> (this file);  kind 8: igemm8_kernel (igemm8.hip: LDS-DMA, 8 waves, bm x 256), bf16 only.
// MEGA_IGEMM_TILE forces a choice (experiments / tests): "128x64" or "8:256" / "8:192".
inline void choose_tile(int M, int Cout, int K, int z, bool bf16, int& kind, int& bm, int& bn) {
  kind = 0;
  const char* force = getenv("MEGA_IGEMM_TILE");
  if (force && sscanf(force, "8:%d", &bm) == 1) { kind = 8; bn = 256; return; }
  if (force && sscanf(force, "4:%d", &bm) == 1) { kind = 4; bn = 256; return; }
  if (force && sscanf(force, "2:%d", &bm) == 1) { kind = 2; bm = 128; bn = 256; return; }
  if (force && sscanf(force, "s:%d", &bm) == 1) { kind = 1; bm = 32; bn = 256; return; }
  if (force && sscanf(force, "%dx%d", &bm, &bn) == 2) return;
  const long b256 = (long)cdiv(M, 256) * cdiv(Cout, 256) * z;
  const long b128 = (long)cdiv(M, 128) * cdiv(Cout, 128) * z;
  const long b12864 = (long)cdiv(M, 128) * cdiv(Cout, 64) * z;
  static const int use8 = getenv("MEGA_IGEMM8") ? atoi(getenv("MEGA_IGEMM8")) : 1;
  // K = 64 (one K-tile: layer1's 64 -> 256 convs) runs on igemm8 too, bit-identically and 8 % faster (0.164 -> 0.151 ms
  // per 20 frames each).  Rounds 1-2 kept them on the 128x128 tiles so that the igemm8 symbols of a profile stayed the
  // matrix-core-bound layers; since round 3 the streaming layers have their own igemm8 symbol (CLS = 1), so they go over by
  // default (MEGA_IGEMM8_MIN_KTILES=2 restores the old dispatch).
  static const int min_kt = getenv("MEGA_IGEMM8_MIN_KTILES") ? atoi(getenv("MEGA_IGEMM8_MIN_KTILES")) : 1;
  if (bf16 && use8 && Cout >= 256 && K >= 64 * min_kt) {
    // igemm8 runs one block per CU: pick the row count that wastes the fewest CU-rounds (cost ~ rounds x rows)
    const long t256 = (long)cdiv(M, 256) * cdiv(Cout, 256) * z, t192 = (long)cdiv(M, 192) * cdiv(Cout, 256) * z;
    const long c256 = cdiv((int)t256, 256) * 256L * 8, c192 = cdiv((int)t192, 256) * 192L * 9;   // 192: ~12 % slower per row
    if (t256 >= 128 || t192 >= 128) {
      kind = 8; bn = 256; bm = c192 < c256 ? 192 : 256;
      return;
    }
  }
  if (K >= 8192 && Cout >= 1024 && b256 >= 700) { bm = 256; bn = 256; }
  else if (Cout > 64 && b128 >= 384) { bm = 128; bn = 128; }
  else if (b12864 >= 384) { bm = 128; bn = 64; }
  else { bm = 64; bn = 64; }
}

template <typename T, typename OT>
int dispatch_tile(const ConvParams& p, hipStream_t st) {
  int kind = 0, bm = 0, bn = 0, rc;
  constexpr bool is_bf16 = sizeof(T) == 2;                     // (a 16-bit operand type: bf16 or f16)
  choose_tile(p.M, p.Cout, p.K, p.ksplit, is_bf16, kind, bm, bn);
  if (kind == 8 && !(is_bf16 && mega_igemm8_supports(p))) {    // forced onto a shape it cannot take
    kind = 0; bm = 128; bn = 128;
  }
  if constexpr (is_bf16 && sizeof(OT) == 2) {
    // stream1x1 (persistent, wave-owned 32 x 256 tiles, weights resident in LDS): layer3's / layer2's conv3.  Measured SLOWER than
    // the tile kernels (0.173 against 0.118 ms on layer3's conv3: its row-strided 8-byte epilogue accesses and 32-byte operand
    // pieces bind the CU's address path, profiles/r06_streaming_class_experiments.txt) -- opt-in with MEGA_STREAM1X1=1;
    // MEGA_IGEMM_TILE=s:32 forces it (shapes it does not take fall through)
    static const int use_s = getenv("MEGA_STREAM1X1") ? atoi(getenv("MEGA_STREAM1X1")) : 0;
    if ((kind == 1 || (kind == 8 && use_s && !getenv("MEGA_IGEMM_TILE"))) && mega_stream1x1_supports(p, 0))
      return mega_stream1x1_launch(p, Half16<T>::CODE, st);
  }
  if (kind == 1) { kind = is_bf16 && mega_igemm8_supports(p) ? 8 : 0; bm = kind ? 256 : 128; bn = kind ? 256 : 128; }
  if constexpr (is_bf16) {
    // igemm2 (4 waves, 128 x 256 tile, K-tile 32, TWO blocks per CU).  Measured (profiles/r06_streaming_class_experiments.txt):
    // +6-10 % on layer2's conv3 (K = 128), +-1 % on layer3's conv3, slower on everything with a long K loop -- opt-in:
    // MEGA_IGEMM2=1 the streaming class (1x1, K <= 512) with K <= 128, =2 the whole streaming class, =3 every igemm8 launch it
    // supports; MEGA_IGEMM_TILE=2:128 forces it.  Default 0: the product path stays on igemm8.
    static const int use2 = getenv("MEGA_IGEMM2") ? atoi(getenv("MEGA_IGEMM2")) : 0;
    {
      const bool streaming2 = mega_igemm8_streaming(p.R * p.S, p.K) && p.ksplit == 1;
      if (kind == 2 || (kind == 8 && use2 && (use2 > 2 || (streaming2 && (use2 > 1 || p.K <= 128))) && !getenv("MEGA_IGEMM_TILE"))) {
        if (mega_igemm2_supports(p, sizeof(OT) == 4)) return mega_igemm2_launch(p, sizeof(OT) == 4, Half16<T>::CODE, st);
        if (kind == 2) kind = mega_igemm8_supports(p) ? 8 : 0;
        if (kind == 0) { bm = 128; bn = 128; } else if (bm != 192) bm = 256;
      }
    }
    // igemm4 (4 waves x 512 registers, 128 x 128 outputs per wave): the matrix-core-bound launch class -- 3x3 convs and every
    // layer with more than 8 K-tiles -- when its epilogue serves the shape; MEGA_IGEMM4=0 keeps everything on igemm8,
    // MEGA_IGEMM4=2 also sends the streaming class (1x1, K <= 512) over; MEGA_IGEMM_TILE=4:256 / 4:192 forces it
    static const int use4 = getenv("MEGA_IGEMM4") ? atoi(getenv("MEGA_IGEMM4")) : 0;
    const bool streaming = mega_igemm8_streaming(p.R * p.S, p.K) && p.ksplit == 1;
    if (kind == 4 || (kind == 8 && use4 && (use4 > 1 || !streaming) && !getenv("MEGA_IGEMM_TILE"))) {
      if (mega_igemm4_supports(p, sizeof(OT) == 4)) {
        rc = mega_igemm4_launch(p, bm == 192 ? 192 : 256, sizeof(OT) == 4, Half16<T>::CODE, st);
        if (rc == MEGA_OK && p.ksplit > 1) rc = launch_finalize<T, OT>(p, st);
        return rc;
      }
      if (kind == 4) kind = mega_igemm8_supports(p) ? 8 : 0;
      if (kind == 0) { bm = 128; bn = 128; }
    }
    if (kind == 8) {
      rc = mega_igemm8_launch(p, bm, sizeof(OT) == 4, Half16<T>::CODE, st);
      if (rc == MEGA_OK && p.ksplit > 1) rc = launch_finalize<T, OT>(p, st);
      return rc;
    }
  }
  if (kind == 8 || kind == 4 || kind == 2) rc = MEGA_ERR_ARG;
  else if (bm == 256 && bn == 256) rc = launch<T, OT, 256, 256>(p, st);
  else if (bm == 256 && bn == 128) rc = launch<T, OT, 256, 128>(p, st);
  else if (bm == 128 && bn == 128) rc = launch<T, OT, 128, 128>(p, st);
  else if (bm == 128 && bn == 64) rc = launch<T, OT, 128, 64>(p, st);
  else rc = launch<T, OT, 64, 64>(p, st);
  if (rc == MEGA_OK && p.ksplit > 1) rc = launch_finalize<T, OT>(p, st);
  return rc;
}

}  // namespace

// kind * 1e6 + bm * 1e3 + bn of the kernel a launch is dispatched to.  kinds: 0 igemm_kernel (this file), 8 igemm8 matrix
// class, 7 igemm8 streaming class (1x1, K <= 512), 6 conv64.hip.  The shape-complete form asks the SAME predicates the
// launch path uses (mega_conv64_supports, mega_igemm8_supports): profiler families, the roofline attribution and
// mega_conv2d_nhwc_tile name the kernel that really runs (the (M, Cout, K) form guessed conv64 from K = 576).
static int plan_of(const ConvParams& p, bool bf16_in, bool f32_out) {
  if (bf16_in && !f32_out && !getenv("MEGA_IGEMM_TILE") && mega_conv64_supports(p, 0)) return 6 * 1000000 + 256 * 1000 + 64;
  int kind = 0, bm = 0, bn = 0;
  choose_tile(p.M, p.Cout, p.K, choose_ksplit(p.K), bf16_in, kind, bm, bn);
  if (kind == 1 || kind == 8) {      // stream1x1 first, then igemm2 (the launch path's own rule: dispatch_tile)
    static const int use_s = getenv("MEGA_STREAM1X1") ? atoi(getenv("MEGA_STREAM1X1")) : 0;
    ConvParams q = p;
    q.ksplit = choose_ksplit(p.K);
    if (bf16_in && !f32_out && (kind == 1 || (use_s && !getenv("MEGA_IGEMM_TILE"))) && mega_stream1x1_supports(q, 0)) return 1 * 1000000 + 32 * 1000 + 256;
    if (kind == 1) { kind = 8; bm = 256; bn = 256; }
  }
  if (kind == 2 || kind == 8) {      // igemm2 first (the launch path's own rule: dispatch_tile)
    static const int use2 = getenv("MEGA_IGEMM2") ? atoi(getenv("MEGA_IGEMM2")) : 0;
    ConvParams q = p;
    q.ksplit = choose_ksplit(p.K);
    const bool streaming2 = mega_igemm8_streaming(p.R * p.S, p.K) && q.ksplit == 1;
    if (bf16_in && (kind == 2 || (use2 && (use2 > 2 || (streaming2 && (use2 > 1 || p.K <= 128))) && !getenv("MEGA_IGEMM_TILE"))) && mega_igemm2_supports(q, f32_out))
      return 2 * 1000000 + 128 * 1000 + 256;
    if (kind == 2) { kind = 8; bm = 256; }
  }
  if ((kind == 8 || kind == 4) && !(bf16_in && mega_igemm8_supports(p))) { kind = 0; bm = 128; bn = 128; }
  if (kind == 8 || kind == 4) {      // (the launch path's own rule: dispatch_tile)
    static const int use4 = getenv("MEGA_IGEMM4") ? atoi(getenv("MEGA_IGEMM4")) : 0;
    const bool streaming = mega_igemm8_streaming(p.R * p.S, p.K) && choose_ksplit(p.K) == 1;
    ConvParams q = p;
    q.ksplit = choose_ksplit(p.K);
    const bool take4 = (kind == 4 || (use4 && (use4 > 1 || !streaming) && !getenv("MEGA_IGEMM_TILE"))) && mega_igemm4_supports(q, f32_out || q.ksplit > 1);
    if (take4) kind = streaming ? 3 : 4;     // 4: igemm4 matrix class, 3: igemm4 streaming class
    else kind = streaming ? 7 : 8;
    if (bm != 192) bm = 256;
  }
  return kind * 1000000 + bm * 1000 + bn;
}

extern "C" int mega_conv2d_nhwc_plan_ex(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int dil,
                                        int ldo, int has_residual, int in_dtype, int out_dtype) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 || dil <= 0 || pad < 0) return -1;
  ConvParams p = {};
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.R = R; p.S = S; p.stride = stride; p.pad = pad; p.dil = dil;
  p.Ho = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1;
  p.Wo = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
  if (p.Ho <= 0 || p.Wo <= 0) return -1;
  p.M = N * p.Ho * p.Wo;
  p.K = R * S * Cin;
  p.ldo = ldo > 0 ? ldo : Cout;
  p.ldr = p.ldo;
  p.res = has_residual ? (const void*)&p : nullptr;      // (only tested against null)
  p.ksplit = 1;
  const size_t esz = in_dtype == MEGA_F32 ? 4 : 2;
  const size_t ib = (size_t)N * H * W * Cin * esz, wb = (size_t)Cout * p.K * esz;
  p.in_bytes = ib >= 0xFFFFFFF0ull ? 0xFFFFFFF0u : (unsigned)ib;
  p.w_bytes = wb >= 0xFFFFFFF0ull ? 0xFFFFFFF0u : (unsigned)wb;
  return plan_of(p, in_dtype != MEGA_F32, out_dtype == MEGA_F32);
}

// the (M, Cout, K) form: a 1x1 layer / linear of that GEMM shape (bf16 or f32 output does not change the tile)
extern "C" int mega_conv2d_nhwc_plan(int M, int Cout, int K, int in_dtype) {
  return mega_conv2d_nhwc_plan_ex(M, 1, 1, K, Cout, 1, 1, 1, 0, 1, Cout, 0, in_dtype, in_dtype);
}

extern "C" int mega_conv2d_nhwc_tile(int M, int Cout, int K) { return mega_conv2d_nhwc_plan(M, Cout, K, MEGA_BF16) % 1000000; }

extern "C" size_t mega_conv2d_nhwc_workspace_bytes(int M, int Cout, int K) {
  const int z = choose_ksplit(K);
  return z > 1 ? (size_t)z * M * Cout * sizeof(float) : 0;
}

// the largest split count <= want whose every K range holds at least one K-tile (the kernels assume nkt >= 1)
static int clamp_ksplit(int K, int in_dtype, int want) {
  const int nkt = K / (in_dtype == MEGA_F32 ? 32 : 64);
  int z = want < 1 ? 1 : (want > nkt ? nkt : want);
  while (z > 1 && (z - 1) * cdiv(nkt, z) >= nkt) --z;
  return z < 1 ? 1 : z;
}

extern "C" size_t mega_conv2d_nhwc_ks_workspace_bytes(int M, int Cout, int K, int in_dtype, int ksplit) {
  const int z = clamp_ksplit(K, in_dtype, ksplit);
  return z > 1 ? (size_t)z * M * Cout * sizeof(float) : 0;
}

static int conv2d_impl(const void* in, const void* w, const float* scale, const float* bias,
                       const void* residual, void* out, int N, int H, int W, int Cin, int Cout,
                       int R, int S, int stride, int pad, int dil, int relu, int ldo, int ldr,
                       int in_dtype, int out_dtype, void* ws, size_t ws_bytes, int ksplit, void* stream);

extern "C" int mega_conv2d_nhwc_ws(const void* in, const void* w, const float* scale, const float* bias,
                                   const void* residual, void* out, int N, int H, int W, int Cin, int Cout,
                                   int R, int S, int stride, int pad, int dil, int relu, int ldo, int ldr,
                                   int in_dtype, int out_dtype, void* ws, size_t ws_bytes, void* stream) {
  return conv2d_impl(in, w, scale, bias, residual, out, N, H, W, Cin, Cout, R, S, stride, pad, dil, relu, ldo, ldr, in_dtype,
                     out_dtype, ws, ws_bytes, 0, stream);
}

// mega_conv2d_nhwc_ws with the split count chosen by the CALLER (small-M, long-K layers whose tiles would not fill the chip:
// FlowNetS's coarse levels, the FGFA box head's fc6 on 300 rows).  ksplit >= 1 K ranges (clamped so that every range holds a
// K-tile); workspace of mega_conv2d_nhwc_ks_workspace_bytes.  The result depends on ksplit (summation order), not on M.
extern "C" int mega_conv2d_nhwc_ks(const void* in, const void* w, const float* scale, const float* bias,
                                   const void* residual, void* out, int N, int H, int W, int Cin, int Cout,
                                   int R, int S, int stride, int pad, int dil, int relu, int ldo, int ldr,
                                   int in_dtype, int out_dtype, int ksplit, void* ws, size_t ws_bytes, void* stream) {
  if (ksplit < 1) return MEGA_ERR_ARG;
  return conv2d_impl(in, w, scale, bias, residual, out, N, H, W, Cin, Cout, R, S, stride, pad, dil, relu, ldo, ldr, in_dtype,
                     out_dtype, ws, ws_bytes, ksplit, stream);
}

static int conv2d_impl(const void* in, const void* w, const float* scale, const float* bias,
                       const void* residual, void* out, int N, int H, int W, int Cin, int Cout,
                       int R, int S, int stride, int pad, int dil, int relu, int ldo, int ldr,
                       int in_dtype, int out_dtype, void* ws, size_t ws_bytes, int ksplit, void* stream) {
  mega_clear_error();
  if (!in || !w || !out || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 ||
      dil <= 0 || pad < 0)
    return MEGA_ERR_ARG;
  ConvParams p = {};
  p.in = in; p.w = w; p.scale = scale; p.bias = bias; p.res = residual; p.out = out;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.R = R; p.S = S;
  p.stride = stride; p.pad = pad; p.dil = dil;
  p.Ho = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1;
  p.Wo = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
  if (p.Ho <= 0 || p.Wo <= 0) return MEGA_ERR_ARG;
  p.M = N * p.Ho * p.Wo;
  p.K = R * S * Cin;
  p.ldo = ldo > 0 ? ldo : Cout;
  p.ldr = ldr > 0 ? ldr : Cout;
  p.relu = relu;
  p.ksplit = 1;
  p.partial = nullptr;
  if (ksplit > 0 && in_dtype != MEGA_F32 && in_dtype != MEGA_BF16 && in_dtype != MEGA_F16) return MEGA_ERR_ARG;
  if (ksplit > 0 || ws) {   // with a workspace the K range of long-K layers is split (mega_conv2d_nhwc_workspace_bytes)
    const int z = ksplit > 0 ? clamp_ksplit(p.K, in_dtype, ksplit) : choose_ksplit(p.K);
    if (z > 1) {
      if (!ws) return MEGA_ERR_ARG;
      if (ws_bytes < (size_t)z * p.M * Cout * sizeof(float)) return MEGA_ERR_ARG;
      p.ksplit = z;
      p.partial = (float*)ws;
    }
  }
  {
    if (in_dtype != MEGA_F32 && in_dtype != MEGA_BF16 && in_dtype != MEGA_F16) return MEGA_ERR_ARG;
    const size_t esz = in_dtype == MEGA_F32 ? 4 : 2;
    const size_t ib = (size_t)N * H * W * Cin * esz, wb = (size_t)Cout * p.K * esz;
    if (ib >= 0xFFFFFFF0ull || wb >= 0xFFFFFFF0ull) return MEGA_ERR_ARG;  // 32-bit buffer offsets
    p.in_bytes = (unsigned)ib;
    p.w_bytes = (unsigned)wb;
  }
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == MEGA_BF16) {
    if (Cin % 64 != 0) return MEGA_ERR_ARG;
    // layer1's 3x3 64 -> 64 conv: its own persistent streaming kernel (a forced tile, MEGA_IGEMM_TILE, keeps it on the
    // generic path: that is how the bit-equality test compares the two)
    if (out_dtype == MEGA_BF16 && !getenv("MEGA_IGEMM_TILE") && mega_conv64_supports(p, 0)) return mega_conv64_launch(p, MEGA_BF16, st);
    if (out_dtype == MEGA_BF16) return dispatch_tile<bf16_t, bf16_t>(p, st);
    if (out_dtype == MEGA_F32) return dispatch_tile<bf16_t, float>(p, st);
    return MEGA_ERR_ARG;
  }
  if (in_dtype == MEGA_F16) {      // IEEE half operands: the same kernels instantiated for f16_t (same tiles, same K order)
    if (Cin % 64 != 0) return MEGA_ERR_ARG;
    if (out_dtype == MEGA_F16 && !getenv("MEGA_IGEMM_TILE") && mega_conv64_supports(p, 0)) return mega_conv64_launch(p, MEGA_F16, st);
    if (out_dtype == MEGA_F16) return dispatch_tile<f16_t, f16_t>(p, st);
    if (out_dtype == MEGA_F32) return dispatch_tile<f16_t, float>(p, st);
    return MEGA_ERR_ARG;
  }
  if (in_dtype == MEGA_F32) {
    if (Cin % 32 != 0 || out_dtype != MEGA_F32) return MEGA_ERR_ARG;
    return dispatch_tile<float, float>(p, st);
  }
  return MEGA_ERR_ARG;
}

extern "C" int mega_conv2d_nhwc(const void* in, const void* w, const float* scale, const float* bias,
                                const void* residual, void* out, int N, int H, int W, int Cin, int Cout,
                                int R, int S, int stride, int pad, int dil, int relu, int ldo, int ldr,
                                int in_dtype, int out_dtype, void* stream) {
  return mega_conv2d_nhwc_ws(in, w, scale, bias, residual, out, N, H, W, Cin, Cout, R, S, stride, pad, dil, relu, ldo,
                             ldr, in_dtype, out_dtype, nullptr, 0, stream);
}

// ConvTranspose2d(Cin -> C, kernel 4, stride 2, no padding) + bias + activation, cropped, written into a channel slice of an
// NHWC tensor -- FlowNetS's refinement deconvolutions (mega_core/modeling/backbone/flownet.py:40-52 deconv5..deconv2 and
// :94-111: crop_like + torch.cat) without the zero-stuffed input (4x the matrix work) and without the crop / cat copies.
// Output pixel (2 m + a, 2 n + b) of the full (2H + 2) x (2W + 2) map only sees input pixels (m - dy, n - dx), dy, dx in {0, 1},
// through kernel taps (a + 2 dy, b + 2 dx): the four phases (a, b) are ONE 2 x 2 / pad 1 convolution over the input with
// 4 C output columns ordered (a, b, co),  w4[(a*2 + b)*C + co][r][s][ci] = Wt[ci][co][a + 2 (1 - r)][b + 2 (1 - s)]
// (ops.pack_deconv4x4s2), whose epilogue scatters row (t, m, n), column (a, b, co) to
//   out[t][2 m + a - crop][2 n + b - crop][coff + co]      (dropped outside [0, out_H) x [0, out_W); ldo = pixel stride).
// bias4 f32 [4 C] (the bias repeated per phase) or NULL; relu as mega_conv2d_nhwc; dtype = operand AND output type
// (MEGA_BF16 / MEGA_F16: Cin % 64 == 0; MEGA_F32: Cin % 32 == 0); C, ldo, coff multiples of the 16-byte vector (8 / 4
// elements).  ksplit > 1: split-K as mega_conv2d_nhwc_ks (workspace mega_conv2d_nhwc_ks_workspace_bytes(N (H+1) (W+1), 4 C,
// 4 Cin, dtype, ksplit)).
extern "C" int mega_conv2d_nhwc_subpixel(const void* in, const void* w4, const float* bias4, void* out, int N, int H, int W,
                                         int Cin, int C, int relu, int out_H, int out_W, int crop, int ldo, int coff, int dtype,
                                         int ksplit, void* ws, size_t ws_bytes, void* stream) {
  mega_clear_error();
  if (!in || !w4 || !out || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || C <= 0 || out_H <= 0 || out_W <= 0 || crop < 0 ||
      ksplit < 1 || coff < 0)
    return MEGA_ERR_ARG;
  if (dtype != MEGA_F32 && dtype != MEGA_BF16 && dtype != MEGA_F16) return MEGA_ERR_ARG;
  const size_t esz = dtype == MEGA_F32 ? 4 : 2;
  const int ove = (int)(16 / esz);
  if (Cin % (dtype == MEGA_F32 ? 32 : 64) != 0 || C % ove != 0 || ldo % ove != 0 || coff % ove != 0 || coff + C > ldo) return MEGA_ERR_ARG;
  ConvParams p = {};
  p.in = in; p.w = w4; p.scale = nullptr; p.bias = bias4; p.res = nullptr; p.out = out;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = 4 * C; p.R = 2; p.S = 2; p.stride = 1; p.pad = 1; p.dil = 1;
  p.Ho = H + 1; p.Wo = W + 1;
  p.M = N * p.Ho * p.Wo;
  p.K = 4 * Cin;
  p.ldo = ldo; p.ldr = ldo; p.relu = relu;
  p.ps = 1; p.ps_H = out_H; p.ps_W = out_W; p.ps_C = C; p.ps_crop = crop; p.ps_coff = coff;
  const size_t ib = (size_t)N * H * W * Cin * esz, wb = (size_t)p.Cout * p.K * esz;
  if (ib >= 0xFFFFFFF0ull || wb >= 0xFFFFFFF0ull || (size_t)N * out_H * out_W * ldo * esz >= 0x7FF00000ull) return MEGA_ERR_ARG;
  p.in_bytes = (unsigned)ib;
  p.w_bytes = (unsigned)wb;
  p.ksplit = clamp_ksplit(p.K, dtype, ksplit);
  p.partial = nullptr;
  if (p.ksplit > 1) {
    if (!ws || ws_bytes < (size_t)p.ksplit * p.M * p.Cout * sizeof(float)) return MEGA_ERR_ARG;
    p.partial = (float*)ws;
  }
  hipStream_t st = (hipStream_t)stream;
  const long b128 = (long)cdiv(p.M, 128) * cdiv(p.Cout, 128) * p.ksplit, b12864 = (long)cdiv(p.M, 128) * cdiv(p.Cout, 64) * p.ksplit;
  // (the sub-pixel read-out is the tiles' vector path: whole tiles of columns only)
  if (p.Cout % 64 != 0) return MEGA_ERR_ARG;
  // large launches on the LDS-DMA tiles (igemm8.hip, ABL = 6 = the same read-out there): choose_tile's rule -- Cout a multiple of
  // 256 and at least half a round of tiles; same K order, same MFMA: the same bits as the register-staged tiles
  if (dtype != MEGA_F32 && p.ksplit == 1 && p.Cout % 256 == 0 && !getenv("MEGA_IGEMM_TILE") && mega_igemm8_supports(p)) {
    const long t256 = (long)cdiv(p.M, 256) * (p.Cout / 256), t192 = (long)cdiv(p.M, 192) * (p.Cout / 256);
    const long c256 = cdiv((int)t256, 256) * 256L * 8, c192 = cdiv((int)t192, 256) * 192L * 9;
    if (t256 >= 128 || t192 >= 128) return mega_igemm8_launch(p, c192 < c256 ? 192 : 256, 0, dtype, st);
  }
  const int tile = (b128 >= 384 && p.Cout % 128 == 0) ? 0 : (b12864 >= 384 ? 1 : 2);     // (choose_tile's rule for the register-staged tiles)
  auto go = [&](auto tag) -> int {
    typedef decltype(tag) T;
    int rc = tile == 0 ? launch<T, T, 128, 128, true>(p, st) : (tile == 1 ? launch<T, T, 128, 64, true>(p, st) : launch<T, T, 64, 64, true>(p, st));
    if (rc == MEGA_OK && p.ksplit > 1) rc = launch_finalize<T, T>(p, st);
    return rc;
  };
  if (dtype == MEGA_BF16) return go(bf16_t{});
  if (dtype == MEGA_F16) return go(f16_t{});
  return go(float{});
}

// Split-precision activation planes (igemm_params.h, igemm8.hip SP kernels).  An f32 activation x [N,H,W,C] lives in HBM as
// bf16 [N,H,W,2C] = [hi | lo] (mega_split_f32_to_planes; ~2^-17 relative).  This entry point runs conv + FrozenBN (+ split
// residual) + activation on such tensors with the bf16 matrix cores:
//   in       bf16 [N,H,W,ldi]; the contraction runs over Cin channels per tap whose SOURCE channel is k < kwrap ? k : k - kwrap
//            (kwrap = 0: no wrap).  Split precision ("bf16 x 3"): ldi = 2C, Cin = 3C, kwrap = 2C, w = [Wh | Wh | Wl] per tap
//            ([Cout,R,S,3C], ops.split_weight_bf16x3): x_hi.Wh + x_lo.Wh + x_hi.Wl = x.W to ~2^-16, f32 accumulation.
//            bf16 compute on a wide residual stream: ldi = 2C, Cin = C, kwrap = 0, plain bf16 weights (the hi plane is read).
//   residual split planes [M][ldr] (ldr >= 2 Cout) or null: hi + lo is added in f32 before the activation
//   out_mode 0: bf16 [M][ldo];  1: split planes [M][ldo] (ldo >= 2 Cout);  2: f32 [M][ldo]
// Cin % 64 == 0, Cout % 8 == 0, every tensor below 2 GiB.  ws: split-K workspace as for mega_conv2d_nhwc_ws (f32 output, no
// residual).  Replaces, in the split-precision parity mode, the same reference layers as mega_conv2d_nhwc
// (backbone/resnet.py:324-344, rpn/rpn.py:99-106, roi_box_feature_extractors.py:894,:907).
// dtype (MEGA_BF16 / MEGA_F16): the 16-bit type of the planes and weights.  MEGA_F16: IEEE-half pairs -- the two-pass form of
// the fp16 mode (conv_mode "h2": ldi = 2C, Cin = 2C, kwrap = 0, w = [W | W] per tap with W rounded to fp16 ONCE: x_hi.W + x_lo.W,
// i.e. exact activations against single-rounded weights at twice the matrix-core work).
extern "C" int mega_conv2d_nhwc_sp_dt(const void* in, int ldi, int kwrap, const void* w, const float* scale, const float* bias,
                                      const void* residual, int ldr, void* out, int ldo, int out_mode, int N, int H, int W,
                                      int Cin, int Cout, int R, int S, int stride, int pad, int dil, int relu, int dtype,
                                      void* ws, size_t ws_bytes, void* stream) {
  mega_clear_error();
  if (dtype != MEGA_BF16 && dtype != MEGA_F16) return MEGA_ERR_ARG;
  if (!in || !w || !out || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 ||
      dil <= 0 || pad < 0 || out_mode < 0 || out_mode > 2 || ldi <= 0 || kwrap < 0 || Cin % 64 != 0)
    return MEGA_ERR_ARG;
  ConvParams p = {};
  p.in = in; p.w = w; p.scale = scale; p.bias = bias; p.res = residual; p.out = out;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.R = R; p.S = S;
  p.stride = stride; p.pad = pad; p.dil = dil;
  p.Ho = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1;
  p.Wo = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
  if (p.Ho <= 0 || p.Wo <= 0) return MEGA_ERR_ARG;
  p.M = N * p.Ho * p.Wo;
  p.K = R * S * Cin;
  p.sp = 1; p.ldi = ldi; p.kwrap = kwrap; p.split_out = out_mode == 1;
  p.ldo = ldo > 0 ? ldo : (out_mode == 1 ? 2 * Cout : Cout);
  p.ldr = ldr > 0 ? ldr : 2 * Cout;
  p.relu = relu;
  p.ksplit = 1;
  p.partial = nullptr;
  if (ws) {
    const int z = choose_ksplit(p.K);
    if (z > 1) {
      if (ws_bytes < (size_t)z * p.M * Cout * sizeof(float) || out_mode != 2 || residual) return MEGA_ERR_ARG;
      p.ksplit = z;
      p.partial = (float*)ws;
    }
  }
  {
    const size_t ib = (size_t)N * H * W * ldi * 2, wb = (size_t)Cout * p.K * 2;
    if (ib >= 0x7FF00000ull || wb >= 0x7FF00000ull) return MEGA_ERR_ARG;  // 32-bit buffer offsets
    p.in_bytes = (unsigned)ib;
    p.w_bytes = (unsigned)wb;
  }
  hipStream_t st = (hipStream_t)stream;
  // one block per CU: the row count that wastes the fewest CU-rounds (the rule of choose_tile)
  const long t256 = (long)cdiv(p.M, 256) * cdiv(Cout, 256) * p.ksplit, t192 = (long)cdiv(p.M, 192) * cdiv(Cout, 256) * p.ksplit;
  const long c256 = cdiv((int)t256, 256) * 256L * 8, c192 = cdiv((int)t192, 256) * 192L * 9;
  int rc = mega_igemm8_launch(p, c192 < c256 ? 192 : 256, out_mode == 2, dtype, st);
  if (rc == MEGA_OK && p.ksplit > 1) rc = launch_finalize<bf16_t, float>(p, st);      // (no residual in split-K launches: type-free)
  return rc;
}

extern "C" int mega_conv2d_nhwc_sp(const void* in, int ldi, int kwrap, const void* w, const float* scale, const float* bias,
                                   const void* residual, int ldr, void* out, int ldo, int out_mode, int N, int H, int W,
                                   int Cin, int Cout, int R, int S, int stride, int pad, int dil, int relu, void* ws,
                                   size_t ws_bytes, void* stream) {
  return mega_conv2d_nhwc_sp_dt(in, ldi, kwrap, w, scale, bias, residual, ldr, out, ldo, out_mode, N, H, W, Cin, Cout, R, S, stride,
                                pad, dil, relu, MEGA_BF16, ws, ws_bytes, stream);
}
