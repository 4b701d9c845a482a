// igemm2: the STREAMING launch class of the MEGA frame stage -- 1x1 convs with K <= 512 (layer3's conv3 256 -> 1024 + residual,
// res5's conv3, layer2's conv3, the stride-1 projections) and, measured per shape, the short-K layers next to them -- at TWO
// blocks per CU.
//
// Same contraction, layouts and epilogue arithmetic as igemm8.hip / igemm4.hip (see igemm.hip's header for the reference layers:
// mega_core/modeling/backbone/resnet.py:324-344); same MFMA (v_mfma_f32_32x32x16_bf16 / _f16, weight fragment first) and the same
// ascending K order per output element, so a row's bits do not depend on which kernel computed it (tools/gpu/igemm2_check.py,
// tests/test_kernels_gpu.py: bit equality against the register-staged tiles).
//
// Why.  igemm8 owns a CU (128 KiB of LDS, 8 waves x 256 registers).  Its streaming launches have 2-8 K-tiles per output tile:
// the round-5 timeline of layer3's conv3 (profiles/r05_igemm8_tile_timeline.txt) is prologue 6.1 k cycles (the first DMA's HBM
// round trip), K loop 12.3 k (8.2 k of MFMAs), epilogue 17 k (residual rows in, output rows out) -- three phases that each wait
// on memory while nothing else runs on that CU: 4.0 TB/s of algorithmic bytes against ~6.3 TB/s achievable.  The K loop is not
// what limits these layers; the number of independent memory streams per CU is.  igemm2 halves the block (4 waves, 128 x 256
// outputs, K-tile 32 = 64-byte rows, 3-stage ring = 72 KiB) so that two blocks share a CU and one block's prologue / epilogue
// latencies run beside the other's K loop -- the hardware scheduler does the overlap that a persistent kernel would have to
// build by hand (round 3's persistent igemm8 with 32 KiB staging: neutral).
//
// Geometry.  Block tile 128 x 256, 4 waves (2 x 2), wave (wr, wc) owns rows wr * 64 + [0, 64) x cols wc * 128 + [0, 128):
//   2 x 4 fragments of 32 x 32 = 128 accumulator registers; per 16-deep K-step 6 fragment reads for 8 MFMAs.
// LDS ring: A stage s at s * 8 KiB ([128 rows][64 B]), B stage s at 24 KiB + s * 16 KiB ([256 rows][64 B]); a row's four
//   16-byte chunks are XOR-swizzled (physical = logical ^ ((row >> 2) & 3)) on the DMA's source side: the 16 lanes of a
//   ds_read_b128 group then touch 16 different bank quads (4 (row & 3) + chunk').
// Pipeline: tile t + 2 is issued after the barrier of tile t into the stage that tile t - 1 was read from; one barrier per
//   K-tile; vmcnt(6) = this wave's 6 pieces of tile t have landed (tile t + 1's 6 may still be in flight).
// Epilogue: igemm4's -- a 64-row f32 slab (the f-th 32 rows of both wave rows x 256 columns) staged in the ring's LDS,
//   read back as 16-byte vectors along n with FrozenBN scale / bias, residual and activation applied there.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "igemm_params.h"

namespace {

constexpr int NT2 = 256;
constexpr int KT2 = 32;                               // K-tile depth (elements)
constexpr int ROWB2 = KT2 * 2;                        // 64 bytes per LDS row
constexpr int ASTG = 128 * ROWB2;                     // one A stage (8 KiB)
constexpr int BSTG = 256 * ROWB2;                     // one B stage (16 KiB)
constexpr int BBASE = 3 * ASTG;                       // B stages start at 24 KiB
constexpr int RING2 = 3 * (ASTG + BSTG);              // 72 KiB
constexpr int CST2 = 256 + 4;                         // f32 row stride of the staged slab
constexpr int LDS2_ALLOC = 64 * CST2 * 4 > RING2 ? 64 * CST2 * 4 : RING2;
static_assert(2 * LDS2_ALLOC <= 160 * 1024, "two blocks per CU");
constexpr unsigned OOB2 = 0x80000000u;

typedef __attribute__((address_space(3))) void* lds2_ptr_t;

struct KPos2 {
  int kc, ks, kr, dh, dw;
  unsigned uni;
};

__device__ __forceinline__ int fast_div2(int n, unsigned mg, unsigned sh) {
  return (int)((__umulhi((unsigned)n, mg) + (unsigned)n) >> sh);
}

// OT: output type (HT or float).  CLS: launch class, part of the symbol only (1 streaming class, 0 matrix class).  HT: bf16_t / f16_t.
// PRE: the residual rows of slab 0 are requested BEFORE the K loop (32 more live registers through the loop).
template <typename OT, int CLS, typename HT, int PRE = 0>
__global__ __launch_bounds__(NT2, 2) void igemm2_kernel(ConvParams p) {
  static_assert(sizeof(OT) == 4 || std::is_same<OT, HT>::value, "16-bit outputs have the operands' type");
  constexpr int BM = 128, BN = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // ---- XCD-aware block -> tile map (as igemm8): the blocks of one XCD walk N first and share A row panels in that XCD's L2
  const int ntn = (p.Cout + BN - 1) / BN;
  int lid;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tile_m = lid / ntn, tile_n = lid - tile_m * ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);

  // ---- staging descriptors: thread (prow = tid >> 2 in 0..63, pch = tid & 3) fetches, for every 64-row piece u of a stage, the
  //      16 bytes (row prow + 64 u, physical chunk pch); the logical chunk it reads is pch ^ swizzle(row) (64 u keeps it)
  const int prow = tid >> 2, pch = tid & 3;
  const unsigned lcb = (unsigned)((pch ^ ((prow >> 2) & 3)) * 16);
  int a_hi0[2], a_wi0[2];
  unsigned a_off[2];
  {
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = m0 + u * 64 + prow;
      const bool ok = m < p.M;
      const int mm = ok ? m : 0;
      const int nimg = fast_div2(mm, p.mg_howo, p.sh_howo);
      const int rem = mm - nimg * HoWo;
      const int ho = fast_div2(rem, p.mg_wo, p.sh_wo);
      const int wo = rem - ho * p.Wo;
      a_hi0[u] = ok ? ho * p.stride - p.pad : -(1 << 20);      // a row past M never passes the range test below
      a_wi0[u] = wo * p.stride - p.pad;
      a_off[u] = ((unsigned)(nimg * p.H * p.W) + (unsigned)(a_hi0[u] * p.W + a_wi0[u])) * (unsigned)(p.Cin * 2) + lcb;
    }
  }
  unsigned b_off[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int n = n0 + u * 64 + prow;
    b_off[u] = n < p.Cout ? (unsigned)n * (unsigned)(p.K * 2) + lcb : OOB2;
  }

  const int nkt = p.K / KT2;
  KPos2 pa;                                          // K position of the next K-tile to be issued
  pa.kc = 0; pa.ks = 0; pa.kr = 0; pa.dh = 0; pa.dw = 0; pa.uni = 0;
  auto kpos_next = [&](KPos2& s) {
    s.kc += KT2;
    if (s.kc >= p.Cin) {
      s.kc = 0;
      s.dw += p.dil;
      if (++s.ks == p.S) { s.ks = 0; s.dw = 0; ++s.kr; s.dh += p.dil; }
    }
    s.uni = (unsigned)(((s.dh * p.W + s.dw) * p.Cin + s.kc) * 2);
  };
  int tnext = 0;

  unsigned char* const wbase = smem + wave * 1024;   // a wave's 16 rows x 64 B of every 64-row piece
  // Tiles past the end of K are still "issued" with every lane out of range (zeros into a dead stage: uniform vmcnt bookkeeping)
  auto issue_tile = [&](int sa, int sb) {            // sa / sb: byte offsets of the A / B stage
    const unsigned dead = tnext < nkt ? 0u : OOB2;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int hi = a_hi0[u] + pa.dh, wi = a_wi0[u] + pa.dw;
      const bool ok = ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);
      const unsigned off = (ok ? a_off[u] + pa.uni : OOB2) | dead;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds2_ptr_t)(wbase + sa + u * 4096), 16, off, 0, 0, 0);
    }
    const unsigned kb = (unsigned)(tnext * ROWB2);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned off = (b_off[u] + kb) | dead;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds2_ptr_t)(wbase + BBASE + sb + u * 4096), 16, off, 0, 0, 0);
    }
    kpos_next(pa);
    ++tnext;
  };

  // ---- fragment read addresses (stage 0): row (lane & 31) of the wave's first fragment, K-step ks: logical chunk
  //      2 ks + (lane >> 5), swizzled by the row; fragment f / j adds 32 rows = 2048 bytes (immediate)
  const int l31 = lane & 31;
  unsigned a_rd[2], b_rd[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const unsigned ch = (unsigned)(((ks * 2 + (lane >> 5)) ^ ((l31 >> 2) & 3)) * 16);
    a_rd[ks] = (unsigned)((wr * 64 + l31) * ROWB2) + ch;
    b_rd[ks] = (unsigned)(BBASE + (wc * 128 + l31) * ROWB2) + ch;
  }

  f32x16_t acc[2][4];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][j][r] = 0.f;

#define M2_RD(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory")
#define M2_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define M2_WAIT_LGKM(n)                                  \
  do {                                                   \
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory"); \
    __builtin_amdgcn_sched_barrier(0);                   \
  } while (0)
#define M2_BAR()                      \
  do {                                \
    asm volatile("" ::: "memory");   \
    __builtin_amdgcn_s_barrier();     \
    asm volatile("" ::: "memory");   \
  } while (0)
#define M2_SB() __builtin_amdgcn_sched_barrier(0)

  // ---- the residual rows of slab 0 ahead of the K loop (PRE): see the epilogue
  constexpr int OVE = 16 / (int)sizeof(OT);
  constexpr int VPR = BN / OVE;
  constexpr int NIT = 64 * VPR / NT2;                // 16-byte output vectors per thread per slab (8 half / 16 f32)
  constexpr int RSTEP = NT2 / VPR;                   // slab rows between a thread's consecutive vectors (8 / 4)
  const int row0 = tid / VPR, ncol = n0 + (tid % VPR) * OVE;
  auto slab_m = [&](int f, int it) { return (m0 + row0) + (((it * RSTEP) >> 5) * 64 + f * 32 + ((it * RSTEP) & 31)); };
  const bool has_res = p.res != nullptr;
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.res ? p.res : p.out), 0, has_res ? (int)(((size_t)(p.M - 1) * p.ldr + p.Cout) * 2) : 0, 0x00020000);
  u32x4_t rr0[(PRE && sizeof(OT) == 2) ? NIT : 1];
  if constexpr (PRE && sizeof(OT) == 2) {
#pragma unroll
    for (int it = 0; it < NIT; ++it)                 // (no residual: num_records 0, every load returns 0 without touching memory)
      rr0[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, (unsigned)(slab_m(0, it) * p.ldr + ncol) * 2u, 0, 0);
  }

  // ---- prologue: tiles 0 and 1 in flight
  issue_tile(0, 0);
  issue_tile(ASTG, BSTG);

  u32x4_t af[2][2], bfr[2][4];                       // [K-step][fragment]
  int sa = 0, sb = 0;                                // stage of tile t
  int na = 2 * ASTG, nb = 2 * BSTG;                  // stage tile t + 2 goes to
  for (int t = 0; t < nkt; ++t) {
    M2_WAIT_VM(6);                                   // this wave's pieces of tile t have landed (in-order return: the PRE loads are older)
    M2_WAIT_LGKM(0);                                 // ... and it has read everything it needs from tile t - 1
    M2_BAR();
    issue_tile(na, nb);                              // tile t + 2 into the stage tile t - 1 was read from
    const unsigned aa0 = a_rd[0] + (unsigned)sa, aa1 = a_rd[1] + (unsigned)sa;
    const unsigned ba0 = b_rd[0] + (unsigned)sb, ba1 = b_rd[1] + (unsigned)sb;
    M2_RD(af[0][0], aa0, 0); M2_RD(af[0][1], aa0, 2048);
    M2_RD(bfr[0][0], ba0, 0); M2_RD(bfr[0][1], ba0, 2048); M2_RD(bfr[0][2], ba0, 4096); M2_RD(bfr[0][3], ba0, 6144);
    M2_RD(af[1][0], aa1, 0); M2_RD(af[1][1], aa1, 2048);
    M2_RD(bfr[1][0], ba1, 0); M2_RD(bfr[1][1], ba1, 2048); M2_RD(bfr[1][2], ba1, 4096); M2_RD(bfr[1][3], ba1, 6144);
    M2_WAIT_LGKM(6);                                 // K-step 0's six fragments (LDS returns in order)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[f][j] = Half16<HT>::mfma32(bfr[0][j], af[0][f], acc[f][j]);
    M2_SB();
    M2_WAIT_LGKM(0);
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[f][j] = Half16<HT>::mfma32(bfr[1][j], af[1][f], acc[f][j]);
    M2_SB();
    sa = sa == 2 * ASTG ? 0 : sa + ASTG;
    sb = sb == 2 * BSTG ? 0 : sb + BSTG;
    na = na == 2 * ASTG ? 0 : na + ASTG;
    nb = nb == 2 * BSTG ? 0 : nb + BSTG;
  }
  M2_WAIT_VM(0);                                     // the out-of-range tail DMAs also write (zeros) into the LDS re-used below
  M2_BAR();

  // ---- epilogue.  acc[f][j]: lane owns output row (lane & 31) of fragment f and, for g = 0..3, the four consecutive
  //      channels 32 j + 8 g + 4 (lane >> 5) + (0..3) of the wave's 128 columns.  One 64-row slab per fragment index f.
  float* cs = reinterpret_cast<float*>(smem);
  auto stage = [&](int f) {
    float* dst = cs + (wr * 32 + l31) * CST2 + wc * 128 + 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_t q = {acc[f][j][4 * g], acc[f][j][4 * g + 1], acc[f][j][4 * g + 2], acc[f][j][4 * g + 3]};
        *reinterpret_cast<f32x4_t*>(dst + j * 32 + 8 * g) = q;
      }
  };
  // (mega_igemm2_supports admits only launches that qualify for this buffer-addressed epilogue: Cout a multiple of 256, 16-byte
  //  rows, tensors below 2 GiB, a residual of the operands' type only with a 16-bit output, no split-K)
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
      p.out, 0, (int)(((size_t)(p.M - 1) * p.ldo + p.Cout) * sizeof(OT)), 0x00020000);
  float scv[OVE], biv[OVE];
  {
    const bool al16 = ((reinterpret_cast<size_t>(p.scale) | reinterpret_cast<size_t>(p.bias)) & 15) == 0;
#pragma unroll
    for (int t = 0; t < OVE; t += 4) {
      f32x4_t s4 = {1.f, 1.f, 1.f, 1.f}, b4 = {0.f, 0.f, 0.f, 0.f};
      if (al16) {
        if (p.scale) s4 = *reinterpret_cast<const f32x4_t*>(p.scale + ncol + t);
        if (p.bias) b4 = *reinterpret_cast<const f32x4_t*>(p.bias + ncol + t);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (p.scale) s4[u] = p.scale[ncol + t + u];
          if (p.bias) b4[u] = p.bias[ncol + t + u];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { scv[t + u] = s4[u]; biv[t + u] = b4[u]; }
    }
  }
  const float neg_slope = p.relu == 1 ? 0.f : (p.relu == 2 ? 0.1f : 1.f);
  auto act = [&](float x) { return x > 0.f ? x : x * neg_slope; };
  auto run = [&](auto HR, auto RL) {
    constexpr bool HAS_RES = decltype(HR)::value;
    constexpr bool RELU = decltype(RL)::value;
    u32x4_t rr[2][HAS_RES ? NIT : 1];               // the residual vectors of the two slabs
    if constexpr (HAS_RES) {
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        if (PRE && f == 0) {
#pragma unroll
          for (int it = 0; it < NIT; ++it) rr[0][it] = rr0[PRE ? it : 0];
          continue;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it)
          rr[f][it] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, (unsigned)(slab_m(f, it) * p.ldr + ncol) * 2u, 0, 0);
      }
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      if (f > 0) { M2_WAIT_LGKM(0); M2_BAR(); }     // slab 0 has been read out by every wave
      stage(f);
      M2_WAIT_LGKM(0);
      M2_BAR();
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int row = row0 + it * RSTEP;
        float v[OVE];
#pragma unroll
        for (int t = 0; t < OVE; t += 4) {
          const float4 q4 = *reinterpret_cast<const float4*>(cs + row * CST2 + (tid % VPR) * OVE + t);
          v[t] = fmaf(q4.x, scv[t], biv[t]); v[t + 1] = fmaf(q4.y, scv[t + 1], biv[t + 1]);
          v[t + 2] = fmaf(q4.z, scv[t + 2], biv[t + 2]); v[t + 3] = fmaf(q4.w, scv[t + 3], biv[t + 3]);
        }
        if constexpr (HAS_RES) {
          u32x4_t r4 = rr[f][it];
          asm volatile("" : "+v"(r4));
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            v[2 * d] += Half16<HT>::lo(r4[d]);
            v[2 * d + 1] += Half16<HT>::hi(r4[d]);
          }
        }
        const unsigned ooff = (unsigned)(slab_m(f, it) * p.ldo + ncol) * (unsigned)sizeof(OT);
        u32x4_t o;
        if constexpr (sizeof(OT) == 2) {
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            if constexpr (RELU) {
              const s16x2_t z = {0, 0};
              o[d] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, Half16<HT>::pack2(v[2 * d], v[2 * d + 1])), z));
            } else {
              o[d] = Half16<HT>::pack2(act(v[2 * d]), act(v[2 * d + 1]));
            }
          }
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) o[t] = __float_as_uint(RELU ? fmaxf(v[t], 0.f) : act(v[t]));
        }
        __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, ooff, 0, 0);
      }
    }
  };
  if constexpr (sizeof(OT) == 2) {
    if (has_res) {
      if (p.relu == 1) run(std::true_type{}, std::true_type{}); else run(std::true_type{}, std::false_type{});
      return;
    }
  }
  if (p.relu == 1) run(std::false_type{}, std::true_type{}); else run(std::false_type{}, std::false_type{});
#undef M2_RD
#undef M2_WAIT_VM
#undef M2_WAIT_LGKM
#undef M2_BAR
#undef M2_SB
}

inline void magic_div2(int d, unsigned& mg, unsigned& sh) {
  sh = 0;
  while ((1ull << sh) < (unsigned long long)d) ++sh;
  mg = (unsigned)((((1ull << sh) - (unsigned long long)d) << 32) / (unsigned long long)d + 1ull);
}

template <typename OT, int CLS, typename HT, int PRE>
int launch2(const ConvParams& p0, hipStream_t st) {
  ConvParams p = p0;
  magic_div2(p.Ho * p.Wo, p.mg_howo, p.sh_howo);
  magic_div2(p.Wo, p.mg_wo, p.sh_wo);
  const int ntm = cdiv(p.M, 128), ntn = cdiv(p.Cout, 256);
  (void)hipFuncSetAttribute((const void*)igemm2_kernel<OT, CLS, HT, PRE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2_ALLOC);
  hipLaunchKernelGGL((igemm2_kernel<OT, CLS, HT, PRE>), dim3(ntm * ntn, 1, 1), dim3(NT2), LDS2_ALLOC, st, p);
  return mega_check_launch();
}

template <typename HT>
int launch2_any(const ConvParams& p, int out_f32, hipStream_t st) {
  static const int pre = getenv("MEGA_IGEMM2_PRE") ? atoi(getenv("MEGA_IGEMM2_PRE")) : 0;
  const bool stream = mega_igemm8_streaming(p.R * p.S, p.K);
  if (out_f32) return stream ? launch2<float, 1, HT, 0>(p, st) : launch2<float, 0, HT, 0>(p, st);
  if (pre && p.res) return stream ? launch2<HT, 1, HT, 1>(p, st) : launch2<HT, 0, HT, 1>(p, st);
  return stream ? launch2<HT, 1, HT, 0>(p, st) : launch2<HT, 0, HT, 0>(p, st);
}

}  // namespace

// 1 when igemm2 takes this launch: plain (not split-precision) 16-bit operands, Cin a multiple of 32, no split-K, and a shape
// its buffer-addressed epilogue serves (every Cout % 256 == 0 layer of the frame stage)
int mega_igemm2_supports(const ConvParams& p, int out_f32) {
  if (p.sp || p.ksplit != 1) return 0;
  if (p.Cin % KT2 != 0 || p.K < KT2 || p.in_bytes >= 0x7FF00000u || p.w_bytes >= 0x7FF00000u) return 0;
  const size_t osz = out_f32 ? 4 : 2;
  if (p.Cout % 256 != 0 || p.ldo % (int)(16 / osz) != 0) return 0;
  if (((size_t)(p.M - 1) * p.ldo + p.Cout) * osz >= 0x7FF00000ull) return 0;
  if (p.res && (out_f32 || p.ldr % 8 != 0 || ((size_t)(p.M - 1) * p.ldr + p.Cout) * 2 >= 0x7FF00000ull)) return 0;
  return 1;
}

int mega_igemm2_launch(const ConvParams& p, int out_f32, int half_dtype, hipStream_t st) {
  if (half_dtype == MEGA_F16) return launch2_any<f16_t>(p, out_f32, st);
  if (half_dtype == MEGA_BF16) return launch2_any<bf16_t>(p, out_f32, st);
  return MEGA_ERR_ARG;
}
