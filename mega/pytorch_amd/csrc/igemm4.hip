// igemm4: the matrix-core-bound layers of the MEGA frame stage (3x3 convs, long-K 1x1 convs / linears: layer3 conv1 / conv2,
// the RPN conv, res5, fc0, the aggregation's projections) on ONE 512-register wave per SIMD.
//
// Same contraction, layouts, LDS staging and epilogue arithmetic as igemm8.hip (see igemm.hip's header for the reference
// layers: mega_core/modeling/backbone/resnet.py:324-344, rpn/rpn.py:99-106, roi_box_feature_extractors.py:894,:907); same MFMA
// (v_mfma_f32_32x32x16_bf16 / _f16, weight fragment first) and the same ascending K order per output element, so a row's
// bits do not depend on which kernel computed it (tests: bit equality against the register-staged tiles).
//
// What changes is the shape of a wave's work.  igemm8 runs 8 waves of 256 registers on a BM x 256 tile: a wave owns a
// 64 x 32 piece of each of the four (A half, B half) quadrants -- 128 x 64 outputs -- and reads 24 16-byte fragments from LDS
// per 32 MFMAs (0.75 per MFMA); its K loop is co-limited by LDS bandwidth (192 KB of fragment reads + 64 KB of DMA writes
// per K-tile = the 2048 MFMA cycles) and sits at ~80 % of the MFMA rate.  Here 4 waves (2 x 2) own 128 x 128 outputs each
// (96 x 128 on the 192-row tile): MFW + 4 fragment reads per 4 MFW MFMAs (0.5 / 0.58 per MFMA), 128 KB of LDS reads per
// K-tile, the 256 (192) accumulator registers in the upper half of the 512-register file.  With one wave per SIMD nothing
// else hides latency, so the loop is software-pipelined inside the wave: the fragments of K-step s + 1 are requested in
// the MFMA gaps of K-step s (two register sets), a K-tile's DMA is issued in the gaps of the last K-step of the tile
// before the previous one.  One barrier per K-tile.
//
// LDS (128 KiB ring, the layout of igemm8): [A0 A1 B0 B1][parity 2][128 rows][128 B]; a row's eight 16-byte chunks are
// XOR-swizzled (physical = logical ^ ((row >> 1) & 7)) on the DMA's source side.  Wave (wr, wc) reads rows wr * 32 MFW ..
// of A and the whole half B_wc.
// Per K-tile t (parity p = t & 1), per wave:
//   K-steps 0, 1:  MFMAs on one fragment set, the reads of the next K-step into the other
//   K-step 2:      the same; then lgkmcnt(0) [this wave is done with slot p], vmcnt(0) [its pieces of tile t + 1 have landed],
//                  s_barrier [... for every wave]
//   K-step 3:      MFMAs; reads of (t + 1, K-step 0) from parity p ^ 1; the 16 DMA pieces of tile t + 2 into parity p
// A piece is issued ~1.25 K-tiles (> 2500 cycles) before the wait that retires it.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "igemm_params.h"

namespace {

constexpr int NT4 = 256;
constexpr int ROWB4 = 128;
constexpr int HALF4 = 128 * ROWB4;                 // one half-tile slot (16 KiB)
constexpr int LDS4 = 8 * HALF4;                    // 128 KiB ring
constexpr int CST4 = 256 + 4;                      // f32 row stride of a staged slab
constexpr int LDS4_ALLOC = 2 * 64 * CST4 * 4 > LDS4 ? 2 * 64 * CST4 * 4 : LDS4;     // two 64-row staging slabs (133 120 B)
__host__ __device__ constexpr int slot4_a(int i, int par) { return (i * 2 + par) * HALF4; }
__host__ __device__ constexpr int slot4_b(int j, int par) { return 4 * HALF4 + (j * 2 + par) * HALF4; }
constexpr unsigned OOB4 = 0x80000000u;

typedef __attribute__((address_space(3))) void* lds4_ptr_t;

struct KPos4 {
  int kc, ks, kr, dh, dw;
  unsigned uni;
};

__device__ __forceinline__ int fast_div4(int n, unsigned mg, unsigned sh) {
  return (int)((__umulhi((unsigned)n, mg) + (unsigned)n) >> sh);
}

// OT: output type (HT or float).  MFW: 32-row fragments per wave in M (4: 256-row tile, 3: 192-row tile).  CLS: launch class,
// part of the symbol only (0 matrix class, 1 streaming class: see igemm8.hip).  HT: bf16_t / f16_t.
// ABL != 0: timing-only ablations of the K loop (results are garbage by construction; experiments build only, tools/gpu/
// igemm4_check.py --ablate): 1 no DMA issue in the loop, 2 no barrier, 3 no vmcnt wait, 4 no fragment reads, 5 no MFMAs
template <typename OT, int MFW, int CLS, typename HT, int ABL = 0>
__global__ __launch_bounds__(NT4) __attribute__((amdgpu_waves_per_eu(1, 1))) void igemm4_kernel(ConvParams p) {
  static_assert(sizeof(OT) == 4 || std::is_same<OT, HT>::value, "16-bit outputs have the operands' type");
  constexpr int BM = 64 * MFW;
  constexpr int BN = 256;
  constexpr int RW = 32 * MFW;                     // rows per wave row
  constexpr int NA1 = MFW == 4 ? 4 : 2;            // 32-row DMA passes of half-tile A1 (rows 128 .. BM-1)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // ---- XCD-aware block -> tile map (as igemm8)
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

  // ---- staging descriptors: thread (prow = tid >> 3 in 0..31, pch = tid & 7) fetches, for each half-tile, the pieces
  //      (row prow + 32 u, physical chunk pch), u = 0..3; the logical chunk it reads is pch ^ swizzle(row) (32 u keeps it)
  const int prow = tid >> 3, pch = tid & 7;
  const unsigned lcb = (unsigned)((pch ^ ((prow >> 1) & 7)) * 16);
  int a_hi0[8], a_wi0[8];
  unsigned a_off[8];
  {
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (r >= 4 + NA1) { a_hi0[r] = -(1 << 20); a_wi0[r] = 0; a_off[r] = 0; continue; }
      const int m = m0 + (r >> 2) * 128 + prow + 32 * (r & 3);
      const bool ok = m < p.M;
      const int mm = ok ? m : 0;
      const int nimg = fast_div4(mm, p.mg_howo, p.sh_howo);
      const int rem = mm - nimg * HoWo;
      const int ho = fast_div4(rem, p.mg_wo, p.sh_wo);
      const int wo = rem - ho * p.Wo;
      a_hi0[r] = ok ? ho * p.stride - p.pad : -(1 << 20);
      a_wi0[r] = wo * p.stride - p.pad;
      a_off[r] = ((unsigned)(nimg * p.H * p.W) + (unsigned)(a_hi0[r] * p.W + a_wi0[r])) * (unsigned)(p.Cin * 2) + lcb;
    }
  }
  unsigned b_off[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int n = n0 + (r >> 2) * 128 + prow + 32 * (r & 3);
    b_off[r] = n < p.Cout ? (unsigned)n * (unsigned)(p.K * 2) + lcb : OOB4;
  }

  // ---- K range of this block (split-K: blockIdx.z owns a contiguous range of K-tiles)
  const int nkt_all = p.K >> 6;
  const int kt_per = (nkt_all + p.ksplit - 1) / p.ksplit;
  const int kt0 = (int)blockIdx.z * kt_per;
  const int nkt = min(kt_per, nkt_all - kt0);
  KPos4 pa;                                         // K position of the next K-tile to be issued
  {
    const int kk = kt0 * 64;
    pa.kc = kk % p.Cin;
    pa.ks = (kk / p.Cin) % p.S;
    pa.kr = (kk / p.Cin) / p.S;
    pa.dh = pa.kr * p.dil;
    pa.dw = pa.ks * p.dil;
    pa.uni = (unsigned)(((pa.dh * p.W + pa.dw) * p.Cin + pa.kc) * 2);
  }
  auto kpos_next = [&](KPos4& s) {
    s.kc += 64;
    if (s.kc >= p.Cin) {
      s.kc = 0;
      s.dw += p.dil;
      if (++s.ks == p.S) { s.ks = 0; s.dw = 0; ++s.kr; s.dh += p.dil; }
    }
    s.uni = (unsigned)(((s.dh * p.W + s.dw) * p.Cin + s.kc) * 2);
  };
  int tnext = 0;                                    // K-tile (relative to kt0) of the next issue

  unsigned char* const wbase = smem + wave * 1024;  // a wave's 8 rows x 128 B of every 32-row pass
  // one 1-KiB piece per wave: piece q of a K-tile, q = 0 .. 7: A (half q >> 2, pass q & 3), q = 8 .. 15: B likewise.
  // Tiles past the end of K are still "issued" with every lane out of range (zeros into a dead slot: uniform vmcnt bookkeeping).
  // The per-lane source offset of a piece is computed AHEAD of its issue (piece_off, in the nearly empty MFMA gaps of a
  // tile's first three K-steps) and the issue itself is two instructions (m0, buffer_load ... lds): with the ~10 address
  // instructions in the gap that also carries the DMA, a gap was 11 instructions against the ~5 an MFMA hides.
  auto piece_off = [&](int q, const KPos4& s, int tile, bool live) -> unsigned {
    const unsigned dead = live ? 0u : OOB4;
    if (q < 8) {
      const int hi = a_hi0[q] + s.dh, wi = a_wi0[q] + s.dw;
      const bool ok = ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);
      return (ok ? a_off[q] + s.uni : OOB4) | dead;
    }
    return (b_off[q - 8] + (unsigned)((kt0 + tile) * 128)) | dead;
  };
  auto issue_off = [&](int q, int par, unsigned off) {
    if (q < 8) {
      const int i = q >> 2, u = q & 3;
      if (i == 1 && u >= NA1) return;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds4_ptr_t)(wbase + slot4_a(i, par) + u * 4096), 16, off, 0, 0, 0);
    } else {
      const int j = (q - 8) >> 2, u = q & 3;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds4_ptr_t)(wbase + slot4_b(j, par) + u * 4096), 16, off, 0, 0, 0);
    }
  };
  auto issue_piece = [&](int q, int par, const KPos4& s, int tile, bool live) { issue_off(q, par, piece_off(q, s, tile, live)); };
  constexpr int NPIECE = 12 + NA1;                  // DMA instructions per wave and K-tile

  // ---- fragment read addresses: row (lane & 31) of fragment f, K-step ks: logical chunk 2 ks + (lane >> 5), swizzled
  const int l31 = lane & 31;
  unsigned a_rd[MFW][4], b_rd[4][4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const unsigned ch = (unsigned)(((ks * 2 + (lane >> 5)) ^ ((l31 >> 1) & 7)) * 16);
#pragma unroll
    for (int f = 0; f < MFW; ++f) {
      const int row = wr * RW + f * 32;             // first tile row of the fragment: inside ONE half-tile (multiples of 32)
      a_rd[f][ks] = (unsigned)(slot4_a(row >> 7, 0) + ((row & 127) + l31) * ROWB4) + ch;
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) b_rd[f][ks] = (unsigned)(slot4_b(wc, 0) + (f * 32 + l31) * ROWB4) + ch;
  }

  f32x16_t acc[MFW][4];
#pragma unroll
  for (int f = 0; f < MFW; ++f)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][j][r] = 0.f;

  u32x4_t af[2][MFW], bfr[2][4];
#define M4_RD(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory")
#define M4_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define M4_WAIT_LDS()                                   \
  do {                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);                  \
  } while (0)
#define M4_BAR()                      \
  do {                                \
    asm volatile("" ::: "memory");   \
    __builtin_amdgcn_s_barrier();     \
    asm volatile("" ::: "memory");   \
  } while (0)
#define M4_SB() __builtin_amdgcn_sched_barrier(0)

  // ---- prologue: tiles 0 and 1 in flight, tile 0 landed, its first K-step in registers
#pragma unroll
  for (int q = 0; q < 16; ++q) issue_piece(q, 0, pa, tnext, tnext < nkt);
  kpos_next(pa); ++tnext;
#pragma unroll
  for (int q = 0; q < 16; ++q) issue_piece(q, 1, pa, tnext, tnext < nkt);
  kpos_next(pa); ++tnext;
  M4_WAIT_VM(NPIECE);
  M4_BAR();
#pragma unroll
  for (int i = 0; i < MFW; ++i) M4_RD(af[0][i], a_rd[i][0], 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) M4_RD(bfr[0][i], b_rd[i][0], 0);
  M4_WAIT_LDS();

  // one K-step: 4 MFW MFMAs on fragment set `cur`; the MFW + 4 reads of the NEXT K-step (K-step rks of the tile in parity
  // rpar) into set cur ^ 1 in the first MFMA gaps; MODE 0 / 1 / 2 (a tile's first three K-steps): the source offsets of
  // pieces 6 MODE .. of the tile after next are computed in the gaps; MODE 3 (its last K-step): those pieces are issued
  unsigned poff[16], psel[8];
  auto kstep = [&](auto CUR, auto RPAR, auto RKS, auto MODE_, int dpar, bool live) {
    constexpr int cur = decltype(CUR)::value, rpar = decltype(RPAR)::value, rks = decltype(RKS)::value;
    constexpr int mode = decltype(MODE_)::value;
    constexpr int NM = 4 * MFW, NR = MFW + 4;
    constexpr int per = (16 + NM - 1) / NM;         // DMA pieces per gap (192-row tile: 12 gaps, two pieces in the first eight)
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      const int f = i / 4, j = i % 4;
      if (ABL != 5) acc[f][j] = Half16<HT>::mfma32(bfr[cur][j], af[cur][f], acc[f][j]);
      else asm volatile("" ::"v"(bfr[cur][j]), "v"(af[cur][f]));
      if (ABL != 4) {
        if (i < MFW) M4_RD(af[cur ^ 1][i < MFW ? i : 0], a_rd[i < MFW ? i : 0][rks], rpar * HALF4);
        else if (i < NR) M4_RD(bfr[cur ^ 1][(i - MFW) & 3], b_rd[(i - MFW) & 3][rks], rpar * HALF4);
      }
      if (mode < 3) {
        // in the gaps the reads leave empty (i >= NR: 8 / 5 per K-step).  256-row tile: an A piece's offset in two halves
        // (range test, then the sum: 5 + 3 instructions -- a whole one is 8, more than an MFMA hides), a B piece in one
        constexpr int FREE = NM - NR;
        const int slot = mode * FREE + (i - NR);
        if (i >= NR) {
          if (MFW == 4) {
            if (slot < 16) {
              const int q = slot >> 1;
              if ((slot & 1) == 0) {
                const int hi = a_hi0[q] + pa.dh, wi = a_wi0[q] + pa.dw;
                const bool ok = ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);
                psel[q] = ok ? 0u : OOB4;
                asm volatile("" : "+v"(psel[q]));   // (pinned here: the optimiser would sink the computation to its use)
              } else {
                poff[q] = (a_off[q] + pa.uni) | psel[q] | (live ? 0u : OOB4);
                asm volatile("" : "+v"(poff[q]));
              }
            } else if (slot < 24) {
              poff[slot - 8] = piece_off(slot - 8, pa, tnext, live);
              asm volatile("" : "+v"(poff[slot - 8]));
            }
          } else {
            const int q = slot < 6 ? slot : slot + 2;
            if (q < 16) {
              poff[q] = piece_off(q, pa, tnext, live);
              asm volatile("" : "+v"(poff[q]));
            }
          }
        }
      } else if (ABL != 1) {
#pragma unroll
        for (int u = 0; u < per; ++u)
          if (i * per + u < 16) issue_off(i * per + u, dpar, poff[i * per + u]);
      }
      M4_SB();
    }
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  auto ktile = [&](auto PAR) {
    constexpr int par = decltype(PAR)::value;
    using P = std::integral_constant<int, par>;
    using Q = std::integral_constant<int, par ^ 1>;
    const bool live = tnext < nkt;
    kstep(I0{}, P{}, I1{}, I0{}, 0, live);
    M4_WAIT_LDS();
    kstep(I1{}, P{}, I2{}, I1{}, 0, live);
    M4_WAIT_LDS();
    kstep(I0{}, P{}, I3{}, I2{}, 0, live);
    M4_WAIT_LDS();                                  // this wave has read everything it needs from parity `par`
    if (ABL != 3 && ABL != 1) M4_WAIT_VM(0);        // its pieces of the next tile (parity par ^ 1) have landed
    if (ABL != 2) M4_BAR();
    kstep(I1{}, Q{}, I0{}, I3{}, par, live);
    kpos_next(pa); ++tnext;
    M4_WAIT_LDS();
  };
  const int nkt2 = (nkt + 1) & ~1;                  // an odd tail tile runs on all-zero operands (its DMAs are out of range)
  for (int t = 0; t < nkt2; t += 2) {
    ktile(I0{});
    ktile(I1{});
  }
  M4_WAIT_VM(0);                                    // the out-of-range tail DMAs also write (zeros) into the LDS re-used below
  M4_BAR();

  // ---- epilogue.  acc[f][j]: lane owns output row (lane & 31) of fragment f and, for g = 0..3, the four consecutive
  //      channels 32 j + 8 g + 4 (lane >> 5) + (0..3) of the wave's 128 columns.  One 64-row slab per fragment index f:
  //      the f-th 32 rows of BOTH wave rows x all 256 columns, staged as f32 (4 ds_write_b128 per accumulator) and read
  //      back as 16-byte vectors along n, where FrozenBN scale / bias (one fma per element), the residual and the
  //      activation are applied -- the arithmetic of igemm8's epilogue, hence its bits.
  float* cs = reinterpret_cast<float*>(smem);
  OT* __restrict__ out = (OT*)p.out;
  constexpr int OVE = 16 / (int)sizeof(OT);
  constexpr int VPR = BN / OVE;
  constexpr int NIT = 64 * VPR / NT4;               // 16-byte output vectors per thread per slab (8 half / 16 f32)
  constexpr int RSTEP = NT4 / VPR;                  // slab rows between a thread's consecutive vectors (8 / 4)
  auto stage = [&](float* csb, int f) {
    float* dst = csb + (wr * 32 + l31) * CST4 + wc * 128 + 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_t q = {acc[f][j][4 * g], acc[f][j][4 * g + 1], acc[f][j][4 * g + 2], acc[f][j][4 * g + 3]};
        *reinterpret_cast<f32x4_t*>(dst + j * 32 + 8 * g) = q;
      }
  };
  if (p.ksplit > 1) {                               // raw partial sums; splitk_finalize_kernel (igemm.hip) finishes
    float* part = p.partial + (size_t)blockIdx.z * p.M * p.Cout;
#pragma unroll
    for (int f = 0; f < MFW; ++f) {
      if (f > 0) { M4_WAIT_LDS(); M4_BAR(); }
      stage(cs, f);
      M4_WAIT_LDS();
      M4_BAR();
      for (int e = tid; e < 64 * (BN / 4); e += NT4) {
        const int row = e / (BN / 4), cv = e - row * (BN / 4);
        const int m = m0 + (row >> 5) * RW + f * 32 + (row & 31), n = n0 + cv * 4;
        if (m >= p.M || n >= p.Cout) continue;
        const float4 v = *reinterpret_cast<const float4*>(cs + row * CST4 + cv * 4);
        if (n + 4 <= p.Cout && p.Cout % 4 == 0) {
          *reinterpret_cast<float4*>(part + (size_t)m * p.Cout + n) = v;
        } else {
          const float vv[4] = {v.x, v.y, v.z, v.w};
          for (int t = 0; t < 4 && n + t < p.Cout; ++t) part[(size_t)m * p.Cout + n + t] = vv[t];
        }
      }
    }
    return;
  }
  // (mega_igemm4_supports admits only launches that qualify for this buffer-addressed epilogue: Cout a multiple of 256, 16-byte
  //  rows, tensors below 2 GiB, a residual of the operands' type only with a 16-bit output)
  const bool has_res = p.res != nullptr;
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
      p.out, 0, (int)(((size_t)(p.M - 1) * p.ldo + p.Cout) * sizeof(OT)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.res ? p.res : p.out), 0, has_res ? (int)(((size_t)(p.M - 1) * p.ldr + p.Cout) * 2) : 0, 0x00020000);
  const int row0 = tid / VPR, ncol = n0 + (tid % VPR) * OVE;
  auto slab_m = [&](int f, int it) { return (m0 + row0) + (((it * RSTEP) >> 5) * RW + f * 32 + ((it * RSTEP) & 31)); };
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
  int slab1 = 64 * CST4;
  asm volatile("" : "+v"(slab1));                   // (opaque: folded into the ds offsets it would exceed their 16 bits)
  auto run = [&](auto HR, auto RL) {
    constexpr bool HAS_RES = decltype(HR)::value;
    constexpr bool RELU = decltype(RL)::value;
    u32x4_t rr[2][HAS_RES ? NIT : 1];               // the residual vectors of two slabs, rolling
    auto ldres = [&](int f, u32x4_t (&dst)[HAS_RES ? NIT : 1]) {
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        dst[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, (unsigned)(slab_m(f, it) * p.ldr + ncol) * 2u, 0, 0);
    };
    if constexpr (HAS_RES) {
      ldres(0, rr[0]);
      ldres(1, rr[1]);
    }
#pragma unroll
    for (int f = 0; f < MFW; ++f) {
      float* csb = (f & 1) ? cs + slab1 : cs;
      stage(csb, f);
      M4_WAIT_LDS();
      M4_BAR();
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int row = row0 + it * RSTEP;
        float v[OVE];
#pragma unroll
        for (int t = 0; t < OVE; t += 4) {
          const float4 q4 = *reinterpret_cast<const float4*>(csb + row * CST4 + (tid % VPR) * OVE + t);
          v[t] = fmaf(q4.x, scv[t], biv[t]); v[t + 1] = fmaf(q4.y, scv[t + 1], biv[t + 1]);
          v[t + 2] = fmaf(q4.z, scv[t + 2], biv[t + 2]); v[t + 3] = fmaf(q4.w, scv[t + 3], biv[t + 3]);
        }
        if constexpr (HAS_RES) {
          u32x4_t r4 = rr[f & 1][it];
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
      if constexpr (HAS_RES) {
        if (f + 2 < MFW) ldres(f + 2, rr[f & 1]);   // slab f + 2 into the registers slab f just freed
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
#undef M4_RD
#undef M4_WAIT_VM
#undef M4_WAIT_LDS
#undef M4_BAR
#undef M4_SB
}

inline void magic_div4(int d, unsigned& mg, unsigned& sh) {
  sh = 0;
  while ((1ull << sh) < (unsigned long long)d) ++sh;
  mg = (unsigned)((((1ull << sh) - (unsigned long long)d) << 32) / (unsigned long long)d + 1ull);
}

template <typename OT, int MFW, int CLS, typename HT, int ABL = 0>
int launch4(const ConvParams& p0, hipStream_t st) {
  constexpr int BM = 64 * MFW;
  ConvParams p = p0;
  magic_div4(p.Ho * p.Wo, p.mg_howo, p.sh_howo);
  magic_div4(p.Wo, p.mg_wo, p.sh_wo);
  const int ntm = cdiv(p.M, BM), ntn = cdiv(p.Cout, 256);
  (void)hipFuncSetAttribute((const void*)igemm4_kernel<OT, MFW, CLS, HT, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS4_ALLOC);
  hipLaunchKernelGGL((igemm4_kernel<OT, MFW, CLS, HT, ABL>), dim3(ntm * ntn, 1, p.ksplit), dim3(NT4), LDS4_ALLOC, st, p);
  return mega_check_launch();
}

template <typename HT>
int launch4_any(const ConvParams& p, int bm, int out_f32, hipStream_t st) {
  const bool stream = mega_igemm8_streaming(p.R * p.S, p.K) && p.ksplit == 1;
  if (bm == 256 && stream) return out_f32 ? launch4<float, 4, 1, HT>(p, st) : launch4<HT, 4, 1, HT>(p, st);
  if (bm == 192 && stream) return out_f32 ? launch4<float, 3, 1, HT>(p, st) : launch4<HT, 3, 1, HT>(p, st);
  if (bm == 256) return out_f32 ? launch4<float, 4, 0, HT>(p, st) : launch4<HT, 4, 0, HT>(p, st);
  if (bm == 192) return out_f32 ? launch4<float, 3, 0, HT>(p, st) : launch4<HT, 3, 0, HT>(p, st);
  return MEGA_ERR_ARG;
}

}  // namespace

// 1 when igemm4 takes this launch: what igemm8 takes, restricted to the shapes its buffer-addressed epilogue serves (every
// layer of the frame stage) or raw split-K partial sums; no split-precision planes.
int mega_igemm4_supports(const ConvParams& p, int out_f32) {
  if (p.sp || !mega_igemm8_supports(p)) return 0;
  if (p.ksplit > 1) return p.res == nullptr;
  const size_t osz = out_f32 ? 4 : 2;
  if (p.Cout % 256 != 0 || p.ldo % (int)(16 / osz) != 0) return 0;
  if (((size_t)(p.M - 1) * p.ldo + p.Cout) * osz >= 0x7FF00000ull) return 0;
  if (p.res && (out_f32 || p.ldr % 8 != 0 || ((size_t)(p.M - 1) * p.ldr + p.Cout) * 2 >= 0x7FF00000ull)) return 0;
  return 1;
}

int mega_igemm4_launch(const ConvParams& p, int bm, int out_f32, int half_dtype, hipStream_t st) {
#ifdef MEGA_EXPERIMENTS
  {      // timing ablations (garbage results): MEGA_IGEMM4_ABLATE=1..5, bf16 -> bf16, 256-row tile, matrix class only
    const char* e = getenv("MEGA_IGEMM4_ABLATE");
    const int abl = e ? atoi(e) : 0;
    if (abl && bm == 256 && !out_f32 && half_dtype == MEGA_BF16 && p.ksplit == 1) {
      if (abl == 1) return launch4<bf16_t, 4, 0, bf16_t, 1>(p, st);
      if (abl == 2) return launch4<bf16_t, 4, 0, bf16_t, 2>(p, st);
      if (abl == 3) return launch4<bf16_t, 4, 0, bf16_t, 3>(p, st);
      if (abl == 4) return launch4<bf16_t, 4, 0, bf16_t, 4>(p, st);
      if (abl == 5) return launch4<bf16_t, 4, 0, bf16_t, 5>(p, st);
    }
  }
#endif
  if (half_dtype == MEGA_F16) return launch4_any<f16_t>(p, bm, out_f32, st);
  if (half_dtype == MEGA_BF16) return launch4_any<bf16_t>(p, bm, out_f32, st);
  return MEGA_ERR_ARG;
}
