// Implicit-GEMM convolution / linear layer, bf16, for the big layers of the MEGA frame stage: 256-wide tiles,
// 8 waves, operands staged HBM -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`: no VGPR round trip, no ds_write),
// 8 half-tile slots in flight-order ring, counted vmcnt, raw s_barrier, two wave groups running half a phase apart so
// that one group's MFMA segment always runs beside the other group's LDS-read / DMA-issue segment.
//
// Same contraction, layouts and epilogue as igemm.hip (see its header for the reference layers this replaces:
// mega_core/modeling/backbone/resnet.py:324-344, rpn/rpn.py:99-106, roi_box_feature_extractors.py:894,:907).
// Same MFMA instruction (v_mfma_f32_32x32x16_bf16) and the same ascending-K order per output element as the
// igemm.hip tiles, so a row's result does not depend on which kernel / tile computed it (batch invariance).
//
// Geometry.  Block tile BM x 256 (BM = 256, or 192 with MF1 = 1), K-tile 64 bf16 = one 128-byte line per row.
//   half-tiles: A0 = rows 0..127, A1 = rows 128..BM-1, B0 = cols 0..127, B1 = cols 128..255 (16 KiB each);
//   wave (wr, wc) of a 2 x 4 grid owns, in every (Ai, Bj) quadrant of the tile, a 64(32) x 32 piece:
//     rows i*128 + wr*64 + [0,64)   (A1 with MF1 = 1: 128 + wr*32 + [0,32)),   cols j*128 + wc*32 + [0,32).
//   A K-tile is processed in 2 phases, A0 x (B0,B1) then A1 x (B1,B0); each half-tile is read from LDS exactly once
//   per K-tile, the B fragments stay in registers for the second phase.
// LDS (128 KiB): [A0 A1 B0 B1][parity 2][128 rows][128 B].  A row's eight 16-byte chunks are stored XOR-swizzled
//   (physical chunk = logical chunk ^ ((row >> 1) & 7)): every 16-lane group of a ds_read_b128 then touches 16
//   different bank quads.  The DMA writes lane-linearly, so the swizzle is applied to the per-lane SOURCE address;
//   the eight lanes of a row still read one whole 128-byte line.
// Pipeline.  Half-tiles are issued in the order of use A0(t) B0(t) B1(t) A1(t) A0(t+1) ..., three phases (1.5
//   K-tiles: > 2000 cycles) ahead of the phase that reads them.  A slot is re-filled one or two phases after its
//   last ds_read.  Every phase = [load segment: ds_reads of this phase, counted vmcnt] barrier
//   [MFMA segment with the phase's two DMA issues in the MFMA shadow] barrier; wave group 1 (waves 4-7, one per SIMD like group 0) runs one barrier behind group 0.
//   RAW: the wait that retires a half-tile sits before the barrier that ends the phase BEFORE the one that reads it
//   (for both groups); WAR: see the schedule table in DESIGN.md section 3.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "igemm_params.h"

namespace {

constexpr int NT8 = 512;
constexpr int ROWB = 128;              // bytes per LDS row = one K-tile of one row
constexpr int HALF = 128 * ROWB;       // one half-tile slot (16 KiB)
constexpr int LDS8 = 8 * HALF;         // 128 KiB: A slots in the first 64 KiB, B slots in the second (ds_read offsets are 16 bit)
constexpr int LDS8_ALLOC = 2 * 64 * (256 + 4) * 4 > LDS8 ? 2 * 64 * (256 + 4) * 4 : LDS8;   // + 2 KiB: the epilogue's two staging slabs
__host__ __device__ constexpr int slot_a(int i, int par) { return (i * 2 + par) * HALF; }
__host__ __device__ constexpr int slot_b(int j, int par) { return 4 * HALF + (j * 2 + par) * HALF; }
constexpr unsigned OOB = 0x80000000u;  // >= num_records of every operand (operands are < 2 GiB): the DMA writes zeros

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// scalar (wave-uniform) position of a K-tile inside the (r, s, c) loop of the implicit GEMM
struct KPos {
  int kc, ks, kr, dh, dw;
  unsigned uni;                        // ((dh * W + dw) * Cin + kc) * 2 bytes
};

// ABL != 0: timing-only ablations for tools/gpu (1: no fragment ds_reads in the loop, 2: no DMA in the loop, 3: no MFMA);
// results are garbage by construction.  Only ABL = 0 is instantiated unless the library is built with
// -DMEGA_EXPERIMENTS (see mega_igemm8_launch).
// CLS: launch class, part of the symbol only (the code is the same): 0 = matrix-core-bound layers (3x3 convs and every
// layer with more than 8 K-tiles), 1 = streaming layers (1x1 convs / linears with K <= 512: at most 8 K-tiles per output
// tile, so prologue, epilogue and the operand fetch set their time -- layer3's conv3, res5's conv3, layer2's 1x1s).  The
// two classes sit under different roofs (MFMA vs HBM / CU fetch rate); as ONE symbol their rocprofv3 average blends
// 1290 TF/s (RPN conv) with 480 TF/s (layer3 conv3) and says nothing about either.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

// ABL == 5: whole-tile timeline of EVERY block (waves 0 and 4, one per MFMA group): 16 u64 slots per wave in p.partial
// (tools/gpu/timeline8.py): s_memtime (shader cycles) at 0 entry, 1 K loop starts, 2 K loop done, 3 + 2s slab s staged,
// 4 + 2s slab s stores issued, 11 stores acknowledged; 12 = HW_ID | XCC_ID << 32; 13 / 14 = s_memrealtime (100 MHz) at
// entry / end.  The results stay correct.
#define MEGA_TS(k)                                                                                                      \
  do {                                                                                                                  \
    if (ABL == 5 && (wave & 3) == 0) {                                                                                  \
      const unsigned long long tt = __builtin_amdgcn_s_memtime();                                                       \
      if (lane == 0) reinterpret_cast<unsigned long long*>(p.partial)[((size_t)blockIdx.x * 2 + (wave >> 2)) * 16 + (k)] = tt; \
    }                                                                                                                   \
  } while (0)
#define MEGA_TS_RT(k)                                                                                                   \
  do {                                                                                                                  \
    if (ABL == 5 && (wave & 3) == 0) {                                                                                  \
      const unsigned long long tt = __builtin_amdgcn_s_memrealtime();                                                   \
      if (lane == 0) reinterpret_cast<unsigned long long*>(p.partial)[((size_t)blockIdx.x * 2 + (wave >> 2)) * 16 + (k)] = tt; \
    }                                                                                                                   \
  } while (0)

// SP != 0: split-precision activation planes (igemm_params.h: ldi, kwrap, split_out, split residual) -- its own symbols, the
// plain kernels' code does not change.
__device__ __forceinline__ int fast_div(int n, unsigned mg, unsigned sh) {   // n / d for 0 <= n < 2^31 (launch8 computes mg, sh)
  return (int)((__umulhi((unsigned)n, mg) + (unsigned)n) >> sh);
}

// HT: the 16-bit operand type (bf16_t, or f16_t = IEEE half: round 6) of in / w / residual; OT = HT or float.
template <typename OT, int MF1, int CLS = 0, int ABL = 0, int SP = 0, typename HT = bf16_t>
__global__ __launch_bounds__(NT8, 2) void igemm8_kernel(ConvParams p) {
  static_assert(sizeof(OT) == 4 || std::is_same<OT, HT>::value, "16-bit outputs have the operands' type");
  constexpr int BM = 128 + 64 * MF1;
  constexpr int BN = 256;
  constexpr int WROWS1 = 32 * MF1;     // rows a wave owns in A1
  constexpr int CA1 = MF1;             // DMA instructions per wave for A1 (A0, B0, B1: 2 each)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  MEGA_TS(0);
  MEGA_TS_RT(13);

  // ---- XCD-aware block -> tile map (bijective for any grid size): the blocks of one XCD walk N first, so they share
  //      A row panels in that XCD's L2
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

  // ---- staging descriptors: thread (prow, pch) fetches, for each half-tile, the pieces (row prow + 64 u, physical
  //      chunk pch), u = 0, 1; the logical chunk it reads from memory is pch ^ swizzle(row)  (64 u does not change it)
  const int prow = tid >> 3, pch = tid & 7;
  const unsigned lcb = (unsigned)((pch ^ ((prow >> 1) & 7)) * 16);
  const int ldi = SP ? p.ldi : p.Cin;          // pixel stride of the input in elements
  const int kwrap = SP && p.kwrap > 0 ? p.kwrap : 0x7fffffff;
  int a_hi0[4], a_wi0[4];
  unsigned a_off[4];
  {
    // (m -> (image, ho, wo) by magic-number multiplication: the eight integer divisions this replaces were ~1 k cycles
    //  of every tile's prologue -- ~35 VALU instructions each, before the first DMA could be issued)
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (MF1 == 1 && r == 3) { a_hi0[r] = -(1 << 20); a_wi0[r] = 0; a_off[r] = 0; continue; }
      const int m = m0 + (r >> 1) * 128 + prow + 64 * (r & 1);
      const bool ok = m < p.M;
      const int mm = ok ? m : 0;
      const int nimg = fast_div(mm, p.mg_howo, p.sh_howo);
      const int rem = mm - nimg * HoWo;
      const int ho = fast_div(rem, p.mg_wo, p.sh_wo);
      const int wo = rem - ho * p.Wo;
      a_hi0[r] = ok ? ho * p.stride - p.pad : -(1 << 20);      // a row past M never passes the range test below
      a_wi0[r] = wo * p.stride - p.pad;
      a_off[r] = ((unsigned)(nimg * p.H * p.W) + (unsigned)(a_hi0[r] * p.W + a_wi0[r])) * (unsigned)(ldi * 2) + lcb;
    }
  }
  unsigned b_off[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + (r >> 1) * 128 + prow + 64 * (r & 1);
    b_off[r] = n < p.Cout ? (unsigned)n * (unsigned)(p.K * 2) + lcb : OOB;
  }

  // ---- K range of this block (split-K: blockIdx.z owns a contiguous range of K-tiles)
  const int nkt_all = p.K >> 6;
  const int kt_per = (nkt_all + p.ksplit - 1) / p.ksplit;
  const int kt0 = (int)blockIdx.z * kt_per;
  const int nkt = min(kt_per, nkt_all - kt0);
  auto kpos_init = [&](KPos& s) {
    const int kk = kt0 * 64;
    s.kc = kk % p.Cin;
    s.ks = (kk / p.Cin) % p.S;
    s.kr = (kk / p.Cin) / p.S;
    s.dh = s.kr * p.dil;
    s.dw = s.ks * p.dil;
    s.uni = (unsigned)(((s.dh * p.W + s.dw) * ldi + (s.kc >= kwrap ? s.kc - kwrap : s.kc)) * 2);
  };
  auto kpos_next = [&](KPos& s) {
    s.kc += 64;
    if (s.kc >= p.Cin) {
      s.kc = 0;
      s.dw += p.dil;
      if (++s.ks == p.S) { s.ks = 0; s.dw = 0; ++s.kr; s.dh += p.dil; }
    }
    s.uni = (unsigned)(((s.dh * p.W + s.dw) * ldi + (s.kc >= kwrap ? s.kc - kwrap : s.kc)) * 2);
  };
  KPos pa0, pa1;                       // K position of the next A0 / A1 half-tile to be issued
  kpos_init(pa0);
  kpos_init(pa1);
  int ta0 = 0, ta1 = 0, tb0 = 0, tb1 = 0;   // tile index (relative to kt0) of the next A0 / A1 / B0 / B1 issue

  unsigned char* const wbase = smem + wave * 1024;
  // `live` = the tile exists (tiles past the end of K are still "issued", with every lane out of range: the DMA
  // writes zeros into a dead slot and the vmcnt bookkeeping stays uniform).  Pure data flow, no branches.
  bool in_loop = false;
  auto issue_A1 = [&](int i, int u, int par, const KPos& s, bool live) {     // one 1-KiB piece per wave
    if (ABL == 2 && in_loop) return;
    const unsigned dead = live ? 0u : OOB;
    const int r = i * 2 + u;
    const int hi = a_hi0[r] + s.dh, wi = a_wi0[r] + s.dw;
    const bool ok = ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);
    const unsigned off = (ok ? a_off[r] + s.uni : OOB) | dead;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(wbase + slot_a(i, par) + u * 8192), 16, off, 0, 0, 0);
  };
  auto issue_B1 = [&](int j, int u, int par, int tile, bool live) {
    if (ABL == 2 && in_loop) return;
    const unsigned dead = live ? 0u : OOB;
    const unsigned kb = (unsigned)((kt0 + tile) * 128);
    const unsigned off = (b_off[j * 2 + u] + kb) | dead;         // b_off = OOB for rows past Cout: stays out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(wbase + slot_b(j, par) + u * 8192), 16, off, 0, 0, 0);
  };
  auto issue_A = [&](int i, int par, const KPos& s, bool live) {
#pragma unroll
    for (int u = 0; u < (i == 1 ? CA1 : 2); ++u) issue_A1(i, u, par, s, live);
  };
  auto issue_B = [&](int j, int par, int tile, bool live) {
#pragma unroll
    for (int u = 0; u < 2; ++u) issue_B1(j, u, par, tile, live);
  };

  // ---- fragment read addresses (bytes inside a half-tile slot): row (lane & 31) of the wave's piece, K step ks:
  //      logical chunk 2 ks + (lane >> 5), swizzled by the row
  const int l31 = lane & 31;
  unsigned a_rd[4], a1_rd[4], b_rd[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const unsigned ch = (unsigned)(((ks * 2 + (lane >> 5)) ^ ((l31 >> 1) & 7)) * 16);
    a_rd[ks] = (unsigned)((wr * 64 + l31) * ROWB) + ch;
    a1_rd[ks] = (unsigned)((wr * WROWS1 + l31) * ROWB) + ch;
    b_rd[ks] = (unsigned)(4 * HALF + (wc * 32 + l31) * ROWB) + ch;   // B slots start at 64 KiB: base in the register
  }

  f32x16_t acc[2][2][2];               // [A half i][M fragment f][B half j]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][f][j][r] = 0.f;

  // Fragment reads are inline asm: hipcc would otherwise protect every ds_read that follows an LDS-DMA with
  // `s_waitcnt vmcnt(0)` (it cannot tell the slots apart), which serialises the whole pipeline.  The compiler does
  // not know these are asynchronous, so each batch is followed (after the barrier) by an explicit lgkmcnt(0) and a
  // scheduling barrier before the first MFMA that consumes it.
  u32x4_t af[2][4], b0f[4], b1f[4];
  // ABL == 4: s_memtime stamps of block 0 at every segment boundary (timeline experiments, tools/gpu/ablate8.py)
  __shared__ unsigned long long tr_buf[ABL == 4 ? 8 * 96 : 1];
  int tr_n = 0;
#define MEGA_STAMP()                                                                                   \
  do {                                                                                                 \
    if (ABL == 4 && blockIdx.x == 0 && tr_n < 96) {                                                    \
      const unsigned long long tt = __builtin_amdgcn_s_memtime();                                      \
      if (lane == 0) tr_buf[wave * 96 + tr_n] = tt;                                                    \
      ++tr_n;                                                                                          \
    }                                                                                                  \
  } while (0)
#define MEGA_LDS_RD(dst, addr, imm)                                                                          \
  do {                                                                                                       \
    if (ABL != 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory"); \
  } while (0)
#define MEGA_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define MEGA_WAIT_LDS()                                   \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_sched_barrier(0);                    \
  } while (0)
#define MEGA_BAR()                    \
  do {                                \
    asm volatile("" ::: "memory");   \
    __builtin_amdgcn_s_barrier();     \
    asm volatile("" ::: "memory");   \
  } while (0)
// The WEIGHT fragment is the MFMA's first operand: the 32 x 32 result is then indexed [n][m] -- a lane owns ONE output row
// m = lane & 31 and, for each r >> 2, FOUR CONSECUTIVE channels n = 8 (r >> 2) + 4 (lane >> 5) + (r & 3) -- so the epilogue
// stages a fragment with 4 ds_write_b128 instead of 16 ds_write_b32 (its staging was LDS store-issue bound: 1.4 k cycles
// per slab).  Same products, same K order per output element: the same bits as the (activation, weight) operand order.
#define MEGA_MMA(acc_, a_, b_)                                                                                          \
  do {                                                                                                                  \
    if (ABL != 3)                                                                                                       \
      acc_ = Half16<HT>::mfma32(b_, a_, acc_);                                                                          \
    else                                                                                                                \
      asm volatile("" ::"v"(a_), "v"(b_));                                                                              \
  } while (0)

  // ---- prologue: six half-tiles in the steady-state issue order
  issue_A(0, 0, pa0, ta0 < nkt); kpos_next(pa0); ++ta0;
  issue_B(0, 0, tb0, tb0 < nkt); ++tb0;
  issue_B(1, 0, tb1, tb1 < nkt); ++tb1;
  issue_A(1, 0, pa1, ta1 < nkt); kpos_next(pa1); ++ta1;
  issue_A(0, 1, pa0, ta0 < nkt); kpos_next(pa0); ++ta0;
  issue_B(0, 1, tb0, tb0 < nkt); ++tb0;
  MEGA_WAIT_VM(4 + CA1);               // A0(0) B0(0) B1(0) have landed
  in_loop = true;
  if (ABL == 1) {                      // the ablation reads its fragments once
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(af[0][ks]) : "v"(a_rd[ks]) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(af[1][ks]) : "v"(a_rd[ks]) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(b0f[ks]) : "v"(b_rd[ks]) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:32768" : "=v"(b1f[ks]) : "v"(b_rd[ks]) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  MEGA_BAR();
  if (wr == 1) MEGA_BAR();             // group 1 runs one barrier behind group 0

  // Two phases per K-tile: (a) A0 x (B0, B1) = 16 MFMAs, (b) A1 x (B0, B1) = 8 MF1 MFMAs.  Timeline measurements
  // (s_memtime stamps, tools/gpu/ablate8.py): the waves never wait for DMA data; what stretches a K-tile beyond its
  // 2048 MFMA cycles is the ISSUE time of the load-side instructions -- an LDS-DMA costs ~110 cycles in a segment
  // with ds_reads and ~45 in the shadow of MFMAs, a fragment ds_read ~20 -- so the 8 DMA pieces of a K-tile are spread
  // over all four segments so that each interval's two concurrent segments (one group's MFMA cluster, the other's
  // load segment) are as even as the hazards allow.  Per wave and K-tile t (parity t & 1):
  //   L(a,t): 16 fragment reads, B1(t+1) x2          M(a,t): 16 MFMAs
  //   L(b,t):  8 fragment reads, A1(t+1) x CA1       M(b,t): 16 MFMAs with A0(t+2) x2, B0(t+2) x2 in their shadow
  //   WAR: a slot is refilled at the earliest from the MFMA segment of the phase AFTER its last read (both groups'
  //        reads are retired by their lgkmcnt(0) at least one barrier before any wave issues that DMA);
  //   RAW: L(a,t) waits for A1(t) [6 newer DMAs in flight], L(b,t) for A0 B0 B1 (t+1) [CA1 newer], each one barrier
  //        before the first wave reads them; every half-tile is issued >= 2 segments before the wait that retires it.
#define MEGA_SB() __builtin_amdgcn_sched_barrier(0)
  // SP kernels also serve narrow layers (Cout 64 / 128: layer1 / layer2 in the split-precision mode): a wave whose 32 columns
  // of a B half lie past Cout skips that half's MFMAs (wave-uniform branch; its accumulators stay zero and are never
  // stored) -- the all-zero columns of a 256-wide tile would otherwise cost 2-4x the layer's matrix-core time.
  const bool use_b0 = !SP || n0 + wc * 32 < p.Cout, use_b1 = !SP || n0 + 128 + wc * 32 < p.Cout;
#define MEGA_MMA_IF(c_, acc_, a_, b_) \
  do {                                 \
    if (!SP || (c_)) MEGA_MMA(acc_, a_, b_); \
  } while (0)
  auto tile_phases = [&](auto PAR) {
    constexpr int par = decltype(PAR)::value;
    // ================= phase a: A0 x (B0, B1)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      MEGA_LDS_RD(af[0][ks], a_rd[ks], slot_a(0, par));
      MEGA_LDS_RD(af[1][ks], a_rd[ks], slot_a(0, par) + 32 * ROWB);
      MEGA_LDS_RD(b0f[ks], b_rd[ks], slot_b(0, par) - 4 * HALF);
      MEGA_LDS_RD(b1f[ks], b_rd[ks], slot_b(1, par) - 4 * HALF);
    }
    issue_B1(1, 0, par ^ 1, tb1, tb1 < nkt); issue_B1(1, 1, par ^ 1, tb1, tb1 < nkt); ++tb1;
    MEGA_STAMP();
    MEGA_WAIT_VM(6);
    MEGA_STAMP();
    MEGA_BAR();
    MEGA_STAMP();
    MEGA_WAIT_LDS();
    MEGA_STAMP();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      MEGA_MMA_IF(use_b0, acc[0][0][0], af[0][ks], b0f[ks]); MEGA_MMA_IF(use_b0, acc[0][1][0], af[1][ks], b0f[ks]);
      MEGA_MMA_IF(use_b1, acc[0][0][1], af[0][ks], b1f[ks]); MEGA_MMA_IF(use_b1, acc[0][1][1], af[1][ks], b1f[ks]);
    }
    __builtin_amdgcn_s_setprio(0);
    MEGA_SB();
    MEGA_STAMP();
    MEGA_BAR();
    MEGA_STAMP();
    // ================= phase b: A1 x (B1, B0)  -- the B fragments are still in registers
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      MEGA_LDS_RD(af[0][ks], a1_rd[ks], slot_a(1, par));
      if (MF1 == 2) MEGA_LDS_RD(af[1][ks], a1_rd[ks], slot_a(1, par) + 32 * ROWB);
    }
    issue_A1(1, 0, par ^ 1, pa1, ta1 < nkt);
    if (CA1 == 2) issue_A1(1, 1, par ^ 1, pa1, ta1 < nkt);
    kpos_next(pa1); ++ta1;
    MEGA_STAMP();
    MEGA_WAIT_VM(CA1);
    MEGA_STAMP();
    MEGA_BAR();
    MEGA_STAMP();
    MEGA_WAIT_LDS();
    MEGA_STAMP();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      MEGA_MMA_IF(use_b1, acc[1][0][1], af[0][ks], b1f[ks]); if (MF1 == 2) MEGA_MMA_IF(use_b1, acc[1][1][1], af[1][ks], b1f[ks]);
      MEGA_MMA_IF(use_b0, acc[1][0][0], af[0][ks], b0f[ks]); if (MF1 == 2) MEGA_MMA_IF(use_b0, acc[1][1][0], af[1][ks], b0f[ks]);
      MEGA_SB();
      if (ks == 0) issue_A1(0, 0, par, pa0, ta0 < nkt);
      if (ks == 1) { issue_A1(0, 1, par, pa0, ta0 < nkt); kpos_next(pa0); ++ta0; issue_B1(0, 0, par, tb0, tb0 < nkt); }
      if (ks == 2) { issue_B1(0, 1, par, tb0, tb0 < nkt); ++tb0; }
      MEGA_SB();
    }
    __builtin_amdgcn_s_setprio(0);
    MEGA_SB();
    MEGA_STAMP();
    MEGA_BAR();
    MEGA_STAMP();
  };

  MEGA_TS(1);
  const int nkt2 = (nkt + 1) & ~1;     // an odd tail tile is computed on all-zero operands (its DMAs are out of range)
  for (int t = 0; t < nkt2; t += 2) {
    tile_phases(std::integral_constant<int, 0>{});
    tile_phases(std::integral_constant<int, 1>{});
  }
  if (ABL == 4 && blockIdx.x == 0 && p.partial) {
    __syncthreads();
    for (int e = tid; e < 8 * 96; e += NT8) reinterpret_cast<unsigned long long*>(p.partial)[e] = tr_buf[e];
  }
  if (wr == 0) MEGA_BAR();             // group 0 waits for group 1's last MFMA segment
  MEGA_WAIT_VM(0);                     // the out-of-range tail DMAs also write (zeros) into the LDS re-used below
  MEGA_BAR();
  MEGA_TS(2);

  // ---- epilogue.  Accumulator fragment [n][m] (see MEGA_MMA): lane owns output row m = lane & 31 of its fragment and
  //      channels 8 g + 4 (lane >> 5) + (0..3), g = 0..3.  The RAW accumulators are staged through LDS as f32, one 64-row
  //      slab per (A half, M fragment) -- the two wave rows' 32-row fragments, all 256 columns: 8 ds_write_b128 per lane --
  //      and read back as 16-byte vectors along n; FrozenBN scale / bias (one fma per element: the same v_fma the staging
  //      used to apply), residual add and activation happen at the read-out, where a thread's column vector is fixed
  //      (its 8 / 4 scale and bias values live in registers).  Same scheme as igemm.hip, same bits.
  constexpr int CST = BN + 4;
  static_assert(64 * CST * 4 <= LDS8, "staging slab must fit");
  float* cs = reinterpret_cast<float*>(smem);
  OT* __restrict__ out = (OT*)p.out;
  const HT* __restrict__ res = (const HT*)p.res;
  constexpr int OVE = 16 / (int)sizeof(OT);
  constexpr int VPR = BN / OVE;
  const float neg_slope = p.relu == 1 ? 0.f : (p.relu == 2 ? 0.1f : 1.f);
  auto act = [&](float x) { return x > 0.f ? x : x * neg_slope; };
  const bool vec_ok = (p.ldo % OVE == 0) && (!res || (SP ? p.ldr % 8 == 0 : (sizeof(OT) == 2 && p.ldr % OVE == 0)));
  auto stage = [&](float* csb, int i, int f) {
    float* dst = csb + (wr * 32 + l31) * CST + wc * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_t q = {acc[i][f][j][4 * g], acc[i][f][j][4 * g + 1], acc[i][f][j][4 * g + 2], acc[i][f][j][4 * g + 3]};
        *reinterpret_cast<f32x4_t*>(dst + j * 128 + 8 * g) = q;
      }
  };
  // The slab loop's barriers are raw s_barrier + lgkmcnt(0) (LDS traffic only), NOT __syncthreads(): a __syncthreads()
  // is also a vmcnt(0) fence, which made every slab wait for its own global stores to be acknowledged and for every
  // residual load in flight -- four exposed HBM round trips per 256-row tile (round 2), two with the residual rows
  // requested per slab pair (first half of round 3).  Now stores drain behind the next slab's staging and the residual
  // rows are a rolling two-slab prefetch: slab s+2's rows are requested into slab s's registers as soon as slab s is
  // written (the K loop's fragment registers are dead here; all four at once would need 64 registers and spill).  The
  // 1x1 "conv3 + residual" layers are HBM-bound: the epilogue is most of their time.
  constexpr int NIT = 64 * VPR / NT8;                  // 16-byte output vectors per thread per slab (4 bf16 / 8 f32)
  // ---- fast path (every bf16 / f32 layer whose Cout is a multiple of 256 and whose output is below 2 GiB: all of the
  //      frame stage): buffer loads / stores with the hardware range check instead of per-vector bounds branches.  With
  //      branches in the read-out loop hipcc merges the wait counts at every join into vmcnt(0) -- each 16-byte store
  //      then waited for the previous one to be acknowledged, four serialized round trips per slab.  Straight-line code
  //      keeps the counts exact: the stores of a slab go out back to back and drain behind the next slab's staging.
  //      SP kernels: any Cout % 8 == 0 (threads whose column vector lies past Cout carry an out-of-range offset), split
  //      residual / output planes; they have no other epilogue (launch8 refuses what does not qualify), except the raw
  //      partial sums of a split-K launch.
  const int oplanes = SP && p.split_out ? 2 : 1;       // planes of Cout columns behind each other in an output row
  // ABL == 6 (a product instantiation, not an ablation): the sub-pixel read-out of mega_conv2d_nhwc_subpixel -- GEMM row
  // (t, mh, mw), column (a, b, co) of a transposed conv's four phases goes to pixel (2 mh + a - crop, 2 mw + b - crop), channel
  // ps_coff + co of an NHWC tensor [N][ps_H][ps_W][ldo] (igemm_params.h); the launcher admits only what this path serves
  constexpr bool PS = ABL == 6;
  const size_t out_elems = PS ? (size_t)p.N * p.ps_H * p.ps_W * p.ldo : (size_t)(p.M - 1) * p.ldo + (size_t)oplanes * p.Cout;
  const bool fast = vec_ok && p.ksplit == 1 && (SP ? p.Cout % OVE == 0 : n0 + BN <= p.Cout) && (res == nullptr || sizeof(OT) == 2 || SP) &&
                    out_elems * sizeof(OT) < 0x7FF00000ull &&
                    (res == nullptr || ((size_t)(p.M - 1) * p.ldr + (size_t)(SP ? 2 : 1) * p.Cout) * 2 < 0x7FF00000ull);
  if (fast) {
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        p.out, 0, (int)(out_elems * sizeof(OT)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.res ? p.res : p.out), 0, res ? (int)(((size_t)(p.M - 1) * p.ldr + (size_t)(SP ? 2 : 1) * p.Cout) * 2) : 0, 0x00020000);
    constexpr int RSTEP = NT8 / VPR;                   // slab rows between a thread's consecutive vectors
    const int row0 = tid / VPR, ncol = n0 + (tid % VPR) * OVE;
    const unsigned cmask = SP && ncol >= p.Cout ? OOB : 0u;      // (SP) this thread's columns do not exist: loads give 0, stores are dropped
    const int ps_q = PS ? ncol / p.ps_C : 0;                     // (PS) this thread's phase (a, b) and channel inside it
    const int ps_col = PS ? p.ps_coff + (ncol - ps_q * p.ps_C) : 0;
    auto ps_off = [&](int m) -> unsigned {
      const int t = fast_div(m, p.mg_howo, p.sh_howo), rem = m - t * (p.Ho * p.Wo);
      const int mh = fast_div(rem, p.mg_wo, p.sh_wo), mw = rem - mh * p.Wo;
      const int y = 2 * mh + (ps_q >> 1) - p.ps_crop, x = 2 * mw + (ps_q & 1) - p.ps_crop;
      const bool ok = m < p.M && (unsigned)y < (unsigned)p.ps_H && (unsigned)x < (unsigned)p.ps_W;
      return ok ? (unsigned)(((t * p.ps_H + y) * p.ps_W + x) * p.ldo + ps_col) * (unsigned)sizeof(OT) : OOB;
    };
    // (row0 < RSTEP and RSTEP divides 32: the slab row row0 + it * RSTEP splits into a per-thread part and a
    // compile-time part -- one add per vector instead of the shift / mask / multiply chain)
    auto slab_m = [&](int i, int f, int it) {
      return (m0 + row0) + (i * 128 + ((it * RSTEP) >> 5) * (i == 1 ? WROWS1 : 64) + f * 32 + ((it * RSTEP) & 31));
    };
    float scv[OVE], biv[OVE];
    {
      const bool colok = !SP || ncol < p.Cout;
      // (16-byte loads when the vectors are 16-byte aligned -- every tensor torch allocates is; element loads otherwise:
      //  the C ABI does not promise an alignment for scale / bias)
      const bool al16 = ((reinterpret_cast<size_t>(p.scale) | reinterpret_cast<size_t>(p.bias)) & 15) == 0;
#pragma unroll
      for (int t = 0; t < OVE; t += 4) {
        f32x4_t s4 = {1.f, 1.f, 1.f, 1.f}, b4 = {0.f, 0.f, 0.f, 0.f};
        if (al16) {
          if (p.scale && colok) s4 = *reinterpret_cast<const f32x4_t*>(p.scale + ncol + t);
          if (p.bias && colok) b4 = *reinterpret_cast<const f32x4_t*>(p.bias + ncol + t);
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (p.scale && colok) s4[u] = p.scale[ncol + t + u];
            if (p.bias && colok) b4[u] = p.bias[ncol + t + u];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { scv[t + u] = s4[u]; biv[t + u] = b4[u]; }
      }
    }
    // RL = 1: ReLU applied to the ROUNDED value -- for bf16 one v_pk_max_i16 per pair on the packed bits (sign bit set
    // -> 0), for f32 one v_max: round(relu(x)) == relu(round(x)) bit for bit except that a negative input gives +0
    // instead of the generic form's -0 (x * 0).  The generic x > 0 ? x : x * slope costs cmp + cndmask + mul per element:
    // with it the read-out of a slab was bound by its ~106 vector-ALU instructions per thread, not by LDS or memory.
    // two staging slabs, used alternately: ONE barrier per slab.  Slab s + 2 overwrites slab s's buffer only after the
    // barrier that follows the staging of slab s + 1, and every wave reads slab s before it stages s + 1.  (The second
    // slab's offset is opaque to the compiler: folded into the ds offsets it would exceed their 16 bits and cost an
    // address register per store.)
    int slab1 = 64 * CST;
    asm volatile("" : "+v"(slab1));
    auto run = [&](auto HR, auto RL, auto SO) {
      constexpr bool HAS_RES = decltype(HR)::value;
      constexpr bool RELU = decltype(RL)::value;
      constexpr bool SPLIT_OUT = decltype(SO)::value;
      // plain kernels: rr[f][it] = the bf16 residual vectors of slab (i, f), two slabs rolling.
      // SP kernels: rr[plane][it] = the hi / lo vectors of ONE slab (the next slab's are requested right after this
      // slab's stores); an f32-output thread's vector is 4 elements: hi in words 0-1, lo in words 2-3 of rr[0][it].
      u32x4_t rr[2][NIT];
      auto ldres = [&](int i, int f, u32x4_t (&dst)[NIT]) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
          dst[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, (unsigned)(slab_m(i, f, it) * p.ldr + ncol) * 2u, 0, 0);
      };
      auto ldres_sp = [&](int i, int f) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const unsigned off = ((unsigned)(slab_m(i, f, it) * p.ldr + ncol) * 2u) | cmask;
          if constexpr (sizeof(OT) == 2) {
            rr[0][it] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, off, 0, 0);
            rr[1][it] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, off + (unsigned)p.Cout * 2u, 0, 0);
          } else {
            const u32x2_t h = __builtin_amdgcn_raw_buffer_load_b64(rs_res, off, 0, 0);
            const u32x2_t l = __builtin_amdgcn_raw_buffer_load_b64(rs_res, off + (unsigned)p.Cout * 2u, 0, 0);
            rr[0][it] = u32x4_t{h[0], h[1], l[0], l[1]};
          }
        }
      };
      if (HAS_RES) {
        if constexpr (SP) {
          ldres_sp(0, 0);
        } else {
          ldres(0, 0, rr[0]);
          ldres(0, 1, rr[1]);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int f = 0; f < (i == 1 ? MF1 : 2); ++f) {
          float* csb = ((2 * i + f) & 1) ? cs + slab1 : cs;
          stage(csb, i, f);
          MEGA_WAIT_LDS();
          MEGA_BAR();
          MEGA_TS(3 + 2 * (2 * i + f));
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const int row = row0 + it * RSTEP;
            float v[OVE];
#pragma unroll
            for (int t = 0; t < OVE; t += 4) {
              const float4 q4 = *reinterpret_cast<const float4*>(csb + row * CST + (tid % VPR) * OVE + t);
              v[t] = fmaf(q4.x, scv[t], biv[t]); v[t + 1] = fmaf(q4.y, scv[t + 1], biv[t + 1]);
              v[t + 2] = fmaf(q4.z, scv[t + 2], biv[t + 2]); v[t + 3] = fmaf(q4.w, scv[t + 3], biv[t + 3]);
            }
            if constexpr (HAS_RES && !SP) {
              u32x4_t r4 = rr[f][it];
              asm volatile("" : "+v"(r4));             // unpack HERE: hoisted, the 64 unpacked floats of two slabs spill
#pragma unroll
              for (int d = 0; d < 4; ++d) {
                v[2 * d] += Half16<HT>::lo(r4[d]);
                v[2 * d + 1] += Half16<HT>::hi(r4[d]);
              }
            }
            if constexpr (HAS_RES && SP) {             // residual value = hi + lo (exact in f32), then one add
              u32x4_t h4 = rr[0][it];
              asm volatile("" : "+v"(h4));
              if constexpr (sizeof(OT) == 2) {
                u32x4_t l4 = rr[1][it];
                asm volatile("" : "+v"(l4));
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                  v[2 * d] += Half16<HT>::lo(h4[d]) + Half16<HT>::lo(l4[d]);
                  v[2 * d + 1] += Half16<HT>::hi(h4[d]) + Half16<HT>::hi(l4[d]);
                }
              } else {
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                  v[2 * d] += Half16<HT>::lo(h4[d]) + Half16<HT>::lo(h4[2 + d]);
                  v[2 * d + 1] += Half16<HT>::hi(h4[d]) + Half16<HT>::hi(h4[2 + d]);
                }
              }
            }
            const unsigned ooff = PS ? ps_off(slab_m(i, f, it)) : (((unsigned)(slab_m(i, f, it) * p.ldo + ncol) * (unsigned)sizeof(OT)) | cmask);
            u32x4_t o;                                 // packed explicitly (no type-punned stores into o)
            if constexpr (SPLIT_OUT) {                 // hi = bf16(x), lo = bf16(x - hi) of x = act(v), both planes
              u32x4_t ol;
#pragma unroll
              for (int d = 0; d < 4; ++d) {
                const float x0 = RELU ? fmaxf(v[2 * d], 0.f) : act(v[2 * d]), x1 = RELU ? fmaxf(v[2 * d + 1], 0.f) : act(v[2 * d + 1]);
                const unsigned h = Half16<HT>::pack2(x0, x1);
                o[d] = h;
                ol[d] = Half16<HT>::pack2(x0 - Half16<HT>::lo(h), x1 - Half16<HT>::hi(h));
              }
              __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, ooff, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b128(ol, rs_out, ooff + (unsigned)p.Cout * 2u, 0, 0);
            } else {
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
          if constexpr (SP) {
            // the NEXT slab's residual rows, into the registers this slab's read-out just released
            if (HAS_RES && !(i == 1 && f == MF1 - 1)) ldres_sp(f + 1 < (i == 1 ? MF1 : 2) ? i : i + 1, f + 1 < (i == 1 ? MF1 : 2) ? f + 1 : 0);
          } else {
            if (HAS_RES && i == 0 && f < MF1) ldres(1, f, rr[f]);   // slab s + 2 into the registers slab s just freed
          }
          MEGA_TS(4 + 2 * (2 * i + f));
        }
      }
    };
    const bool so = SP && sizeof(OT) == 2 && p.split_out;
    if constexpr (SP && sizeof(OT) == 2) {
      if (so) {
        if (res) {
          if (p.relu == 1) run(std::true_type{}, std::true_type{}, std::true_type{}); else run(std::true_type{}, std::false_type{}, std::true_type{});
        } else {
          if (p.relu == 1) run(std::false_type{}, std::true_type{}, std::true_type{}); else run(std::false_type{}, std::false_type{}, std::true_type{});
        }
      }
    }
    if (!so) {
      if (res) {
        if (p.relu == 1) run(std::true_type{}, std::true_type{}, std::false_type{}); else run(std::true_type{}, std::false_type{}, std::false_type{});
      } else {
        if (p.relu == 1) run(std::false_type{}, std::true_type{}, std::false_type{}); else run(std::false_type{}, std::false_type{}, std::false_type{});
      }
    }
    if (ABL == 5) {
      MEGA_WAIT_VM(0);
      MEGA_TS(11);
      MEGA_TS_RT(14);
      if ((wave & 3) == 0 && lane == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        reinterpret_cast<unsigned long long*>(p.partial)[((size_t)blockIdx.x * 2 + (wave >> 2)) * 16 + 12] =
            (unsigned long long)hwid | ((unsigned long long)xcc << 32);
      }
    }
    return;
  }
  // ---- general path: partial sums of a split-K launch, Cout not a multiple of 256, unaligned rows, outputs >= 2 GiB
  //      (SP kernels get here for split-K partial sums only)
  if (SP && p.ksplit == 1) return;                     // (launch8 refuses such launches: nothing to do here)
  const bool res_vec = res != nullptr && vec_ok && sizeof(OT) == 2 && p.ksplit == 1;
  uint4 rres2[2][NIT];
  auto load_res = [&](int i, int f, uint4 (&dst)[NIT]) {
    const int wrows = i == 1 ? WROWS1 : 64;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + it * NT8;
      const int row = e / VPR, cv = e - row * VPR;
      const int m = m0 + i * 128 + (row >> 5) * wrows + f * 32 + (row & 31), n = n0 + cv * OVE;
      dst[it] = make_uint4(0, 0, 0, 0);
      if (m < p.M && n + OVE <= p.Cout) dst[it] = *reinterpret_cast<const uint4*>(res + (size_t)m * p.ldr + n);
    }
  };
  auto scb = [&](int n, float x) {                       // FrozenBN scale / bias of column n (split-K: finalize applies them)
    const float sc = (p.ksplit == 1 && p.scale && n < p.Cout) ? p.scale[n] : 1.f;
    const float bi = (p.ksplit == 1 && p.bias && n < p.Cout) ? p.bias[n] : 0.f;
    return fmaf(x, sc, bi);
  };
  if (res_vec) {
    load_res(0, 0, rres2[0]);
    load_res(0, 1, rres2[1]);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int f = 0; f < (i == 1 ? MF1 : 2); ++f) {
      const int wrows = i == 1 ? WROWS1 : 64;          // rows per wave row inside this A half
      uint4 (&rres)[NIT] = rres2[f];
      if (i + f > 0) { MEGA_WAIT_LDS(); MEGA_BAR(); }  // the previous slab has been read out
      stage(cs, i, f);
      MEGA_WAIT_LDS();
      MEGA_BAR();
      if (p.ksplit > 1) {                              // raw partial sums; splitk_finalize_kernel (igemm.hip) finishes
        float* part = p.partial + (size_t)blockIdx.z * p.M * p.Cout;
        for (int e = tid; e < 64 * (BN / 4); e += NT8) {
          const int row = e / (BN / 4), cv = e - row * (BN / 4);
          const int m = m0 + i * 128 + (row >> 5) * wrows + f * 32 + (row & 31), n = n0 + cv * 4;
          if (m >= p.M || n >= p.Cout) continue;
          const float4 v = *reinterpret_cast<const float4*>(cs + row * CST + cv * 4);
          if (n + 4 <= p.Cout && p.Cout % 4 == 0) {
            *reinterpret_cast<float4*>(part + (size_t)m * p.Cout + n) = v;
          } else {
            const float vv[4] = {v.x, v.y, v.z, v.w};
            for (int t = 0; t < 4 && n + t < p.Cout; ++t) part[(size_t)m * p.Cout + n + t] = vv[t];
          }
        }
        continue;
      }
      if constexpr (!SP) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int e = tid + it * NT8;
        const int row = e / VPR, cv = e - row * VPR;
        const int m = m0 + i * 128 + (row >> 5) * wrows + f * 32 + (row & 31), n = n0 + cv * OVE;
        if (m >= p.M || n >= p.Cout) continue;
        float v[OVE];
#pragma unroll
        for (int t = 0; t < OVE; t += 4) {
          const float4 q4 = *reinterpret_cast<const float4*>(cs + row * CST + cv * OVE + t);
          v[t] = scb(n + t, q4.x); v[t + 1] = scb(n + t + 1, q4.y); v[t + 2] = scb(n + t + 2, q4.z); v[t + 3] = scb(n + t + 3, q4.w);
        }
        if (vec_ok && n + OVE <= p.Cout) {
          if (res_vec) {
            const HT* re = reinterpret_cast<const HT*>(&rres[it]);
#pragma unroll
            for (int t = 0; t < OVE; ++t) v[t] += Elem<HT>::ld(re + t);
          } else if (res) {
            for (int t = 0; t < OVE; ++t) v[t] += Elem<HT>::ld(res + (size_t)m * p.ldr + n + t);
          }
          uint4 o;
          OT* oe = reinterpret_cast<OT*>(&o);
#pragma unroll
          for (int t = 0; t < OVE; ++t) Elem<OT>::st(oe + t, act(v[t]));
          *reinterpret_cast<uint4*>(out + (size_t)m * p.ldo + n) = o;
        } else {
          for (int t = 0; t < OVE && n + t < p.Cout; ++t) {
            float x = v[t];
            if (res) x += Elem<HT>::ld(res + (size_t)m * p.ldr + n + t);
            Elem<OT>::st(out + (size_t)m * p.ldo + n + t, act(x));
          }
        }
      }
      if (res_vec && i == 0 && f < MF1) load_res(1, f, rres);   // slab s + 2 into the registers slab s just freed
      }
    }
  }
#undef MEGA_LDS_RD
#undef MEGA_WAIT_VM
#undef MEGA_WAIT_LDS
#undef MEGA_BAR
#undef MEGA_MMA
#undef MEGA_SB
#undef MEGA_MMA_IF
#undef MEGA_STAMP
}

// n / d == (umulhi(n, mg) + n) >> sh for every 0 <= n < 2^31 (round-up method: sh = ceil(log2 d), mg = floor(2^32 (2^sh - d) / d) + 1)
inline void magic_div(int d, unsigned& mg, unsigned& sh) {
  sh = 0;
  while ((1ull << sh) < (unsigned long long)d) ++sh;
  mg = (unsigned)((((1ull << sh) - (unsigned long long)d) << 32) / (unsigned long long)d + 1ull);
}

template <typename OT, int MF1, int CLS = 0, int ABL = 0, int SP = 0, typename HT = bf16_t>
int launch8(const ConvParams& p0, hipStream_t st) {
  constexpr int BM = 128 + 64 * MF1;
  ConvParams p = p0;
  magic_div(p.Ho * p.Wo, p.mg_howo, p.sh_howo);
  magic_div(p.Wo, p.mg_wo, p.sh_wo);
  const int ntm = cdiv(p.M, BM), ntn = cdiv(p.Cout, 256);
  // set on every launch (a per-process flag would miss the second device of a multi-GPU process)
  (void)hipFuncSetAttribute((const void*)igemm8_kernel<OT, MF1, CLS, ABL, SP, HT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS8_ALLOC);
  hipLaunchKernelGGL((igemm8_kernel<OT, MF1, CLS, ABL, SP, HT>), dim3(ntm * ntn, 1, p.ksplit), dim3(NT8), LDS8_ALLOC, st, p);
  return mega_check_launch();
}

}  // namespace

// the streaming launch class of igemm8_kernel's CLS parameter (see there)
int mega_igemm8_streaming(int taps, int K) { return taps == 1 && K <= 512; }

int mega_igemm8_supports(const ConvParams& p) {
  // capability (a single K-tile works: the ring's second tile is then all zeros); which shapes are SENT here by default
  // is choose_tile's policy (igemm.hip)
  return p.Cin % 64 == 0 && p.in_bytes < 0x7FF00000u && p.w_bytes < 0x7FF00000u && (p.K >> 6) >= 1;
}

int mega_igemm8_launch(const ConvParams& p, int bm, int out_f32, int half_dtype, hipStream_t st) {
  if (half_dtype == MEGA_F16 && p.sp) {  // fp16 [hi | lo] planes (conv_mode "h2"): the SP kernels on IEEE-half pairs
    const size_t oplanes = p.split_out ? 2 : 1, osz = out_f32 ? 4 : 2;
    const bool ok = p.ldi >= (p.kwrap > 0 ? p.kwrap : p.Cin) && p.ldi % 64 == 0 && (p.kwrap == 0 || (p.kwrap % 64 == 0 && p.kwrap < p.Cin && p.Cin - p.kwrap <= p.kwrap)) &&
                    p.Cout % 8 == 0 && !(p.split_out && out_f32) && p.ldo % (out_f32 ? 4 : 8) == 0 && p.ldo >= (int)oplanes * p.Cout &&
                    (!p.res || (p.ldr % 8 == 0 && p.ldr >= 2 * p.Cout)) &&
                    ((size_t)(p.M - 1) * p.ldo + oplanes * p.Cout) * osz < 0x7FF00000ull &&
                    (!p.res || ((size_t)(p.M - 1) * p.ldr + 2 * (size_t)p.Cout) * 2 < 0x7FF00000ull) &&
                    (p.ksplit == 1 || (!p.split_out && !p.res));
    if (!ok) return MEGA_ERR_ARG;
    const bool streamf = mega_igemm8_streaming(p.R * p.S, p.K) && p.ksplit == 1;
    if (bm == 256 && streamf) return out_f32 ? launch8<float, 2, 1, 0, 1, f16_t>(p, st) : launch8<f16_t, 2, 1, 0, 1, f16_t>(p, st);
    if (bm == 192 && streamf) return out_f32 ? launch8<float, 1, 1, 0, 1, f16_t>(p, st) : launch8<f16_t, 1, 1, 0, 1, f16_t>(p, st);
    if (bm == 256) return out_f32 ? launch8<float, 2, 0, 0, 1, f16_t>(p, st) : launch8<f16_t, 2, 0, 0, 1, f16_t>(p, st);
    if (bm == 192) return out_f32 ? launch8<float, 1, 0, 0, 1, f16_t>(p, st) : launch8<f16_t, 1, 0, 0, 1, f16_t>(p, st);
    return MEGA_ERR_ARG;
  }
  if (p.ps) {                          // sub-pixel read-out (mega_conv2d_nhwc_subpixel): 16-bit in / out, no residual, no split-K
    if (p.sp || out_f32 || p.res || p.ksplit != 1 || p.Cout % 256 != 0 || p.ps_C % 8 != 0 || p.ldo % 8 != 0 || p.ps_coff % 8 != 0 ||
        (size_t)p.N * p.ps_H * p.ps_W * p.ldo * 2 >= 0x7FF00000ull)
      return MEGA_ERR_ARG;
    if (half_dtype == MEGA_F16) return bm == 192 ? launch8<f16_t, 1, 0, 6, 0, f16_t>(p, st) : launch8<f16_t, 2, 0, 6, 0, f16_t>(p, st);
    if (half_dtype == MEGA_BF16) return bm == 192 ? launch8<bf16_t, 1, 0, 6>(p, st) : launch8<bf16_t, 2, 0, 6>(p, st);
    return MEGA_ERR_ARG;
  }
  if (half_dtype == MEGA_F16) {        // IEEE half operands (same tiles, same launch classes)
    const bool streamf = mega_igemm8_streaming(p.R * p.S, p.K) && p.ksplit == 1;
    if (bm == 256 && streamf) return out_f32 ? launch8<float, 2, 1, 0, 0, f16_t>(p, st) : launch8<f16_t, 2, 1, 0, 0, f16_t>(p, st);
    if (bm == 192 && streamf) return out_f32 ? launch8<float, 1, 1, 0, 0, f16_t>(p, st) : launch8<f16_t, 1, 1, 0, 0, f16_t>(p, st);
    if (bm == 256) return out_f32 ? launch8<float, 2, 0, 0, 0, f16_t>(p, st) : launch8<f16_t, 2, 0, 0, 0, f16_t>(p, st);
    if (bm == 192) return out_f32 ? launch8<float, 1, 0, 0, 0, f16_t>(p, st) : launch8<f16_t, 1, 0, 0, 0, f16_t>(p, st);
    return MEGA_ERR_ARG;
  }
  if (half_dtype != MEGA_BF16) return MEGA_ERR_ARG;
#ifdef MEGA_EXPERIMENTS
  // Timing ablations / s_memtime timeline (tools/gpu/ablate8.py).  NOT part of the product library: this block
  // allocates, synchronises and prints, and its kernels return garbage by construction -- it only exists in a library
  // built with MEGA_BUILD_EXPERIMENTS=1 (mega/pytorch_amd/build.py adds -DMEGA_EXPERIMENTS).
  static const int abl = getenv("MEGA_IGEMM8_ABLATE") ? atoi(getenv("MEGA_IGEMM8_ABLATE")) : 0;
  if (abl && bm == 256 && !out_f32) {
    if (abl == 1) return launch8<bf16_t, 2, 0, 1>(p, st);
    if (abl == 2) return launch8<bf16_t, 2, 0, 2>(p, st);
    if (abl == 3) return launch8<bf16_t, 2, 0, 3>(p, st);
    if (abl == 5) {          // whole-tile timeline of every block -> MEGA_IGEMM8_TIMELINE_OUT (raw u64 [grid][2][16])
      static unsigned long long* d_tr = nullptr;
      static size_t cap = 0;
      const size_t nblk = (size_t)cdiv(p.M, 256) * cdiv(p.Cout, 256), bytes = nblk * 2 * 16 * sizeof(unsigned long long);
      if (bytes > cap) { if (d_tr) (void)hipFree(d_tr); (void)hipMalloc(&d_tr, bytes); cap = bytes; }
      (void)hipMemsetAsync(d_tr, 0, bytes, st);
      ConvParams q = p;
      q.partial = reinterpret_cast<float*>(d_tr);
      const int rc = launch8<bf16_t, 2, 0, 5>(q, st);
      (void)hipStreamSynchronize(st);
      const char* path = getenv("MEGA_IGEMM8_TIMELINE_OUT");
      if (path) {
        unsigned long long* h = (unsigned long long*)malloc(bytes);
        (void)hipMemcpy(h, d_tr, bytes, hipMemcpyDeviceToHost);
        FILE* f = fopen(path, "wb");
        if (f) { fwrite(h, 1, bytes, f); fclose(f); }
        free(h);
      }
      return rc;
    }
    if (abl == 4) {          // timeline of block 0: 12 stamps per K-tile per wave, first 8 K-tiles
      static unsigned long long* d_tr = nullptr;
      if (!d_tr) (void)hipMalloc(&d_tr, 8 * 96 * sizeof(unsigned long long));
      ConvParams q = p;
      q.partial = reinterpret_cast<float*>(d_tr);
      const int rc = launch8<bf16_t, 2, 0, 4>(q, st);
      (void)hipStreamSynchronize(st);
      static unsigned long long h[8 * 96];
      (void)hipMemcpy(h, d_tr, sizeof(h), hipMemcpyDeviceToHost);
      static int printed = 0;
      if (printed++ == 2) {    // third launch: warm
        const char* names[12] = {"a:reads", "a:vmcnt", "a:bar", "a:lgkm", "a:mfma", "a:bar2", "b:reads", "b:vmcnt", "b:bar", "b:lgkm", "b:mfma", "b:bar2"};
        for (int w = 0; w < 8; w += 4) {
          printf("wave %d (group %d): stamp deltas (100 MHz ticks x? -> raw s_memtime units) over K-tiles 2..5\n", w, w >> 2);
          for (int t = 2; t < 6; ++t) {
            printf("  tile %d:", t);
            for (int e = 0; e < 12; ++e) printf(" %s %llu", names[e], h[w * 96 + t * 12 + e] - h[w * 96 + t * 12 + e - 1]);
            printf("  | tile total %llu\n", h[w * 96 + t * 12 + 11] - h[w * 96 + (t - 1) * 12 + 11]);
          }
        }
        fflush(stdout);
      }
      return rc;
    }
  }
#endif
  const bool stream = mega_igemm8_streaming(p.R * p.S, p.K) && p.ksplit == 1;
  if (p.sp) {
    // split-precision planes: the SP kernels have the buffer-addressed epilogue only (+ raw split-K partial sums)
    const size_t oplanes = p.split_out ? 2 : 1, osz = out_f32 ? 4 : 2;
    const bool ok = p.ldi >= (p.kwrap > 0 ? p.kwrap : p.Cin) && p.ldi % 64 == 0 && (p.kwrap == 0 || (p.kwrap % 64 == 0 && p.kwrap < p.Cin && p.Cin - p.kwrap <= p.kwrap)) &&
                    p.Cout % 8 == 0 && !(p.split_out && out_f32) && p.ldo % (out_f32 ? 4 : 8) == 0 && p.ldo >= (int)oplanes * p.Cout &&
                    (!p.res || (p.ldr % 8 == 0 && p.ldr >= 2 * p.Cout)) &&
                    ((size_t)(p.M - 1) * p.ldo + oplanes * p.Cout) * osz < 0x7FF00000ull &&
                    (!p.res || ((size_t)(p.M - 1) * p.ldr + 2 * (size_t)p.Cout) * 2 < 0x7FF00000ull) &&
                    (p.ksplit == 1 || (!p.split_out && !p.res));
    if (!ok) return MEGA_ERR_ARG;
    if (bm == 256 && stream) return out_f32 ? launch8<float, 2, 1, 0, 1>(p, st) : launch8<bf16_t, 2, 1, 0, 1>(p, st);
    if (bm == 192 && stream) return out_f32 ? launch8<float, 1, 1, 0, 1>(p, st) : launch8<bf16_t, 1, 1, 0, 1>(p, st);
    if (bm == 256) return out_f32 ? launch8<float, 2, 0, 0, 1>(p, st) : launch8<bf16_t, 2, 0, 0, 1>(p, st);
    if (bm == 192) return out_f32 ? launch8<float, 1, 0, 0, 1>(p, st) : launch8<bf16_t, 1, 0, 0, 1>(p, st);
    return MEGA_ERR_ARG;
  }
  if (bm == 256 && stream) return out_f32 ? launch8<float, 2, 1>(p, st) : launch8<bf16_t, 2, 1>(p, st);
  if (bm == 192 && stream) return out_f32 ? launch8<float, 1, 1>(p, st) : launch8<bf16_t, 1, 1>(p, st);
  if (bm == 256) return out_f32 ? launch8<float, 2>(p, st) : launch8<bf16_t, 2>(p, st);
  if (bm == 192) return out_f32 ? launch8<float, 1>(p, st) : launch8<bf16_t, 1>(p, st);
  return MEGA_ERR_ARG;
}
