// stream1x1: the HBM-bound 1x1 convolutions of the MEGA frame stage with a short contraction -- layer3's conv3 (256 -> 1024,
// + residual + ReLU: 23 launches per frame batch), layer2's conv3 (128 -> 512) -- as a PERSISTENT kernel whose waves run whole
// output tiles on their own (mega_core/modeling/backbone/resnet.py:324-344: conv3 + bn3 + `out += identity` + relu_).
//
// Why a third GEMM kernel.  These layers move 4-9 bytes per FLOP-pair less than the matrix cores need: they are priced against
// HBM (441 MB per 40-frame launch of layer3's conv3; torch's element-wise `add` over the same three tensors runs at 5.8 TB/s,
// tools/probes/stream_bw.py).  The tile kernels reach 3.7-3.8 TB/s on them whether one block owns the CU (igemm8) or two share
// it (igemm2): a tile is a CHAIN of dependent memory phases -- first operand tile from HBM, K-tiles behind a 2-deep ring (each
// a new HBM round trip for the activations), residual rows, stores -- and a CU runs one or two such chains at a time.
// Here the chain is one HBM round trip long and a CU runs eight of them:
//   * the block's weight tile [256 columns][K] is loaded into LDS ONCE (128 KiB at K = 256) and stays for the block's life;
//   * a WAVE owns a 32-row x 256-column output tile: its A operand (32 rows x K: one contiguous 16 KiB of the NHWC map) goes
//     straight from global memory into registers in MFMA fragment layout -- all K / 16 loads in flight at once --, the weight
//     fragments come from LDS (no L2 latency in the K loop), the residual's 8-byte pieces are requested during the K loop into
//     the registers the consumed A fragments leave behind, and the epilogue runs from the accumulator registers (FrozenBN
//     scale / bias from LDS, + residual, activation, 8-byte stores: a lane owns 4 consecutive channels of one row);
//   * no barrier after the weight load: the 8 waves of a CU are 8 independent streams (32 KiB of loads in flight each).
// Same MFMA (v_mfma_f32_32x32x16_bf16 / _f16, weight fragment first), same ascending K order and the same epilogue arithmetic
// as igemm8.hip: bit-identical outputs (tools/gpu/stream1x1_check.py, tests/test_kernels_gpu.py).
//
// Work split: block b runs on XCD b & 7 (round-robin dispatch); an XCD owns one eighth of the M range and ALL N tiles of it, so
// the Cout / 256 readers of an activation row share it in that XCD's L2; inside (XCD, N tile) the 32-row tiles are dealt to the
// group's waves round-robin.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "igemm_params.h"

namespace {

typedef unsigned int u32x2s_t __attribute__((ext_vector_type(2)));

constexpr int NTS = 512;

template <typename HT, int K, bool HAS_RES, bool RELU>
__global__ __launch_bounds__(NTS, 2) void stream1x1_kernel(ConvParams p) {
  constexpr int NKS = K / 16;                        // 16-deep K-steps
  constexpr int RB = K * 2;                          // bytes per weight row in LDS
  constexpr int CPR = K / 8;                         // 16-byte chunks per row (16 or 32: the swizzle XORs the low 4 bits)
  constexpr int RPK = 32 / NKS;                      // residual pieces requested per K-step (32 per tile)
  static_assert(K == 128 || K == 256, "weight tile must fit LDS and a row must hold >= 16 chunks");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sb = reinterpret_cast<float*>(smem + 256 * RB);      // [2][256] scale | bias of this N tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;

  const int ntn = p.Cout >> 8;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int tile_n = loc % ntn, sub = loc / ntn, nsub = per / ntn;
  const int n0 = tile_n * 256;

  // ---- the weight tile, once: row n's logical chunk c at physical chunk c ^ (n & 15): the 16 lanes of a ds_read_b128 group
  //      (16 consecutive columns, one logical chunk) touch 16 different bank quads
  {
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.w) + (size_t)n0 * RB;
    for (int e = tid; e < 256 * CPR; e += NTS) {
      const int n = e / CPR, c = e % CPR;
      const uint4 v = *reinterpret_cast<const uint4*>(wsrc + (size_t)n * RB + c * 16);
      *reinterpret_cast<uint4*>(smem + n * RB + ((c ^ (n & 15)) * 16)) = v;
    }
    for (int e = tid; e < 256; e += NTS) {
      sb[e] = p.scale ? p.scale[n0 + e] : 1.f;
      sb[256 + e] = p.bias ? p.bias[n0 + e] : 0.f;
    }
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
      p.out, 0, (int)(((size_t)(p.M - 1) * p.ldo + p.Cout) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(HAS_RES ? p.res : p.out), 0, HAS_RES ? (int)(((size_t)(p.M - 1) * p.ldr + p.Cout) * 2) : 0, 0x00020000);
  const float neg_slope = p.relu == 1 ? 0.f : (p.relu == 2 ? 0.1f : 1.f);
  auto act = [&](float x) { return x > 0.f ? x : x * neg_slope; };

  // fragment read address of column (32 j + l31), K-step ks: logical chunk 2 ks + h, swizzled by (n & 15) = (l31 & 15):
  // ((2 ks + h) ^ s) * 16 = ((h ^ s) * 16) ^ (32 ks)  (2 ks and h share no bit)
  const unsigned brd = (unsigned)(l31 * RB) + (unsigned)((h ^ (l31 & 15)) * 16);
  const float* sbl = sb + 4 * h;

  const int T = (p.M + 31) >> 5;
  const int t_lo = (int)((long long)T * xcd >> 3), t_hi = (int)((long long)T * (xcd + 1) >> 3);
  const int wid = sub * 8 + wave, nw = nsub * 8;
  for (int t = t_lo + wid; t < t_hi; t += nw) {
    const int m = t * 32 + l31;
    // ---- the A operand: row m, K-step ks = 16 bytes at column byte 32 ks + 16 h (rows past M: beyond num_records -> zeros)
    const unsigned aoff = (unsigned)m * (unsigned)RB + (unsigned)(h * 16);
    u32x4_t a[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) a[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, aoff + ks * 32, 0, 0);
    const unsigned roff = ((unsigned)m * (unsigned)p.ldr + (unsigned)(n0 + 4 * h)) * 2u;
    const unsigned ooff = ((unsigned)m * (unsigned)p.ldo + (unsigned)(n0 + 4 * h)) * 2u;
    u32x2s_t rr[HAS_RES ? 32 : 1];

    f32x16_t acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      u32x4_t b[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) b[j] = *reinterpret_cast<const u32x4_t*>(smem + ((brd ^ (unsigned)(32 * ks)) + (unsigned)(j * 32 * RB)));
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = Half16<HT>::mfma32(b[j], a[ks], acc[j]);
      if constexpr (HAS_RES) {
        // the residual's pieces (j, g) = 4 channels 32 j + 8 g + 4 h .. of row m, in epilogue order, into the registers the
        // consumed A fragments leave behind
#pragma unroll
        for (int u = 0; u < RPK; ++u) {
          const int q = ks * RPK + u;
          rr[q] = __builtin_amdgcn_raw_buffer_load_b64(rs_res, roff + (unsigned)((q >> 2) * 64 + (q & 3) * 16), 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);             // (K-steps stay in order: hoisted fragment reads of later steps spill)
    }

    // ---- epilogue from the accumulators: acc[j][4 g + i] = row m, channel n0 + 32 j + 8 g + 4 h + i
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_t s4 = *reinterpret_cast<const f32x4_t*>(sbl + 32 * j + 8 * g);
        const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(sbl + 256 + 32 * j + 8 * g);
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaf(acc[j][4 * g + i], s4[i], b4[i]);
        if constexpr (HAS_RES) {
          u32x2s_t r2 = rr[j * 4 + g];
          v[0] += Half16<HT>::lo(r2[0]); v[1] += Half16<HT>::hi(r2[0]);
          v[2] += Half16<HT>::lo(r2[1]); v[3] += Half16<HT>::hi(r2[1]);
        }
        u32x2s_t o;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          if constexpr (RELU) {
            const s16x2_t z = {0, 0};
            o[d] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, Half16<HT>::pack2(v[2 * d], v[2 * d + 1])), z));
          } else {
            o[d] = Half16<HT>::pack2(act(v[2 * d]), act(v[2 * d + 1]));
          }
        }
        __builtin_amdgcn_raw_buffer_store_b64(o, rs_out, ooff + (unsigned)(j * 64 + g * 16), 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <typename HT, int K>
int launch_s(const ConvParams& p, hipStream_t st) {
  const int lds = 256 * K * 2 + 2 * 256 * 4;
  const int grid = 256;                              // one block per CU of an MI355X (8 XCDs x 32 CUs), persistent
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTS), lds, st, p);
    return mega_check_launch();
  };
  if (p.res) return p.relu == 1 ? go(stream1x1_kernel<HT, K, true, true>) : go(stream1x1_kernel<HT, K, true, false>);
  return p.relu == 1 ? go(stream1x1_kernel<HT, K, false, true>) : go(stream1x1_kernel<HT, K, false, false>);
}

}  // namespace

// 1 when stream1x1 takes this launch: a plain 16-bit 1x1 / stride-1 / unpadded conv (the input map IS the [M][K] operand) with
// K = 128 or 256, Cout a multiple of 256 whose N-tile count divides an XCD's 32 blocks, a 16-bit output, rows of whole 8-byte
// pieces, tensors below 2 GiB, and enough rows to give every wave of the chip a tile
int mega_stream1x1_supports(const ConvParams& p, int out_f32) {
  if (p.sp || p.ksplit != 1 || out_f32) return 0;
  if (p.R != 1 || p.S != 1 || p.stride != 1 || p.pad != 0 || p.K != p.Cin) return 0;
  if (p.K != 128 && p.K != 256) return 0;
  if (p.Cout % 256 != 0 || 32 % (p.Cout / 256) != 0) return 0;
  if (p.M < 16384) return 0;
  if (p.ldo % 4 != 0 || (p.res && p.ldr % 4 != 0)) return 0;
  if (p.in_bytes >= 0x7FF00000u || (size_t)p.M * p.K * 2 > (size_t)p.in_bytes) return 0;
  if (((size_t)(p.M - 1) * p.ldo + p.Cout) * 2 >= 0x7FF00000ull) return 0;
  if (p.res && ((size_t)(p.M - 1) * p.ldr + p.Cout) * 2 >= 0x7FF00000ull) return 0;
  if ((reinterpret_cast<size_t>(p.w) | reinterpret_cast<size_t>(p.in)) & 15) return 0;
  return 1;
}

int mega_stream1x1_launch(const ConvParams& p, int half_dtype, hipStream_t st) {
  if (half_dtype == MEGA_F16) return p.K == 256 ? launch_s<f16_t, 256>(p, st) : launch_s<f16_t, 128>(p, st);
  if (half_dtype == MEGA_BF16) return p.K == 256 ? launch_s<bf16_t, 256>(p, st) : launch_s<bf16_t, 128>(p, st);
  return MEGA_ERR_ARG;
}
