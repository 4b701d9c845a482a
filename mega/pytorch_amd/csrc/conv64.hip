// 3x3, 64 -> 64 channel convolution + FrozenBN + ReLU for the full-resolution stage of the backbone (layer1's conv2,
// mega_core/modeling/backbone/resnet.py:324-344 with in = out = 64 bottleneck channels; 3 launches per frame batch on
// 150 x 250 maps).  As an implicit GEMM this layer has K = 576 and N = 64: 9 K-tiles, a 64-wide output -- the generic
// tiles spend their time in prologue / epilogue and re-fetch the same 73 KB of weights for every 128-row tile
// (1.45 TB/s of algorithmic bytes, 420 TF/s: under both roofs).  It is pure streaming: 128 B in, 128 B out per pixel.
//
// Persistent kernel, one 512-thread block per CU, looping over 16 x 16-pixel output tiles of one image:
//   * the whole weight matrix [64][576] bf16 sits in LDS for the kernel's life (row stride 1168 B: conflict-free
//     ds_read_b128 of the B fragments);
//   * a tile's 18 x 18-pixel input patch (zero outside the image = the conv padding) is the A operand of all 9 taps:
//     tap (kh, kw) of output pixel (y, x) is patch pixel (y + kh, x + kw), a constant LDS offset -- the input is read
//     from memory once (plus the halo) instead of 9 times from L2.  A pixel's eight 16-byte chunks are XOR-swizzled
//     by its patch column ((x >> 1) & 7), which keeps every 16-lane group of a fragment read on 16 bank quads for
//     every tap offset;
//   * the NEXT tile's patch is loaded global -> registers while this tile computes (6 x 16 B per thread in flight
//     across the whole MFMA phase), and dropped into LDS after the barrier that ends the phase;
//   * wave w owns tile rows 2w, 2w + 1 (one 32-pixel MFMA row block) x all 64 channels: per (tap, 16-channel step) one
//     A fragment, two B fragments, two v_mfma_f32_32x32x16_bf16 -- taps ascending, channels ascending inside a tap =
//     the (r, s, c) K order of igemm.hip / igemm8.hip, so the result has the bits of every other tile (tested);
//   * epilogue: acc * scale + bias, ReLU, bf16, staged through LDS as whole 128-byte pixels, 16-byte stores.
// HBM per tile: 41.5 KB in (27 % halo) + 32 KB out against 147 k MFMA cycles per CU-tile: HBM-bound by design.
#include <cstdlib>

#include "common.h"
#include "igemm_params.h"

namespace {

constexpr int C64_T = 16;                       // tile edge (output pixels)
constexpr int C64_P = C64_T + 2;                // patch edge
constexpr int C64_NPIX = C64_P * C64_P;         // 324 patch pixels
constexpr int C64_WROW = 1168;                  // LDS bytes per weight row (576 bf16 + 16 pad)
constexpr int C64_WBYTES = 64 * C64_WROW;       // 74752
constexpr int C64_PBYTES = C64_NPIX * 128;      // 41472
constexpr int C64_SBYTES = C64_T * C64_T * 128; // 32768 staging
constexpr int C64_LDS = C64_WBYTES + C64_PBYTES + C64_SBYTES;   // 148992
constexpr int C64_NT = 512;
constexpr int C64_PRE = (C64_NPIX * 8 + C64_NT - 1) / C64_NT;   // 16-byte patch pieces per thread (6)

// HT: bf16_t or f16_t (IEEE half, round 6) -- the element type of in / w / out
template <typename HT>
__global__ __launch_bounds__(C64_NT, 2) void conv3x3_c64_kernel(ConvParams p, int tiles_y, int tiles_x, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const wl = smem;
  unsigned char* const pl = smem + C64_WBYTES;
  unsigned char* const sl = smem + C64_WBYTES + C64_PBYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const unsigned short* __restrict__ in = (const unsigned short*)p.in;
  unsigned short* __restrict__ out = (unsigned short*)p.out;

  // ---- weights -> LDS, once
  {
    const uint4* wg = reinterpret_cast<const uint4*>(p.w);
    for (int e = tid; e < 64 * 72; e += C64_NT) {
      const int n = e / 72, c = e - n * 72;
      *reinterpret_cast<uint4*>(wl + n * C64_WROW + c * 16) = wg[e];
    }
  }
  // ---- per-thread patch piece geometry (the same for every tile): piece id -> patch pixel, 16-byte chunk
  int pc_dy[C64_PRE], pc_dx[C64_PRE], pc_lds[C64_PRE];
#pragma unroll
  for (int r = 0; r < C64_PRE; ++r) {
    const int id = tid + C64_NT * r;
    const int px = id >> 3, j = id & 7;
    const int py = px / C64_P, pxx = px - py * C64_P;
    pc_dy[r] = py - 1;
    pc_dx[r] = pxx - 1;
    pc_lds[r] = id < C64_NPIX * 8 ? px * 128 + ((j ^ ((pxx >> 1) & 7)) * 16) : -1;
  }
  auto tile_origin = [&](int t, int& n, int& y0, int& x0) {
    const int per_img = tiles_y * tiles_x;
    n = t / per_img;
    const int r = t - n * per_img;
    const int ty = r / tiles_x;
    y0 = ty * C64_T;
    x0 = (r - ty * tiles_x) * C64_T;
  };
  uint4 pre[C64_PRE];
  auto load_patch = [&](int t) {            // global -> registers (zeros outside the image / past the last tile)
    int n, y0, x0;
    tile_origin(t < ntiles ? t : 0, n, y0, x0);
#pragma unroll
    for (int r = 0; r < C64_PRE; ++r) {
      const int yy = y0 + pc_dy[r], xx = x0 + pc_dx[r];
      const bool ok = t < ntiles && pc_lds[r] >= 0 && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      const int j = (tid + C64_NT * r) & 7;
      pre[r] = make_uint4(0, 0, 0, 0);
      if (ok) pre[r] = *reinterpret_cast<const uint4*>(in + (((size_t)n * p.H + yy) * p.W + xx) * 64 + j * 8);
    }
  };
  auto store_patch = [&]() {                // registers -> LDS
#pragma unroll
    for (int r = 0; r < C64_PRE; ++r)
      if (pc_lds[r] >= 0) *reinterpret_cast<uint4*>(pl + pc_lds[r]) = pre[r];
  };

  // ---- fragment addresses.  A: row block = tile rows 2 wave + (l31 >> 4), column l31 & 15; K half h
  const int ay = 2 * wave + (l31 >> 4), ax = l31 & 15;
  int a_off[3][4];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int xs = ax + kw;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a_off[kw][ks] = (ay * C64_P + xs) * 128 + (((2 * ks + h) ^ ((xs >> 1) & 7)) * 16);
  }
  const int b_off0 = l31 * C64_WROW + h * 16, b_off1 = (32 + l31) * C64_WROW + h * 16;
  const float sc0 = p.scale ? p.scale[l31] : 1.f, sc1 = p.scale ? p.scale[32 + l31] : 1.f;
  const float bi0 = p.bias ? p.bias[l31] : 0.f, bi1 = p.bias ? p.bias[32 + l31] : 0.f;
  const float neg_slope = p.relu == 1 ? 0.f : (p.relu == 2 ? 0.1f : 1.f);

  int t = blockIdx.x;
  load_patch(t);
  store_patch();
  load_patch(t + gridDim.x);
  __syncthreads();
  for (; t < ntiles; t += gridDim.x) {
    // ---- 1. the 9 taps x 4 channel steps: 72 MFMAs per wave
    f32x16_t acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const u32x4_t a = *reinterpret_cast<const u32x4_t*>(pl + a_off[kw][ks] + kh * (C64_P * 128));
          const int kb = (kh * 3 + kw) * 128 + ks * 32;
          const u32x4_t b0 = *reinterpret_cast<const u32x4_t*>(wl + b_off0 + kb);
          const u32x4_t b1 = *reinterpret_cast<const u32x4_t*>(wl + b_off1 + kb);
          acc0 = Half16<HT>::mfma32(a, b0, acc0);
          acc1 = Half16<HT>::mfma32(a, b1, acc1);
        }
    __syncthreads();                        // every wave is done with this tile's patch
    // ---- 2. next tile's patch into LDS; this tile's outputs into the staging area (pixel-major, 128 B per pixel)
    store_patch();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pix = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      float v0 = acc0[r] * sc0 + bi0, v1 = acc1[r] * sc1 + bi1;
      unsigned short h0, h1;
      if (p.relu == 1) {       // ReLU on the ROUNDED value, as the igemm / igemm8 fast epilogues do it (max of the packed bits
        h0 = Half16<HT>::cvt(v0);  // with 0: a negative input gives +0, never -0; a NaN with the sign bit set gives 0 as well) --
        h1 = Half16<HT>::cvt(v1);  // bit identity with the generic tiles includes the sign of zero
        h0 = (short)h0 < 0 ? (unsigned short)0 : h0;
        h1 = (short)h1 < 0 ? (unsigned short)0 : h1;
      } else {
        h0 = Half16<HT>::cvt(v0 > 0.f ? v0 : v0 * neg_slope);
        h1 = Half16<HT>::cvt(v1 > 0.f ? v1 : v1 * neg_slope);
      }
      *reinterpret_cast<unsigned short*>(sl + pix * 128 + l31 * 2) = h0;
      *reinterpret_cast<unsigned short*>(sl + pix * 128 + (32 + l31) * 2) = h1;
    }
    __syncthreads();
    // ---- 3. read-out: 16-byte stores, a tile row = 2 KB of contiguous memory; then request the patch after next
    {
      int n, y0, x0;
      tile_origin(t, n, y0, x0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int id = tid + C64_NT * r;
        const int pix = id >> 3, j = id & 7;
        const int yy = y0 + (pix >> 4), xx = x0 + (pix & 15);
        if (yy < p.H && xx < p.W)
          *reinterpret_cast<uint4*>(out + (((size_t)n * p.H + yy) * p.W + xx) * 64 + j * 8) =
              *reinterpret_cast<const uint4*>(sl + pix * 128 + j * 16);
      }
    }
    load_patch(t + 2 * gridDim.x);
  }
}

}  // namespace

// 1 when mega_conv64_launch takes this layer (bf16 3x3 / stride 1 / pad 1 / dilation 1, 64 -> 64 channels, bf16 out,
// no residual, no split-K, enough tiles to fill the chip)
int mega_conv64_supports(const ConvParams& p, int out_f32) {
  static const bool off = getenv("MEGA_CONV64") != nullptr && getenv("MEGA_CONV64")[0] == '0';
  if (off || out_f32) return 0;
  if (!(p.R == 3 && p.S == 3 && p.Cin == 64 && p.Cout == 64 && p.stride == 1 && p.pad == 1 && p.dil == 1)) return 0;
  if (p.res || p.ksplit != 1 || p.ldo != 64) return 0;
  const long tiles = (long)p.N * cdiv(p.H, C64_T) * cdiv(p.W, C64_T);
  return tiles >= 128;
}

int mega_conv64_launch(const ConvParams& p, int half_dtype, hipStream_t st) {
  if (half_dtype != MEGA_BF16 && half_dtype != MEGA_F16) return MEGA_ERR_ARG;
  const int ty = cdiv(p.H, C64_T), tx = cdiv(p.W, C64_T);
  const long tiles = (long)p.N * ty * tx;
  if (tiles > 0x7FFFFFFF) return MEGA_ERR_ARG;
  int cus = 256;
  {
    static int cached[64] = {0};       // per device id: a multi-GPU process may hold devices with different CU counts
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!cached[dev]) {
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cached[dev] = n;
      else cached[dev] = 256;
    }
    cus = cached[dev];
  }
  const int grid = (int)(tiles < cus ? tiles : cus);
  if (half_dtype == MEGA_F16) {
    (void)hipFuncSetAttribute((const void*)conv3x3_c64_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, C64_LDS);
    hipLaunchKernelGGL(conv3x3_c64_kernel<f16_t>, dim3(grid), dim3(C64_NT), C64_LDS, st, p, ty, tx, (int)tiles);
  } else {
    (void)hipFuncSetAttribute((const void*)conv3x3_c64_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, C64_LDS);
    hipLaunchKernelGGL(conv3x3_c64_kernel<bf16_t>, dim3(grid), dim3(C64_NT), C64_LDS, st, p, ty, tx, (int)tiles);
  }
  return mega_check_launch();
}
