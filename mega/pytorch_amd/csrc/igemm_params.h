// Launch parameters shared by the implicit-GEMM kernels (igemm.hip: register-staged 2-barrier tiles;
// igemm8.hip: LDS-DMA 8-phase 256-wide tiles).
#pragma once
#include <hip/hip_runtime.h>

struct ConvParams {
  const void* in;
  const void* w;
  const float* scale;
  const float* bias;
  const void* res;
  void* out;
  int N, H, W, Cin, Cout, R, S, stride, pad, dil, Ho, Wo;
  int M, K;
  int ldo, ldr;
  int relu;          // 0 none, 1 ReLU, 2 LeakyReLU(0.1)
  unsigned in_bytes, w_bytes;
  int ksplit;        // > 1: blockIdx.z owns a contiguous range of K-tiles and writes raw f32 partial sums
  float* partial;    // [ksplit][M][Cout] f32 when ksplit > 1
  // ---- igemm8 only.  Split-precision ("hi | lo planes") activations, sp != 0 (igemm8.hip, SP kernels): an f32 activation
  //      x [M][C] is stored as bf16 [M][2C] = [hi | lo], hi = bf16(x), lo = bf16(x - hi)  (x = hi + lo to ~2^-17).
  int sp;            // 1: the fields below apply (mega_conv2d_nhwc_sp); 0: plain tensors (every other entry point)
  int ldi;           // pixel stride of `in` in elements (2C for a split input; Cin when the input is a plain tensor)
  int kwrap;         // contraction channel at which the SOURCE channel index wraps to 0 (0: never).  Cin = 3C, kwrap = 2C
                     // reads the planes as [hi | lo | hi]: against weights packed [Wh | Wh | Wl] per tap the K = 3C
                     // contraction is x_hi.Wh + x_lo.Wh + x_hi.Wl = x.W to ~2^-16 on the bf16 matrix cores
  int split_out;     // 1: out is split planes [M][ldo] with hi at column n and lo at column Cout + n (bf16 output only)
                     // (a residual given to an SP launch is ALWAYS split planes: res[m][n] + res[m][Cout + n])
  unsigned mg_howo, sh_howo, mg_wo, sh_wo;   // magic multipliers / shifts for m / (Ho Wo) and rem / Wo (set by the launcher)
  // ---- igemm.hip only.  Sub-pixel output (mega_conv2d_nhwc_subpixel): a ConvTranspose2d(k = 4, s = 2) run as a 2 x 2 / pad 1
  //      conv with 4 ps_C output columns n = (a, b, co): GEMM row m = (t, mh, mw) lands at pixel (2 mh + a - ps_crop,
  //      2 mw + b - ps_crop), channel ps_coff + co of an NHWC tensor [N][ps_H][ps_W][ldo]; pixels outside it are dropped
  int ps;            // 1: the fields below apply
  int ps_H, ps_W, ps_C, ps_crop, ps_coff;
};

// igemm8.hip: bf16 operands, 8 waves, 256 (BM8 rows) x 256 tile; returns MEGA_OK / MEGA_ERR_*.  out_f32: 0 bf16, 1 f32.
// half_dtype: MEGA_BF16 / MEGA_F16 = the 16-bit type of in / w / residual (and of the output when out_f32 == 0)
int mega_igemm8_launch(const ConvParams& p, int bm, int out_f32, int half_dtype, hipStream_t st);
// 1 when the shape is one igemm8 can take (bf16, Cin % 64 == 0, operands < 2 GiB, ...)
int mega_igemm8_supports(const ConvParams& p);
// 1 when a launch of `taps` = R * S kernel taps and GEMM depth K belongs to igemm8's streaming class (1x1, K <= 512)
int mega_igemm8_streaming(int taps, int K);
// igemm4.hip: the same tiles on 4 waves of 512 registers (128 x 128 outputs per wave), for the shapes its buffer-addressed
// epilogue serves (mega_igemm4_supports) -- the matrix-core-bound launch class by default (choose_tile, igemm.hip)
int mega_igemm4_supports(const ConvParams& p, int out_f32);
int mega_igemm4_launch(const ConvParams& p, int bm, int out_f32, int half_dtype, hipStream_t st);
// igemm2.hip: 128 x 256 tiles, K-tile 32, 4 waves, two blocks per CU -- the streaming launch class (1x1, K <= 512) by default
int mega_igemm2_supports(const ConvParams& p, int out_f32);
int mega_igemm2_launch(const ConvParams& p, int out_f32, int half_dtype, hipStream_t st);
// stream1x1.hip: persistent 1x1 / stride-1 conv with K = 128 / 256 (layer3's / layer2's conv3 + residual): weights resident in LDS,
// a wave owns a 32 x 256 output tile, operands straight from global memory into MFMA fragments; bit-identical to the tile kernels
int mega_stream1x1_supports(const ConvParams& p, int out_f32);
int mega_stream1x1_launch(const ConvParams& p, int half_dtype, hipStream_t st);
// conv64.hip: persistent 3x3 / 64 -> 64 channel kernel (layer1's conv2); bit-identical to the generic tiles
int mega_conv64_supports(const ConvParams& p, int out_f32);
int mega_conv64_launch(const ConvParams& p, int half_dtype, hipStream_t st);
