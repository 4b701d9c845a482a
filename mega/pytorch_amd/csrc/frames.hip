// Frame pre-processing on device: uint8 HWC RGB frame -> normalised f32 CHW tensor, i.e. the test-time
// transform chain of the reference for a frame whose size is already a fixed point of the resize rule
// (600x1000): ToTensor (/255), to_bgr255 (channel flip, *255), Normalize(mean, std = 1)
// (mega_core/data/transforms/transforms.py:83-129, data/transforms/build.py:26-45).
#include "common.h"

namespace {
__global__ __launch_bounds__(256) void preprocess_kernel(const unsigned char* __restrict__ in, float* __restrict__ out,
                                                         int N, int H, int W, float m0, float m1, float m2,
                                                         int to_bgr) {
  const size_t hw = (size_t)H * W;
  const size_t total = (size_t)N * hw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / hw, px = i - n * hw;
    const unsigned char* p = in + i * 3;
    // ToTensor's x / 255 followed by to_bgr255's * 255 returns x exactly for every uint8 x under IEEE f32
    // round-to-nearest (checked for all 256 values in tests/test_feed.py), so the pair is the identity; the
    // device's f32 division is NOT correctly rounded by default, so computing it literally would be off by 1 ulp.
    const float r = (float)p[0], g = (float)p[1], b = (float)p[2];
    float* o = out + n * 3 * hw + px;
    if (to_bgr) {
      o[0] = b - m0; o[hw] = g - m1; o[2 * hw] = r - m2;
    } else {
      o[0] = r - m0; o[hw] = g - m1; o[2 * hw] = b - m2;
    }
  }
}

// PIL's 8-bit separable resampler (Pillow src/libImaging/Resample.c ImagingResampleHorizontal_8bpc /
// ImagingResampleVertical_8bpc): fixed-point coefficients (PRECISION_BITS = 22) prepared on the host, integer
// accumulate from 1 << 21, arithmetic shift, clamp to [0,255]; the horizontal pass writes uint8.  Integer work:
// results are bit-identical to PIL's for the same coefficient tables.
constexpr int kPrecisionBits = 32 - 8 - 2;

__device__ __forceinline__ unsigned char clip8(int v) {
  v >>= kPrecisionBits;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// in [N][Hi][Wi][3] -> out [N][Hi][Wo][3]; one thread per output pixel (3 channels).
__global__ __launch_bounds__(256) void resize_h_kernel(const unsigned char* __restrict__ in,
                                                       unsigned char* __restrict__ out,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk,
                                                       int ksize, int N, int Hi, int Wi, int Wo) {
  const size_t total = (size_t)N * Hi * Wo;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xo = (int)(i % Wo);
    const size_t row = i / Wo;
    const int xmin = bounds[2 * xo], xmax = bounds[2 * xo + 1];
    const int* k = kk + (size_t)xo * ksize;
    const unsigned char* p = in + (row * Wi + xmin) * 3;
    int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < xmax; ++x) {
      const int w = k[x];
      s0 += (int)p[3 * x] * w; s1 += (int)p[3 * x + 1] * w; s2 += (int)p[3 * x + 2] * w;
    }
    unsigned char* o = out + i * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
  }
}

// in [N][Hi][W3] -> out [N][Ho][W3] (W3 = W*3 bytes per row); one thread per output byte, coalesced along the row.
__global__ __launch_bounds__(256) void resize_v_kernel(const unsigned char* __restrict__ in,
                                                       unsigned char* __restrict__ out,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk,
                                                       int ksize, int N, int Hi, int Ho, int W3) {
  const size_t total = (size_t)N * Ho * W3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xb = (int)(i % W3);
    const size_t t = i / W3;
    const int yo = (int)(t % Ho);
    const size_t n = t / Ho;
    const int ymin = bounds[2 * yo], ymax = bounds[2 * yo + 1];
    const int* k = kk + (size_t)yo * ksize;
    const unsigned char* p = in + (n * Hi + ymin) * (size_t)W3 + xb;
    int s = 1 << (kPrecisionBits - 1);
    for (int y = 0; y < ymax; ++y) s += (int)p[(size_t)y * W3] * k[y];
    out[i] = clip8(s);
  }
}
}  // namespace

int g_mega_last_hip_error = 0;
int g_mega_pending_hip_error = 0;

extern "C" const char* mega_last_error_string() {
  return hipGetErrorString((hipError_t)g_mega_last_hip_error);
}

extern "C" int mega_preprocess_frames(const unsigned char* in, float* out, int N, int H, int W, float mean0,
                                      float mean1, float mean2, int to_bgr, void* stream) {
  mega_clear_error();
  if (!in || !out || N <= 0 || H <= 0 || W <= 0) return MEGA_ERR_ARG;
  const size_t total = (size_t)N * H * W;
  const int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  hipLaunchKernelGGL(preprocess_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, N, H, W, mean0,
                     mean1, mean2, to_bgr);
  return mega_check_launch();
}

extern "C" int mega_resize_bilinear_u8(const unsigned char* in, unsigned char* out, unsigned char* tmp, int N, int Hi,
                                       int Wi, int Ho, int Wo, const int* bounds_h, const int* coef_h, int ksize_h,
                                       const int* bounds_v, const int* coef_v, int ksize_v, void* stream) {
  mega_clear_error();
  if (!in || !out || N <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return MEGA_ERR_ARG;
  const bool need_h = Wo != Wi, need_v = Ho != Hi;
  if ((need_h && (!bounds_h || !coef_h || ksize_h <= 0)) || (need_v && (!bounds_v || !coef_v || ksize_v <= 0)) ||
      (need_h && need_v && !tmp))
    return MEGA_ERR_ARG;
  auto grid = [](size_t total) { return dim3((unsigned)((total + 255) / 256 > 32768 ? 32768 : (total + 255) / 256)); };
  hipStream_t st = (hipStream_t)stream;
  if (!need_h && !need_v) {
    hipError_t e = hipMemcpyAsync(out, in, (size_t)N * Hi * Wi * 3, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) { g_mega_last_hip_error = (int)e; return MEGA_ERR_LAUNCH; }
    return MEGA_OK;
  }
  const unsigned char* src = in;
  if (need_h) {   // horizontal first, exactly as ImagingResampleInner orders the passes
    unsigned char* dst = need_v ? tmp : out;
    hipLaunchKernelGGL(resize_h_kernel, grid((size_t)N * Hi * Wo), dim3(256), 0, st, src, dst, bounds_h, coef_h,
                       ksize_h, N, Hi, Wi, Wo);
    src = dst;
  }
  if (need_v)
    hipLaunchKernelGGL(resize_v_kernel, grid((size_t)N * Ho * Wo * 3), dim3(256), 0, st, src, out, bounds_v, coef_v,
                       ksize_v, N, Hi, Ho, Wo * 3);
  return mega_check_launch();
}
