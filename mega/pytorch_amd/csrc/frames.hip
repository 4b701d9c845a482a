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
    const float r = ((float)p[0] / 255.f) * 255.f, g = ((float)p[1] / 255.f) * 255.f,
                b = ((float)p[2] / 255.f) * 255.f;
    float* o = out + n * 3 * hw + px;
    if (to_bgr) {
      o[0] = b - m0; o[hw] = g - m1; o[2 * hw] = r - m2;
    } else {
      o[0] = r - m0; o[hw] = g - m1; o[2 * hw] = b - m2;
    }
  }
}
}  // namespace

int g_mega_last_hip_error = 0;

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
