// Shared device helpers for the MEGA gfx950 kernels (wave64, CDNA4).  No CUDA-compat layer:
// this header is HIP/gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MEGA_F32 0
#define MEGA_BF16 1

#define MEGA_OK 0
#define MEGA_ERR_ARG 1
#define MEGA_ERR_LAUNCH 2
#define MEGA_ERR_WS 3

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

// round-to-nearest-even (the rule torch uses for float -> bfloat16), NaN stays NaN: gfx950's hardware conversion
// (v_cvt_pk_bf16_f32) -- one instruction per pair, no NaN branch in every epilogue / softmax pack.
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  const __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}

// two floats -> one dword of two bf16 (low half = a): v_cvt_pk_bf16_f32
typedef short s16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
  const bf16x2_v v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int VE = 4;  // elements per 16-byte vector
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int VE = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// hipGetLastError() is sticky per thread and also reports BENIGN codes left behind by other users of the runtime in
// this process (e.g. hipErrorNotReady from an event query of torch's caching allocator): every entry point
// clears it first, so the check after the launches only sees this call's own errors.
// What was pending is not this call's error, so it is not returned -- but a code other than the known-benign
// hipErrorNotReady is kept in g_mega_pending_hip_error so that mega_last_error_string() can name it after a later
// failure (it usually explains it).
static inline void mega_clear_error();


extern int g_mega_last_hip_error;  // defined in frames.hip; read back through mega_last_error_string()
extern int g_mega_pending_hip_error;  // a non-benign error found pending by mega_clear_error()

static inline void mega_clear_error() {
  hipError_t e = hipGetLastError();
  g_mega_pending_hip_error = (e == hipSuccess || e == hipErrorNotReady) ? 0 : (int)e;
}

static inline int mega_check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) g_mega_last_hip_error = (int)e;
  return e == hipSuccess ? MEGA_OK : MEGA_ERR_LAUNCH;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
