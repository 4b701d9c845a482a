// Shared device helpers for the MEGA gfx950 kernels (wave64, CDNA4).  No CUDA-compat layer:
// this header is HIP/gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MEGA_F32 0
#define MEGA_BF16 1
#define MEGA_F16 2

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

// IEEE half (round 6): the second 16-bit matrix-core operand type.  Same MFMA rate, same bytes as bf16; 11 significant bits
// instead of 8 (1/8 of the rounding noise per hand-off), range +-65504.  A distinct C++ type, so every kernel template
// that is instantiated for bf16_t can be instantiated for f16_t; Half16<T> holds what differs between the two.
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

// What is rounded to f16 is always the f32 VALUE of the epilogue's expression.  Without the (free) register barrier below hipcc
// folds `(_Float16)fmaf(a, b, c)` into v_fma_mixlo_f16 -- one rounding of the exact a b + c instead of f32 then f16 -- in SOME
// kernels (bneck64: 96 of them) and not in others (igemm / igemm8 stage through LDS in between): the fused bottleneck then
// differed from the launches it replaces in 0.14 % of its elements (1 ulp).  Every kernel of the path must give the same bits
// for the same arithmetic (batch invariance), so the conversion's input is made opaque.
__device__ __forceinline__ float f16_src(float x) {
  asm("" : "+v"(x));
  return x;
}

// two floats -> one dword of two f16 (low half = a), round to nearest even (the rule torch uses for float -> half)
__device__ __forceinline__ unsigned pack_f16x2(float a, float b) {
  const f16x2_t v = {(_Float16)f16_src(a), (_Float16)f16_src(b)};
  return __builtin_bit_cast(unsigned, v);
}

template <typename T> struct Half16;
template <> struct Half16<bf16_t> {
  static constexpr int CODE = MEGA_BF16;
  __device__ static __forceinline__ unsigned pack2(float a, float b) { return pack_bf16x2(a, b); }
  __device__ static __forceinline__ unsigned short cvt(float a) { return f32_to_bf16(a); }        // the 16 bits of T(a)
  __device__ static __forceinline__ float lo(unsigned w) { return __uint_as_float(w << 16); }     // element 0 of a packed pair
  __device__ static __forceinline__ float hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
  __device__ static __forceinline__ float one(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
  __device__ static __forceinline__ f32x16_t mfma32(const u32x4_t& a, const u32x4_t& b, const f32x16_t& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
  __device__ static __forceinline__ f32x4_t mfma16(const u32x4_t& a, const u32x4_t& b, const f32x4_t& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Half16<f16_t> {
  static constexpr int CODE = MEGA_F16;
  __device__ static __forceinline__ unsigned pack2(float a, float b) { return pack_f16x2(a, b); }
  __device__ static __forceinline__ unsigned short cvt(float a) { return __builtin_bit_cast(unsigned short, (_Float16)f16_src(a)); }
  __device__ static __forceinline__ float lo(unsigned w) { return (float)__builtin_bit_cast(f16x2_t, w)[0]; }
  __device__ static __forceinline__ float hi(unsigned w) { return (float)__builtin_bit_cast(f16x2_t, w)[1]; }
  __device__ static __forceinline__ float one(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
  __device__ static __forceinline__ f32x16_t mfma32(const u32x4_t& a, const u32x4_t& b, const f32x16_t& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
  __device__ static __forceinline__ f32x4_t mfma16(const u32x4_t& a, const u32x4_t& b, const f32x4_t& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

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
template <> struct Elem<f16_t> {
  static constexpr int VE = 8;
  __device__ static __forceinline__ float ld(const f16_t* p) { return (float)*p; }
  __device__ static __forceinline__ void st(f16_t* p, float v) { *p = (_Float16)f16_src(v); }
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
