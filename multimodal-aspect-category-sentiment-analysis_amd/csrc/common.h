// Shared device helpers for the FCMF gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fcmf_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FCMF_CHECK_LAUNCH()                                            \
  do {                                                                 \
    hipError_t e__ = hipGetLastError();                                \
    if (e__ != hipSuccess) return FCMF_ERR_LAUNCH;                     \
  } while (0)

// ---- scalar conversions ----------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T x);
template <> __device__ __forceinline__ float to_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

// 4-wide vector load/store of T as float4 (T = float: 16 B, T = bf16: 8 B)
template <typename T> struct Vec4;
template <> struct Vec4<float> {
  typedef float4 raw;      // as loaded, before conversion (software-pipelined loops hold the next row in this form)
  static __device__ __forceinline__ raw load_raw(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ float4 cvt(raw v) { return v; }
  static __device__ __forceinline__ float4 load(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void store(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
};
template <> struct Vec4<bf16_t> {
  typedef bf16x4 raw;
  static __device__ __forceinline__ raw load_raw(const bf16_t* p) { return *reinterpret_cast<const bf16x4*>(p); }
  static __device__ __forceinline__ float4 cvt(raw v) { return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]); }
  static __device__ __forceinline__ float4 load(const bf16_t* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  }
  static __device__ __forceinline__ void store(bf16_t* p, float4 v) {
    bf16x4 o;
    o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
    *reinterpret_cast<bf16x4*>(p) = o;
  }
};

// ---- wave / block reductions (wave = 64 lanes) --------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- counter-based dropout RNG --------------------------------------------------------------
// Stateless hash of (seed, element index): forward and backward regenerate the same mask without storing
// it.  One 32-bit hash (the "lowbias32" integer finaliser: two 32-bit multiplies -- integer multiplies are
// quarter rate on CDNA, so they are what a mask costs) decides TWO consecutive elements, 16 bits each: an
// element is dropped when its 16 bits fall below p * 65536 (p is honoured to 1.5e-5).
// (the two multiply rounds, given x = lo ^ seed_lo ^ rotl(hi, 7): kernels whose `hi` and seed are wave-uniform fold that part
//  into one scalar and call this directly)
__device__ __forceinline__ uint32_t fcmf_hash32_rounds(uint32_t x, uint32_t s1) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= s1;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t fcmf_hash32(uint64_t seed, uint64_t pair) {
  const uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
  const uint32_t lo = (uint32_t)pair, hi = (uint32_t)(pair >> 32);
  return fcmf_hash32_rounds(lo ^ s0 ^ ((hi << 7) | (hi >> 25)), s1);
}
__device__ __forceinline__ uint32_t dropout_threshold(float p) { return (uint32_t)(p * 65536.0f + 0.5f); }
// returns the multiplier applied to the element: 0 (dropped) or 1/(1-p) (kept)
__device__ __forceinline__ float dropout_mult(uint64_t seed, uint64_t idx, float p, float inv_keep) {
  const uint32_t h = fcmf_hash32(seed, idx >> 1);
  const uint32_t bits = (idx & 1) ? (h >> 16) : (h & 0xFFFFu);
  return bits >= dropout_threshold(p) ? inv_keep : 0.0f;
}
// multipliers of the 4 consecutive elements base .. base+3 (same values as four dropout_mult calls):
// two hashes when base is even, three otherwise
__device__ __forceinline__ void dropout_mult4(uint64_t seed, uint64_t base, float p, float inv_keep, float (&m)[4]) {
  const uint32_t thr = dropout_threshold(p);
  const uint64_t k0 = base >> 1;
  const uint32_t h0 = fcmf_hash32(seed, k0), h1 = fcmf_hash32(seed, k0 + 1);
  if ((base & 1) == 0) {
    m[0] = (h0 & 0xFFFFu) >= thr ? inv_keep : 0.f; m[1] = (h0 >> 16) >= thr ? inv_keep : 0.f;
    m[2] = (h1 & 0xFFFFu) >= thr ? inv_keep : 0.f; m[3] = (h1 >> 16) >= thr ? inv_keep : 0.f;
  } else {
    const uint32_t h2 = fcmf_hash32(seed, k0 + 2);
    m[0] = (h0 >> 16) >= thr ? inv_keep : 0.f;     m[1] = (h1 & 0xFFFFu) >= thr ? inv_keep : 0.f;
    m[2] = (h1 >> 16) >= thr ? inv_keep : 0.f;     m[3] = (h2 & 0xFFFFu) >= thr ? inv_keep : 0.f;
  }
}

// exact-erf GELU and its derivative (mm_modeling.py:10-15)
__device__ __forceinline__ float gelu_f(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float dgelu_f(float x) {
  float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
