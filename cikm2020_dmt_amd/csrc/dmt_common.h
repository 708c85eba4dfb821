// Shared device/host helpers for libdmt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/dmt_hip.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits

void dmt_set_error(const char* fmt, ...);
extern int g_dmt_route_trace;     // dmt_route_trace(1): every successful launch notes its route label (which kernel variant ran)
void dmt_route_note(const char* what);

#define DMT_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      dmt_set_error(__VA_ARGS__);           \
      return DMT_ERR_ARG;                   \
    }                                       \
  } while (0)

#define DMT_CHECK_LAUNCH(what)                                                       \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      dmt_set_error("%s: launch failed: %s", what, hipGetErrorString(e__));          \
      return DMT_ERR_LAUNCH;                                                         \
    }                                                                                \
    if (g_dmt_route_trace) dmt_route_note(what);                                     \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                           // round to nearest even
  return (bf16_t)(u >> 16);
}

// two floats -> packed pair of bf16 (round to nearest even): compiles to ONE v_cvt_pk_bf16_f32 that hipcc schedules itself
// (an inline-asm cvt feeding an MFMA / LDS store needs hand-placed wait states: wrong results were seen under register pressure)
typedef __bf16 dmt_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float dmt_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned dmt_pack_bf16(float lo, float hi) {
  const dmt_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, dmt_bf16x2_t));
}

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// Wavefront-wide reductions on the VALU: four DPP steps (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror) leave
// every lane with the result of its 16-lane row, four v_readlane combine the rows.  (__shfl_xor is a ds_bpermute_b32: six LDS
// instructions per reduction -- the LayerNorm kernels, two reductions per 640-byte row, were bound by the LDS pipe, not by memory.)
template <int CTRL> __device__ __forceinline__ float dmt_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float dmt_lane_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ float wave_sum(float v) {
  v += dmt_dpp<0xB1>(v);
  v += dmt_dpp<0x4E>(v);
  v += dmt_dpp<0x141>(v);
  v += dmt_dpp<0x140>(v);
  return (dmt_lane_f(v, 0) + dmt_lane_f(v, 16)) + (dmt_lane_f(v, 32) + dmt_lane_f(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dmt_dpp<0xB1>(v));
  v = fmaxf(v, dmt_dpp<0x4E>(v));
  v = fmaxf(v, dmt_dpp<0x141>(v));
  v = fmaxf(v, dmt_dpp<0x140>(v));
  return fmaxf(fmaxf(dmt_lane_f(v, 0), dmt_lane_f(v, 16)), fmaxf(dmt_lane_f(v, 32), dmt_lane_f(v, 48)));
}

// counter-based dropout mask shared by every kernel (and restated in oracle/dmt_oracle.py:dropout_mask)
__device__ __forceinline__ uint32_t dmt_mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ bool dmt_drop_keep(uint32_t seed, uint32_t idx, uint32_t thr24) { return (dmt_mix32(idx ^ seed) >> 8) < thr24; }

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
