// Shared helpers for the gfx950 kernels of libelvis_amd.  CDNA4 only: 64-wide waves,
// no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/elvis_amd.h"

#define ELVIS_WAVE 64

void elvis_set_error(const char* fmt, ...);

#define ELVIS_REQUIRE(cond, ...)                      \
    do {                                              \
        if (!(cond)) {                                \
            elvis_set_error(__VA_ARGS__);             \
            return ELVIS_E_INVALID;                   \
        }                                             \
    } while (0)

#define ELVIS_CHECK_LAUNCH(name)                                              \
    do {                                                                      \
        hipError_t e__ = hipGetLastError();                                   \
        if (e__ != hipSuccess) {                                              \
            elvis_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return ELVIS_E_RUNTIME;                                           \
        }                                                                     \
    } while (0)

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

template <typename T> struct DT;
template <> struct DT<float> {
    static constexpr int code = ELVIS_F32;
    static constexpr int VEC = 4;  // elements per 16 bytes
};
template <> struct DT<half_t> {
    static constexpr int code = ELVIS_F16;
    static constexpr int VEC = 8;
};

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(half_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ half_t from_f<half_t>(float v) { return (half_t)v; }

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// erf-GELU.  erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below f16 / the fp32 parity bar):
// libm's erff is ~3x the instructions and made the Swin MLP's fc1 epilogue VALU-bound.
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float q = fmaf(t, 1.061405429f, -1.453152027f);
    q = fmaf(q, t, 1.421413741f);
    q = fmaf(q, t, -0.284496736f);
    q = fmaf(q, t, 0.254829592f);
    q *= t;
    const float erf_abs = fmaf(-q, __expf(-(z * z)), 1.0f);
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15); every lane of the row gets the total
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
    return v;
}

// max over the 16 lanes of a DPP row; every lane of the row gets the result
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false)));
    return v;
}

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
