// uint8 <-> float conversion, bicubic upsample, VQ nearest-code, reflect-pad + axpy, crop.
// All HBM-bound elementwise / gather kernels.
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void u8_to_float_kernel(const uint8_t* __restrict__ src, T* __restrict__ dst,
                                                          long long pixels, int pitch, float scale, float bias,
                                                          int swap_rb, int div255) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pixels;
         i += (long long)gridDim.x * blockDim.x) {
        const uint8_t* s = src + i * 3;
        float v[3] = {(float)s[0], (float)s[1], (float)s[2]};
        if (swap_rb) { float t = v[0]; v[0] = v[2]; v[2] = t; }
        T* d = dst + i * pitch;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float t = div255 ? __fdiv_rn(v[c], 255.0f) : v[c];
            d[c] = from_f<T>(__fadd_rn(__fmul_rn(t, scale), bias));
        }
        for (int c = 3; c < pitch; ++c) d[c] = from_f<T>(0.f);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void float_to_u8_kernel(const T* __restrict__ src, uint8_t* __restrict__ dst,
                                                          float* __restrict__ f32_out, long long pixels, int pitch,
                                                          float scale, float bias, int mode, int swap_rb) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pixels;
         i += (long long)gridDim.x * blockDim.x) {
        const T* s = src + i * pitch;
        float t[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = __fadd_rn(__fmul_rn(to_f(s[c]), scale), bias);
            t[c] = fminf(fmaxf(v, 0.0f), 1.0f);
        }
        if (swap_rb) { float x = t[0]; t[0] = t[2]; t[2] = x; }
        uint8_t* d = dst + i * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float q = __fmul_rn(t[c], 255.0f);
            int u = mode == 0 ? __float2int_rn(q) : (int)q;
            d[c] = (uint8_t)(u < 0 ? 0 : (u > 255 ? 255 : u));
            if (f32_out) f32_out[i * 3 + c] = t[c];
        }
    }
}

__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

template <typename T>
__global__ __launch_bounds__(256) void bicubic_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w,
                                                      int c, int pitch_in, int pitch_out, int sf) {
    const int ho = h * sf, wo = w * sf;
    const long long total = (long long)n * ho * wo;
    const float A = -0.75f;
    const float rs = 1.0f / (float)sf;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int ox = (int)(i % wo);
        long long r = i / wo;
        int oy = (int)(r % ho);
        int ni = (int)(r / ho);
        float sy = rs * ((float)oy + 0.5f) - 0.5f, sx = rs * ((float)ox + 0.5f) - 0.5f;
        float fy = floorf(sy), fx = floorf(sx);
        float ty = sy - fy, tx = sx - fx;
        int iy = (int)fy, ix = (int)fx;
        float wy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
        float wx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
        T* d = y + i * pitch_out;
        for (int ch = 0; ch < c; ++ch) {
            float acc = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                int yy = iy - 1 + a;
                yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
                float rowv = 0.f;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    int xx = ix - 1 + b;
                    xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
                    rowv += to_f(x[(((long long)ni * h + yy) * w + xx) * pitch_in + ch]) * wx[b];
                }
                acc += rowv * wy[a];
            }
            d[ch] = from_f<T>(acc);
        }
        for (int ch = c; ch < pitch_out; ++ch) d[ch] = from_f<T>(0.f);
    }
}

// VQ nearest code.  Distances are evaluated with explicit (non-contracted) IEEE mul/add in the
// order ((dx*dx) + dy*dy) + dz*dz ... so that identical inputs give the oracle's argmin.
// Four lanes share a pixel: lane part p scans the codebook quarter [p*Q, (p+1)*Q) staged through
// LDS, then the (distance, index) pairs are min-reduced across the four lanes with "smaller
// index wins ties" - identical to a sequential first-minimum scan, 4x the parallelism.
template <typename T>
__global__ __launch_bounds__(256) void vq_nearest_kernel(const T* __restrict__ z, T* __restrict__ zq,
                                                         int32_t* __restrict__ idx_out, long long pixels, int c,
                                                         int pitch_in, int pitch_out,
                                                         const float* __restrict__ codebook, int n_embed) {
    constexpr int CHUNK = 256;   // codes per part per LDS stage
    constexpr int MAXC = 4;
    constexpr int PPT = 4;       // pixels per thread: every staged code is applied to 4 pixels
    __shared__ float cb[4 * CHUNK * MAXC];
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long px0 = (t >> 2) * PPT;
    const int part = (int)(t & 3);
    const int Q = (n_embed + 3) / 4;
    float zv[PPT][MAXC];
#pragma unroll
    for (int q = 0; q < PPT; ++q)
#pragma unroll
        for (int k = 0; k < MAXC; ++k) zv[q][k] = (px0 + q < pixels && k < c) ? to_f(z[(px0 + q) * pitch_in + k]) : 0.f;
    float best[PPT];
    int best_i[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        best[q] = 3.0e38f;
        best_i[q] = 0x7fffffff;
    }
    for (int base = 0; base < Q; base += CHUNK) {
        __syncthreads();
        // stage codes [p*Q + base, p*Q + base + CHUNK) of every part p
        for (int e = threadIdx.x; e < 4 * CHUNK; e += blockDim.x) {
            int p = e / CHUNK, k = e - p * CHUNK;
            int code = p * Q + base + k;
            bool ok = base + k < Q && code < n_embed;
            for (int d = 0; d < c; ++d) cb[(p * CHUNK + k) * MAXC + d] = ok ? codebook[(long long)code * c + d] : 3.0e18f;
        }
        __syncthreads();
        const float* my = cb + part * CHUNK * MAXC;
        const int first = part * Q + base;
        for (int e = 0; e < CHUNK; ++e) {
            float cv[MAXC];
#pragma unroll
            for (int k = 0; k < MAXC; ++k) cv[k] = k < c ? my[e * MAXC + k] : 0.f;
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                float d = 0.f;
#pragma unroll
                for (int k = 0; k < MAXC; ++k) {
                    if (k < c) {
                        float df = __fsub_rn(zv[q][k], cv[k]);
                        float sq = __fmul_rn(df, df);
                        d = k == 0 ? sq : __fadd_rn(d, sq);
                    }
                }
                if (d < best[q]) {
                    best[q] = d;
                    best_i[q] = first + e;
                }
            }
        }
    }
    // lexicographic (distance, index) min over the 4 lanes of the pixel group
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
#pragma unroll
        for (int o = 1; o < 4; o <<= 1) {
            float od = __shfl_xor(best[q], o, 64);
            int oi = __shfl_xor(best_i[q], o, 64);
            if (od < best[q] || (od == best[q] && oi < best_i[q])) {
                best[q] = od;
                best_i[q] = oi;
            }
        }
        const long long px = px0 + q;
        if (px < pixels && part == 0) {
            if (idx_out) idx_out[px] = best_i[q];
            for (int k = 0; k < c; ++k) zq[px * pitch_out + k] = from_f<T>(codebook[(long long)best_i[q] * c + k]);
            for (int k = c; k < pitch_out; ++k) zq[px * pitch_out + k] = from_f<T>(0.f);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pad_reflect_axpy_kernel(const T* __restrict__ x, T* __restrict__ y, int n,
                                                               int h, int w, int c, int pitch_in, int hp, int wp,
                                                               int pitch_out, int coff, float mul,
                                                               const float* __restrict__ add, float add_mul) {
    const long long total = (long long)n * hp * wp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int xx = (int)(i % wp);
        long long r = i / wp;
        int yy = (int)(r % hp);
        int ni = (int)(r / hp);
        int ry = yy < h ? yy : 2 * (h - 1) - yy;
        int rx = xx < w ? xx : 2 * (w - 1) - xx;
        const T* s = x + (((long long)ni * h + ry) * w + rx) * pitch_in;
        T* d = y + i * pitch_out + coff;
        for (int ch = 0; ch < c; ++ch) {
            float v = __fmul_rn(to_f(s[ch]), mul);
            if (add) v = __fadd_rn(v, __fmul_rn(add_mul, add[(((long long)ni * c + ch) * hp + yy) * wp + xx]));
            d[ch] = from_f<T>(v);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void crop_copy_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h_in,
                                                        int w_in, int pitch_in, int h, int w, int c, int pitch_out) {
    const long long total = (long long)n * h * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int xx = (int)(i % w);
        long long r = i / w;
        int yy = (int)(r % h);
        int ni = (int)(r / h);
        const T* s = x + (((long long)ni * h_in + yy) * w_in + xx) * pitch_in;
        T* d = y + i * pitch_out;
        for (int ch = 0; ch < c; ++ch) d[ch] = s[ch];
        for (int ch = c; ch < pitch_out; ++ch) d[ch] = from_f<T>(0.f);
    }
}

// dtype conversion of a pitched NHWC tensor (mixed-precision section boundaries): 8 elements per thread
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void convert_act_kernel(const TS* __restrict__ x, TD* __restrict__ y, long long total8) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
        TS v[8];
        TD o[8];
        if constexpr (sizeof(TS) == 2) {
            *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(x + i * 8);
        } else {
            *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(x + i * 8);
            *reinterpret_cast<uint4*>(v + 4) = *reinterpret_cast<const uint4*>(x + i * 8 + 4);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = from_f<TD>(to_f(v[k]));
        if constexpr (sizeof(TD) == 2) {
            *reinterpret_cast<uint4*>(y + i * 8) = *reinterpret_cast<uint4*>(o);
        } else {
            *reinterpret_cast<uint4*>(y + i * 8) = *reinterpret_cast<uint4*>(o);
            *reinterpret_cast<uint4*>(y + i * 8 + 4) = *reinterpret_cast<uint4*>(o + 4);
        }
    }
}

inline int grid_for(long long total) {
    long long g = (total + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

#define DISPATCH_DTYPE(dtype, NAME, ...)                                          \
    if ((dtype) == ELVIS_F16) { NAME<half_t> __VA_ARGS__; }                       \
    else if ((dtype) == ELVIS_F32) { NAME<float> __VA_ARGS__; }                   \
    else { elvis_set_error("bad dtype %d", (dtype)); return ELVIS_E_INVALID; }

extern "C" int elvis_u8_to_float(const uint8_t* src, void* dst, int dtype, int n, int h, int w, int pitch, float scale,
                                 float bias, int swap_rb, int div255, elvis_stream_t stream) {
    ELVIS_REQUIRE(src && dst && n > 0 && h > 0 && w > 0 && pitch >= 3, "elvis_u8_to_float: bad argument");
    long long px = (long long)n * h * w;
    if (dtype == ELVIS_F16)
        hipLaunchKernelGGL(u8_to_float_kernel<half_t>, dim3(grid_for(px)), dim3(256), 0, (hipStream_t)stream, src,
                           (half_t*)dst, px, pitch, scale, bias, swap_rb, div255);
    else if (dtype == ELVIS_F32)
        hipLaunchKernelGGL(u8_to_float_kernel<float>, dim3(grid_for(px)), dim3(256), 0, (hipStream_t)stream, src,
                           (float*)dst, px, pitch, scale, bias, swap_rb, div255);
    else
        ELVIS_REQUIRE(false, "elvis_u8_to_float: bad dtype");
    ELVIS_CHECK_LAUNCH("elvis_u8_to_float");
    return ELVIS_OK;
}

extern "C" int elvis_float_to_u8(const void* src, int dtype, uint8_t* dst, float* f32_out, int n, int h, int w,
                                 int pitch, float scale, float bias, int mode, int swap_rb, elvis_stream_t stream) {
    ELVIS_REQUIRE(src && dst && n > 0 && h > 0 && w > 0 && pitch >= 3, "elvis_float_to_u8: bad argument");
    ELVIS_REQUIRE(mode == 0 || mode == 1, "elvis_float_to_u8: mode must be 0 (round) or 1 (truncate)");
    long long px = (long long)n * h * w;
    if (dtype == ELVIS_F16)
        hipLaunchKernelGGL(float_to_u8_kernel<half_t>, dim3(grid_for(px)), dim3(256), 0, (hipStream_t)stream,
                           (const half_t*)src, dst, f32_out, px, pitch, scale, bias, mode, swap_rb);
    else if (dtype == ELVIS_F32)
        hipLaunchKernelGGL(float_to_u8_kernel<float>, dim3(grid_for(px)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)src, dst, f32_out, px, pitch, scale, bias, mode, swap_rb);
    else
        ELVIS_REQUIRE(false, "elvis_float_to_u8: bad dtype");
    ELVIS_CHECK_LAUNCH("elvis_float_to_u8");
    return ELVIS_OK;
}

extern "C" int elvis_bicubic_upsample(const void* x, void* y, int dtype, int n, int h, int w, int c, int pitch_in,
                                      int pitch_out, int sf, elvis_stream_t stream) {
    ELVIS_REQUIRE(x && y && n > 0 && h > 0 && w > 0 && c > 0 && sf >= 1 && pitch_in >= c && pitch_out >= c,
                  "elvis_bicubic_upsample: bad argument");
    long long total = (long long)n * h * sf * w * sf;
    if (dtype == ELVIS_F16)
        hipLaunchKernelGGL(bicubic_kernel<half_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const half_t*)x, (half_t*)y, n, h, w, c, pitch_in, pitch_out, sf);
    else if (dtype == ELVIS_F32)
        hipLaunchKernelGGL(bicubic_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, (float*)y, n, h, w, c, pitch_in, pitch_out, sf);
    else
        ELVIS_REQUIRE(false, "elvis_bicubic_upsample: bad dtype");
    ELVIS_CHECK_LAUNCH("elvis_bicubic_upsample");
    return ELVIS_OK;
}

extern "C" int elvis_vq_nearest(const void* z, void* zq, int32_t* idx_out, int dtype, long long pixels, int c,
                                int pitch_in, int pitch_out, const float* codebook, int n_embed,
                                elvis_stream_t stream) {
    ELVIS_REQUIRE(z && zq && codebook && pixels > 0 && n_embed > 0, "elvis_vq_nearest: bad argument");
    ELVIS_REQUIRE(c >= 1 && c <= 4 && pitch_in >= c && pitch_out >= c, "elvis_vq_nearest: c must be 1..4");
    int grid = (int)((((pixels + 3) / 4) * 4 + 255) / 256);   // 4 lanes per group of 4 pixels
    if (dtype == ELVIS_F16)
        hipLaunchKernelGGL(vq_nearest_kernel<half_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const half_t*)z,
                           (half_t*)zq, idx_out, pixels, c, pitch_in, pitch_out, codebook, n_embed);
    else if (dtype == ELVIS_F32)
        hipLaunchKernelGGL(vq_nearest_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)z,
                           (float*)zq, idx_out, pixels, c, pitch_in, pitch_out, codebook, n_embed);
    else
        ELVIS_REQUIRE(false, "elvis_vq_nearest: bad dtype");
    ELVIS_CHECK_LAUNCH("elvis_vq_nearest");
    return ELVIS_OK;
}

extern "C" int elvis_pad_reflect_axpy(const void* x, void* y, int dtype, int n, int h, int w, int c, int pitch_in,
                                      int hp, int wp, int pitch_out, int ch_offset_out, float mul, const float* add,
                                      float add_mul, elvis_stream_t stream) {
    ELVIS_REQUIRE(x && y && n > 0 && h > 0 && w > 0 && c > 0 && hp >= h && wp >= w, "elvis_pad_reflect_axpy: bad argument");
    ELVIS_REQUIRE(hp - h < h && wp - w < w, "elvis_pad_reflect_axpy: reflect padding (%d,%d) must be smaller than the image (%d,%d)", hp - h, wp - w, h, w);
    ELVIS_REQUIRE(pitch_in >= c && ch_offset_out >= 0 && ch_offset_out + c <= pitch_out, "elvis_pad_reflect_axpy: bad pitch/offset");
    long long total = (long long)n * hp * wp;
    if (dtype == ELVIS_F16)
        hipLaunchKernelGGL(pad_reflect_axpy_kernel<half_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const half_t*)x, (half_t*)y, n, h, w, c, pitch_in, hp, wp, pitch_out, ch_offset_out, mul,
                           add, add_mul);
    else if (dtype == ELVIS_F32)
        hipLaunchKernelGGL(pad_reflect_axpy_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, (float*)y, n, h, w, c, pitch_in, hp, wp, pitch_out, ch_offset_out, mul,
                           add, add_mul);
    else
        ELVIS_REQUIRE(false, "elvis_pad_reflect_axpy: bad dtype");
    ELVIS_CHECK_LAUNCH("elvis_pad_reflect_axpy");
    return ELVIS_OK;
}

extern "C" int elvis_crop_copy(const void* x, void* y, int dtype, int n, int h_in, int w_in, int pitch_in, int h,
                               int w, int c, int pitch_out, elvis_stream_t stream) {
    ELVIS_REQUIRE(x && y && n > 0 && h > 0 && w > 0 && h <= h_in && w <= w_in && c > 0 && pitch_in >= c && pitch_out >= c,
                  "elvis_crop_copy: bad argument");
    long long total = (long long)n * h * w;
    if (dtype == ELVIS_F16)
        hipLaunchKernelGGL(crop_copy_kernel<half_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const half_t*)x, (half_t*)y, n, h_in, w_in, pitch_in, h, w, c, pitch_out);
    else if (dtype == ELVIS_F32)
        hipLaunchKernelGGL(crop_copy_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, (float*)y, n, h_in, w_in, pitch_in, h, w, c, pitch_out);
    else
        ELVIS_REQUIRE(false, "elvis_crop_copy: bad dtype");
    ELVIS_CHECK_LAUNCH("elvis_crop_copy");
    return ELVIS_OK;
}

extern "C" int elvis_convert_act(const void* x, int src_dtype, void* y, int dst_dtype, long long pixels, int pitch,
                                 elvis_stream_t stream) {
    ELVIS_REQUIRE(x && y && pixels > 0 && pitch > 0 && pitch % 8 == 0, "elvis_convert_act: bad argument");
    ELVIS_REQUIRE(((uintptr_t)x | (uintptr_t)y) % 16 == 0, "elvis_convert_act: pointers must be 16-byte aligned");
    const long long total8 = pixels * (pitch / 8);
    if (src_dtype == ELVIS_F16 && dst_dtype == ELVIS_F32)
        hipLaunchKernelGGL((convert_act_kernel<half_t, float>), dim3(grid_for(total8)), dim3(256), 0, (hipStream_t)stream,
                           (const half_t*)x, (float*)y, total8);
    else if (src_dtype == ELVIS_F32 && dst_dtype == ELVIS_F16)
        hipLaunchKernelGGL((convert_act_kernel<float, half_t>), dim3(grid_for(total8)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, (half_t*)y, total8);
    else
        ELVIS_REQUIRE(false, "elvis_convert_act: unsupported conversion %d -> %d", src_dtype, dst_dtype);
    ELVIS_CHECK_LAUNCH("elvis_convert_act");
    return ELVIS_OK;
}
