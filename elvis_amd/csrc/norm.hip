// GroupNorm (statistics -> per-(n,channel) affine -> fused apply+SiLU) and LayerNorm.
// HBM-bound: 16-byte vector loads, wavefront-shuffle reductions, fp32 math, fp64 cross-workgroup
// accumulation (one atomic per channel per workgroup) so results are reproducible to fp32 rounding.
#include "common.h"

namespace {

template <typename T> struct Vec16 {
    uint4 raw;
    __device__ __forceinline__ float get(int i) const { return to_f(reinterpret_cast<const T*>(&raw)[i]); }
    __device__ __forceinline__ void set(int i, float v) { reinterpret_cast<T*>(&raw)[i] = from_f<T>(v); }
};

// ---- per-channel sums: grid = (pixel slabs, n).  Thread t owns channel vector t % CV for the
// pixels  plane + k*PL  of its slab (PL = 256 / CV pixel lanes; the remaining threads idle).
template <typename T>
__global__ __launch_bounds__(256) void gn_channel_sums_kernel(const T* __restrict__ x, int hw, int c, int pitch,
                                                              int px_per_block, float* __restrict__ partials) {
    constexpr int VEC = DT<T>::VEC;
    extern __shared__ float red[];  // [PL][c][2]
    const int cv = (c + VEC - 1) / VEC;
    const int pl_count = 256 / cv;
    const int tid = threadIdx.x;
    const int vec = tid % cv, plane = tid / cv;
    const int n = blockIdx.y;
    const long long p0 = (long long)blockIdx.x * px_per_block;
    long long p1 = p0 + px_per_block;
    if (p1 > hw) p1 = hw;
    float s[VEC], ss[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s[i] = ss[i] = 0.f;
    if (plane < pl_count) {
        const T* base = x + ((long long)n * hw) * pitch + vec * VEC;
        for (long long p = p0 + plane; p < p1; p += pl_count) {
            Vec16<T> v;
            v.raw = *reinterpret_cast<const uint4*>(base + p * pitch);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float f = v.get(i);
                s[i] += f;
                ss[i] = fmaf(f, f, ss[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            int ch = vec * VEC + i;
            if (ch < c) {
                red[(plane * c + ch) * 2 + 0] = s[i];
                red[(plane * c + ch) * 2 + 1] = ss[i];
            }
        }
    }
    __syncthreads();
    for (int ch = tid; ch < c; ch += 256) {
        float a = 0.f, b = 0.f;
        for (int pl = 0; pl < pl_count; ++pl) {
            a += red[(pl * c + ch) * 2 + 0];
            b += red[(pl * c + ch) * 2 + 1];
        }
        // one partial row per workgroup, reduced afterwards in a fixed order (bit-reproducible)
        float* dst = partials + (((long long)n * gridDim.x + blockIdx.x) * c + ch) * 2;
        dst[0] = a;
        dst[1] = b;
    }
}

// sums[n, coff + c] = sum over the image's tiles of partials[tile][c][0..1] (fp64 accumulation,
// fixed order -> bit-reproducible).  grid = (c, n), one workgroup per channel.
__global__ __launch_bounds__(256) void gn_partials_reduce_kernel(const float* __restrict__ partials,
                                                                 int tiles_per_image, int c,
                                                                 double* __restrict__ sums, int ctot, int coff) {
    __shared__ double ra[256], rb[256];
    const int ch = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    double a = 0, b = 0;
    const float* base = partials + ((long long)n * tiles_per_image * c + ch) * 2;
    for (int t = tid; t < tiles_per_image; t += 256) {
        a += (double)base[(long long)t * c * 2 + 0];
        b += (double)base[(long long)t * c * 2 + 1];
    }
    ra[tid] = a;
    rb[tid] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            ra[tid] += ra[tid + o];
            rb[tid] += rb[tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {
        sums[((long long)n * ctot + coff + ch) * 2 + 0] = ra[0];
        sums[((long long)n * ctot + coff + ch) * 2 + 1] = rb[0];
    }
}

// Same reduction, four adjacent channels per workgroup: every thread reads 32 contiguous bytes per tile
// (whole sectors instead of 8 of every 32 bytes).  Per channel the accumulation order is exactly that of
// gn_partials_reduce_kernel, so the sums are bit-identical.  grid = (c/4, n), c % 4 == 0.
__global__ __launch_bounds__(256) void gn_partials_reduce4_kernel(const float* __restrict__ partials,
                                                                  int tiles_per_image, int c,
                                                                  double* __restrict__ sums, int ctot, int coff) {
    __shared__ double red[8][256];
    const int ch = blockIdx.x * 4, n = blockIdx.y, tid = threadIdx.x;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float* base = partials + ((long long)n * tiles_per_image * c + ch) * 2;
    for (int t = tid; t < tiles_per_image; t += 256) {
        const float4* p4 = reinterpret_cast<const float4*>(base + (long long)t * c * 2);
        const float4 u = p4[0], v = p4[1];
        acc[0] += (double)u.x; acc[1] += (double)u.y; acc[2] += (double)u.z; acc[3] += (double)u.w;
        acc[4] += (double)v.x; acc[5] += (double)v.y; acc[6] += (double)v.z; acc[7] += (double)v.w;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[k][tid] = acc[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
#pragma unroll
            for (int k = 0; k < 8; ++k) red[k][tid] += red[k][tid + o];
        }
        __syncthreads();
    }
    if (tid < 8) sums[((long long)n * ctot + coff + ch) * 2 + tid] = red[tid][0];
}

__global__ void gn_affine_kernel(const double* __restrict__ sums, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, const float* __restrict__ scale,
                                 const float* __restrict__ shift, float* __restrict__ pa, float* __restrict__ pb,
                                 int n, int hw, int c, int groups, float eps) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * c) return;
    int ni = i / c, ch = i - ni * c;
    int cpg = c / groups;
    int g = ch / cpg;
    double s = 0, ss = 0;
    for (int k = 0; k < cpg; ++k) {
        s += sums[((long long)ni * c + g * cpg + k) * 2 + 0];
        ss += sums[((long long)ni * c + g * cpg + k) * 2 + 1];
    }
    double cnt = (double)hw * cpg;
    double mean = s / cnt;
    double var = ss / cnt - mean * mean;
    if (var < 0) var = 0;
    float rstd = (float)(1.0 / sqrt(var + (double)eps));
    float ga = gamma ? gamma[ch] : 1.f, be = beta ? beta[ch] : 0.f;
    float a = rstd * ga;
    float b = be - (float)mean * a;
    if (scale) {
        float sc = 1.f + scale[ch];
        a *= sc;
        b = b * sc + (shift ? shift[ch] : 0.f);
    } else if (shift) {
        b += shift[ch];
    }
    pa[i] = a;
    pb[i] = b;
}

template <typename T, bool PRECISE>
__global__ __launch_bounds__(256) void affine_act_kernel(const T* __restrict__ x, T* __restrict__ y, int hw, int c,
                                                         int pitch_in, int pitch_out, const float* __restrict__ pa,
                                                         const float* __restrict__ pb, int act, long long total_vec) {
    constexpr int VEC = DT<T>::VEC;
    const int cv = (c + VEC - 1) / VEC;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total_vec;
         i += (long long)gridDim.x * blockDim.x) {
        long long px = i / cv;
        int vec = (int)(i - px * cv);
        int n = (int)(px / hw);
        Vec16<T> v, o;
        v.raw = *reinterpret_cast<const uint4*>(x + px * pitch_in + vec * VEC);
        const float* a = pa + (long long)n * c + vec * VEC;
        const float* b = pb + (long long)n * c + vec * VEC;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            int ch = vec * VEC + k;
            float t = 0.f;
            if (ch < c) {
                t = fmaf(v.get(k), a[k], b[k]);
                if (act == 2) t = PRECISE ? t / (1.0f + expf(-t)) : t / (1.0f + __expf(-t));
            }
            o.set(k, t);
        }
        *reinterpret_cast<uint4*>(y + px * pitch_out + vec * VEC) = o.raw;
    }
}

// ---- LayerNorm: LPT lanes per token (power of two >= c/VEC), 64/LPT tokens per wave.
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, T* __restrict__ y, long long tokens,
                                                        int c, int pitch_in, int pitch_out,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, int lpt) {
    constexpr int VEC = DT<T>::VEC;
    const int tpw = 64 / lpt;                       // tokens per wave
    const int lane = threadIdx.x & 63;
    const int sub = lane / lpt, l = lane % lpt;
    const long long wave_global = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int cv = c / VEC;
    for (long long t0 = wave_global * tpw; t0 < tokens; t0 += nwaves * tpw) {
        long long tok = t0 + sub;
        bool active = tok < tokens && l < cv;
        Vec16<T> v;
        v.raw = make_uint4(0, 0, 0, 0);
        if (active) v.raw = *reinterpret_cast<const uint4*>(x + tok * pitch_in + l * VEC);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < VEC; ++k) s += v.get(k);
        for (int o = lpt >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        float mean = s / (float)c;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float d = active ? v.get(k) - mean : 0.f;
            q = fmaf(d, d, q);
        }
        for (int o = lpt >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        float rstd = 1.0f / sqrtf(q / (float)c + eps);
        if (active) {
            Vec16<T> o;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                int ch = l * VEC + k;
                o.set(k, (v.get(k) - mean) * rstd * gamma[ch] + beta[ch]);
            }
            *reinterpret_cast<uint4*>(y + tok * pitch_out + l * VEC) = o.raw;
        }
    }
}

}  // namespace

static int gn_blocks(int dtype, int hw, int c, int* px_per_block_out) {
    int vec = dtype == ELVIS_F16 ? 8 : 4;
    int cv = (c + vec - 1) / vec;
    int pl = 256 / (cv > 0 ? cv : 1);
    if (pl < 1) pl = 1;
    // ~2048 workgroups over the pixels, at least 16 pixels per lane-plane each
    int px_per_block = (int)((hw + 2047) / 2048);
    if (px_per_block < pl * 16) px_per_block = pl * 16;
    if (px_per_block_out) *px_per_block_out = px_per_block;
    return (hw + px_per_block - 1) / px_per_block;
}

extern "C" size_t elvis_groupnorm_workspace_floats(int dtype, int n, int hw, int c) {
    if (n <= 0 || hw <= 0 || c <= 0) return 0;
    return (size_t)n * gn_blocks(dtype, hw, c, nullptr) * c * 2;
}

extern "C" int elvis_groupnorm_sums(const void* x, int dtype, int n, int hw, int c, int pitch, double* sums,
                                    int sums_ctot, int sums_coff, float* workspace, elvis_stream_t stream) {
    ELVIS_REQUIRE(x && sums && workspace, "elvis_groupnorm_sums: null pointer");
    ELVIS_REQUIRE(n > 0 && hw > 0 && c > 0 && pitch >= c && pitch % 8 == 0, "elvis_groupnorm_sums: bad shape c=%d pitch=%d", c, pitch);
    int vec = dtype == ELVIS_F16 ? 8 : 4;
    int cv = (c + vec - 1) / vec;
    ELVIS_REQUIRE(cv <= 256, "elvis_groupnorm_sums: too many channels (%d)", c);
    ELVIS_REQUIRE(sums_coff >= 0 && sums_coff + c <= sums_ctot, "elvis_groupnorm_sums: channel slice [%d,%d) outside %d", sums_coff, sums_coff + c, sums_ctot);
    int pl = 256 / cv;
    int px_per_block = 0;
    int gx = gn_blocks(dtype, hw, c, &px_per_block);
    size_t lds = (size_t)pl * c * 2 * sizeof(float);
    ELVIS_REQUIRE(lds <= 64 * 1024, "elvis_groupnorm_sums: LDS budget exceeded");
    if (dtype == ELVIS_F16)
        hipLaunchKernelGGL(gn_channel_sums_kernel<half_t>, dim3(gx, n), dim3(256), lds, (hipStream_t)stream,
                           (const half_t*)x, hw, c, pitch, px_per_block, workspace);
    else if (dtype == ELVIS_F32)
        hipLaunchKernelGGL(gn_channel_sums_kernel<float>, dim3(gx, n), dim3(256), lds, (hipStream_t)stream,
                           (const float*)x, hw, c, pitch, px_per_block, workspace);
    else
        ELVIS_REQUIRE(false, "elvis_groupnorm_sums: bad dtype");
    ELVIS_CHECK_LAUNCH("elvis_groupnorm_sums");
    hipLaunchKernelGGL(gn_partials_reduce_kernel, dim3(c, n), dim3(256), 0, (hipStream_t)stream, workspace, gx, c, sums,
                       sums_ctot, sums_coff);
    ELVIS_CHECK_LAUNCH("elvis_groupnorm_sums(reduce)");
    return ELVIS_OK;
}

extern "C" int elvis_gn_partials_to_sums(const float* partials, int tiles_per_image, int n, int c, double* sums,
                                         int sums_ctot, int sums_coff, elvis_stream_t stream) {
    ELVIS_REQUIRE(partials && sums && tiles_per_image > 0 && n > 0 && c > 0, "elvis_gn_partials_to_sums: bad argument");
    ELVIS_REQUIRE(sums_coff >= 0 && sums_coff + c <= sums_ctot, "elvis_gn_partials_to_sums: channel slice outside buffer");
    if (c % 4 == 0 && ((uintptr_t)partials & 15) == 0)
        hipLaunchKernelGGL(gn_partials_reduce4_kernel, dim3(c / 4, n), dim3(256), 0, (hipStream_t)stream, partials,
                           tiles_per_image, c, sums, sums_ctot, sums_coff);
    else
        hipLaunchKernelGGL(gn_partials_reduce_kernel, dim3(c, n), dim3(256), 0, (hipStream_t)stream, partials,
                           tiles_per_image, c, sums, sums_ctot, sums_coff);
    ELVIS_CHECK_LAUNCH("elvis_gn_partials_to_sums");
    return ELVIS_OK;
}

extern "C" int elvis_groupnorm_affine(const double* sums, const float* gamma, const float* beta, const float* scale,
                                      const float* shift, float* pa, float* pb, int n, int hw, int c, int groups,
                                      float eps, elvis_stream_t stream) {
    ELVIS_REQUIRE(sums && pa && pb, "elvis_groupnorm_affine: null pointer");
    ELVIS_REQUIRE(n > 0 && hw > 0 && c > 0 && groups > 0 && c % groups == 0, "elvis_groupnorm_affine: c=%d not divisible by groups=%d", c, groups);
    int total = n * c;
    hipLaunchKernelGGL(gn_affine_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, gamma,
                       beta, scale, shift, pa, pb, n, hw, c, groups, eps);
    ELVIS_CHECK_LAUNCH("elvis_groupnorm_affine");
    return ELVIS_OK;
}

extern "C" int elvis_affine_act(const void* x, void* y, int dtype, int n, int hw, int c, int pitch_in, int pitch_out,
                                const float* pa, const float* pb, int act, elvis_stream_t stream) {
    ELVIS_REQUIRE(x && y && pa && pb, "elvis_affine_act: null pointer");
    ELVIS_REQUIRE(n > 0 && hw > 0 && c > 0 && pitch_in >= c && pitch_out >= c && pitch_in % 8 == 0 && pitch_out % 8 == 0,
                  "elvis_affine_act: bad shape");
    ELVIS_REQUIRE(act == 0 || act == 2, "elvis_affine_act: act must be 0 or 2");
    int vec = dtype == ELVIS_F16 ? 8 : 4;
    long long total_vec = (long long)n * hw * ((c + vec - 1) / vec);
    int grid = (int)((total_vec + 255) / 256);
    if (grid > 256 * 32) grid = 256 * 32;
    if (dtype == ELVIS_F16)
        hipLaunchKernelGGL((affine_act_kernel<half_t, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const half_t*)x, (half_t*)y, hw, c, pitch_in, pitch_out, pa, pb, act, total_vec);
    else if (dtype == ELVIS_F32)
        hipLaunchKernelGGL((affine_act_kernel<float, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, (float*)y, hw, c, pitch_in, pitch_out, pa, pb, act, total_vec);
    else
        ELVIS_REQUIRE(false, "elvis_affine_act: bad dtype");
    ELVIS_CHECK_LAUNCH("elvis_affine_act");
    return ELVIS_OK;
}

extern "C" int elvis_layernorm(const void* x, void* y, int dtype, long long tokens, int c, int pitch_in,
                               int pitch_out, const float* gamma, const float* beta, float eps,
                               elvis_stream_t stream) {
    ELVIS_REQUIRE(x && y && gamma && beta, "elvis_layernorm: null pointer");
    int vec = dtype == ELVIS_F16 ? 8 : 4;
    ELVIS_REQUIRE(tokens > 0 && c > 0 && c % vec == 0 && c / vec <= 64, "elvis_layernorm: c=%d unsupported", c);
    ELVIS_REQUIRE(pitch_in >= c && pitch_out >= c && pitch_in % 8 == 0 && pitch_out % 8 == 0, "elvis_layernorm: bad pitch");
    int lpt = 1;
    while (lpt < c / vec) lpt <<= 1;
    int tpw = 64 / lpt;
    long long waves = (tokens + tpw - 1) / tpw;
    long long blocks = (waves + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (dtype == ELVIS_F16)
        hipLaunchKernelGGL(layernorm_kernel<half_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           (const half_t*)x, (half_t*)y, tokens, c, pitch_in, pitch_out, gamma, beta, eps, lpt);
    else if (dtype == ELVIS_F32)
        hipLaunchKernelGGL(layernorm_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, (float*)y, tokens, c, pitch_in, pitch_out, gamma, beta, eps, lpt);
    else
        ELVIS_REQUIRE(false, "elvis_layernorm: bad dtype");
    ELVIS_CHECK_LAUNCH("elvis_layernorm");
    return ELVIS_OK;
}
