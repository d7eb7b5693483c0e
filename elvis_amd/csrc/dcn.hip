// DCNv2 (modulated deformable 3x3 convolution) and the temporal-window helpers of the
// LaplacianVCAR-style restorer (ELVIS v2 DCT slot; README.md:11-16 mentions only the upstream
// `ops/dcn` CUDA build, which is not in the reference - SURVEY.md F1/F2).
//
//   out[p][co] = b[co] + sum_{g,k} sum_{c in group g} W[co][c][k] * m[g][k][p] * bilinear(x[c], p + tap_k + d[g][k][p])
//
// with offsets laid out channel = (g*9 + k)*2 + {0: dy, 1: dx}, masks channel = g*9 + k (already
// passed through the sigmoid by the caller's choice: `mask_sigmoid`), zero contribution from
// out-of-image corners.  Phase 1 gathers the K = cin*9 modulated samples of 64 pixels into an LDS
// im2col tile (irregular, L1/L2-served 2-byte taps) - the HBM/gather-bound part; phase 2 is the
// small [64 px] x [K = cin*9] x [cout] product in fp32 FMAs (K = 63 for the 7-frame restorer: too
// thin to be worth an MFMA pipeline; moving it to the matrix cores is listed as next in DESIGN.md).
#include "common.h"

namespace {

constexpr int DPX = 64;   // pixels per workgroup

template <typename T>
__global__ __launch_bounds__(256) void dcnv2_kernel(const T* __restrict__ x, const T* __restrict__ om,
                                                    const T* __restrict__ wt, const float* __restrict__ bias,
                                                    T* __restrict__ out, int n, int h, int w, int cin, int x_pitch,
                                                    int dg, int om_pitch, int mask_off, int mask_sigmoid, int cout,
                                                    int out_pitch, int act, int kpad) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* col = reinterpret_cast<float*>(smem_raw);   // [DPX][kpad + 1] modulated samples (f32)
    const int K = cin * 9;
    const long long total = (long long)n * h * w;
    const long long p0 = (long long)blockIdx.x * DPX;
    const int tid = threadIdx.x;
    const int cpg = cin / dg;
    const int colp = kpad + 1;

    // ---- phase 1: thread t handles pixel t & 63 and the k values (t >> 6), (t >> 6) + 4, ...
    {
        const int pl = tid & (DPX - 1);
        const long long pix = p0 + pl;
        const bool pok = pix < total;
        long long pp = pok ? pix : 0;
        const int xo = (int)(pp % w);
        const long long r = pp / w;
        const int yo = (int)(r % h);
        const int ni = (int)(r / h);
        const T* omp = om + pp * om_pitch;
        for (int kk = tid >> 6; kk < kpad; kk += 4) {
            float v = 0.f;
            if (pok && kk < K) {
                const int c = kk / 9, k = kk - c * 9;
                const int g = c / cpg;
                const float dy = to_f(omp[(g * 9 + k) * 2]), dx = to_f(omp[(g * 9 + k) * 2 + 1]);
                float m = to_f(omp[mask_off + g * 9 + k]);
                if (mask_sigmoid) m = 1.0f / (1.0f + expf(-m));
                const float sy = (float)(yo + k / 3 - 1) + dy, sx = (float)(xo + k % 3 - 1) + dx;
                const float fy = floorf(sy), fx = floorf(sx);
                const int y0 = (int)fy, x0 = (int)fx;
                const float ly = sy - fy, lx = sx - fx;
                const T* xb = x + ((long long)ni * h) * w * x_pitch + c;
                auto tap = [&](int yy, int xx) -> float {
                    return (yy >= 0 && yy < h && xx >= 0 && xx < w) ? to_f(xb[((long long)yy * w + xx) * x_pitch]) : 0.f;
                };
                float v00 = tap(y0, x0), v01 = tap(y0, x0 + 1), v10 = tap(y0 + 1, x0), v11 = tap(y0 + 1, x0 + 1);
                v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
                v *= m;
            }
            col[pl * colp + kk] = v;
        }
    }
    __syncthreads();
    // ---- phase 2: out[px][co] = col[px][:] . W[co][:]   (fp32 FMAs; K <= a few hundred)
    for (int o = tid; o < DPX * cout; o += 256) {
        const int pl = o / cout, co = o - pl * cout;
        const long long pix = p0 + pl;
        if (pix >= total) continue;
        float acc = bias ? bias[co] : 0.f;
        const T* wr = wt + (long long)co * K;
        const float* cr = col + pl * colp;
        for (int kk = 0; kk < K; ++kk) acc = fmaf(cr[kk], to_f(wr[kk]), acc);
        if (act == 3) acc = fmaxf(acc, 0.f);
        out[pix * out_pitch + co] = from_f<T>(acc);
    }
}

// frames u8 [F,H,W,3] -> float planes [(f*3 + c), H, W, pitch]: channel t of plane (f,c) is colour c
// of frame clamp(f + t - radius, 0, F-1), scaled to [0,1].
template <typename T>
__global__ __launch_bounds__(256) void temporal_stack_kernel(const uint8_t* __restrict__ frames, T* __restrict__ out,
                                                             int nf, int f0, int nsel, int h, int w, int radius,
                                                             int pitch) {
    const long long total = (long long)nsel * 3 * h * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long hw = (long long)h * w;
        const long long plane = i / hw, px = i - plane * hw;
        const int fi = (int)(plane / 3) + f0, c = (int)(plane % 3);
        T* d = out + i * pitch;
        const int T7 = 2 * radius + 1;
        for (int t = 0; t < T7; ++t) {
            int fs = fi + t - radius;
            fs = fs < 0 ? 0 : (fs > nf - 1 ? nf - 1 : fs);
            d[t] = from_f<T>(__fdiv_rn((float)frames[((long long)fs * hw + px) * 3 + c], 255.0f));
        }
        for (int t = T7; t < pitch; ++t) d[t] = from_f<T>(0.f);
    }
}

// out_u8[f,h,w,c] = round(clip(center + residual, 0, 1) * 255), center = frames[f][c]/255,
// residual = planes [(f*3+c), H, W, pitch] channel 0.
template <typename T>
__global__ __launch_bounds__(256) void plane_merge_kernel(const uint8_t* __restrict__ frames, const T* __restrict__ res,
                                                          uint8_t* __restrict__ out, int f0, int nsel, int h, int w,
                                                          int pitch) {
    const long long hw = (long long)h * w;
    const long long total = (long long)nsel * hw * 3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        const long long px = (i / 3) % hw;
        const long long fl = i / (3 * hw);
        float center = __fdiv_rn((float)frames[((long long)(fl + f0) * hw + px) * 3 + c], 255.0f);
        float v = __fadd_rn(center, to_f(res[((fl * 3 + c) * hw + px) * pitch]));
        v = fminf(fmaxf(v, 0.f), 1.f);
        out[i] = (uint8_t)__float2int_rn(__fmul_rn(v, 255.0f));
    }
}

}  // namespace

extern "C" int elvis_dcnv2(const void* x, const void* offset_mask, const void* weight, const float* bias, void* out,
                           int dtype, int n, int h, int w, int cin, int x_pitch, int deformable_groups, int om_pitch,
                           int mask_sigmoid, int cout, int out_pitch, int act, elvis_stream_t stream) {
    ELVIS_REQUIRE(x && offset_mask && weight && out, "elvis_dcnv2: null pointer");
    ELVIS_REQUIRE(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && deformable_groups > 0 && cin % deformable_groups == 0,
                  "elvis_dcnv2: cin %d not divisible by deformable_groups %d", cin, deformable_groups);
    ELVIS_REQUIRE(x_pitch >= cin && om_pitch >= deformable_groups * 27 && out_pitch >= cout, "elvis_dcnv2: bad pitch");
    ELVIS_REQUIRE(act == 0 || act == 3, "elvis_dcnv2: act must be 0 or 3 (ReLU)");
    const int K = cin * 9;
    const int kpad = (K + 3) / 4 * 4;
    size_t lds = (size_t)DPX * (kpad + 1) * sizeof(float);
    ELVIS_REQUIRE(lds <= 64 * 1024, "elvis_dcnv2: cin*9 = %d too large for the LDS im2col tile", K);
    const long long total = (long long)n * h * w;
    const unsigned grid = (unsigned)((total + DPX - 1) / DPX);
    const int mask_off = deformable_groups * 18;
    if (dtype == ELVIS_F16)
        hipLaunchKernelGGL(dcnv2_kernel<half_t>, dim3(grid), dim3(256), lds, (hipStream_t)stream, (const half_t*)x,
                           (const half_t*)offset_mask, (const half_t*)weight, bias, (half_t*)out, n, h, w, cin, x_pitch,
                           deformable_groups, om_pitch, mask_off, mask_sigmoid, cout, out_pitch, act, kpad);
    else if (dtype == ELVIS_F32)
        hipLaunchKernelGGL(dcnv2_kernel<float>, dim3(grid), dim3(256), lds, (hipStream_t)stream, (const float*)x,
                           (const float*)offset_mask, (const float*)weight, bias, (float*)out, n, h, w, cin, x_pitch,
                           deformable_groups, om_pitch, mask_off, mask_sigmoid, cout, out_pitch, act, kpad);
    else
        ELVIS_REQUIRE(false, "elvis_dcnv2: bad dtype");
    ELVIS_CHECK_LAUNCH("elvis_dcnv2");
    return ELVIS_OK;
}

extern "C" int elvis_temporal_stack(const uint8_t* frames, void* out, int dtype, int nf, int f0, int nsel, int h, int w,
                                    int radius, int pitch, elvis_stream_t stream) {
    ELVIS_REQUIRE(frames && out && nf > 0 && nsel > 0 && f0 >= 0 && f0 + nsel <= nf && h > 0 && w > 0 && radius >= 0 &&
                      pitch >= 2 * radius + 1,
                  "elvis_temporal_stack: bad argument");
    long long total = (long long)nsel * 3 * h * w;
    int grid = (int)((total + 255) / 256);
    if (grid > 256 * 32) grid = 256 * 32;
    if (dtype == ELVIS_F16)
        hipLaunchKernelGGL(temporal_stack_kernel<half_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, frames,
                           (half_t*)out, nf, f0, nsel, h, w, radius, pitch);
    else if (dtype == ELVIS_F32)
        hipLaunchKernelGGL(temporal_stack_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, frames,
                           (float*)out, nf, f0, nsel, h, w, radius, pitch);
    else
        ELVIS_REQUIRE(false, "elvis_temporal_stack: bad dtype");
    ELVIS_CHECK_LAUNCH("elvis_temporal_stack");
    return ELVIS_OK;
}

extern "C" int elvis_plane_merge(const uint8_t* frames, const void* residual, uint8_t* out, int dtype, int f0, int nsel,
                                 int h, int w, int pitch, elvis_stream_t stream) {
    ELVIS_REQUIRE(frames && residual && out && nsel > 0 && f0 >= 0 && h > 0 && w > 0 && pitch >= 1, "elvis_plane_merge: bad argument");
    long long total = (long long)nsel * 3 * h * w;
    int grid = (int)((total + 255) / 256);
    if (grid > 256 * 32) grid = 256 * 32;
    if (dtype == ELVIS_F16)
        hipLaunchKernelGGL(plane_merge_kernel<half_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, frames,
                           (const half_t*)residual, out, f0, nsel, h, w, pitch);
    else if (dtype == ELVIS_F32)
        hipLaunchKernelGGL(plane_merge_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, frames,
                           (const float*)residual, out, f0, nsel, h, w, pitch);
    else
        ELVIS_REQUIRE(false, "elvis_plane_merge: bad dtype");
    ELVIS_CHECK_LAUNCH("elvis_plane_merge");
    return ELVIS_OK;
}
