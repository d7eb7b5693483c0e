// Server-side per-block degrade filters on the device: what produces the hot path's
// inputs.  Each b x b block of a uint8 NHWC frame is filtered as its own little image, so nothing leaks
// between blocks:
//   elvis_degrade_downsample_u8  - filter_frame_downsample (elvis.py:2141-2169): box-downscale the block by
//                                  2**level (INTER_AREA), bilinear back to b x b (INTER_LINEAR)
//   elvis_degrade_gaussian_u8    - filter_frame_gaussian (elvis.py:2171-2196): `rounds` x GaussianBlur 5x5,
//                                  sigma 1, BORDER_REFLECT_101 at the block's own edges
//   elvis_degrade_dct_u8         - the build's definition of "DCT coefficient dampening" (README.md:44; the
//                                  reference has no code for it): 8x8 DCT-II per block and channel, coefficient
//                                  (u,v) scaled by 2^(-level*(u+v)/14), inverse DCT
// OpenCV is absent from the build and GPU environments, so its u8 rounding rules are restated from its
// documented fixed-point arithmetic ("parity unpinned"); what is pinned is bit-exactness against
// oracle/degrade_ref.py, which restates the same rules in numpy.  Evaluation order of every float expression
// is fixed (explicit round-to-nearest intrinsics; the library is built with -ffp-contract=off).
#include "common.h"

namespace {

// cv2.resize(INTER_LINEAR) source index and 11-bit weights of destination index d (s source samples, b results)
__device__ __forceinline__ void linear_coef(int d, int s, int b, int& i0, int& a0, int& a1) {
    float f = (float)(((double)d + 0.5) * ((double)s / (double)b) - 0.5);   // double arithmetic, one rounding to float
    int i = (int)floorf(f);
    f = __fsub_rn(f, (float)i);
    if (i < 0) { i = 0; f = 0.f; }
    if (i >= s - 1) { i = s - 1; f = 0.f; }
    i0 = i;
    a0 = __float2int_rn(__fmul_rn(__fsub_rn(1.0f, f), 2048.0f));
    a1 = __float2int_rn(__fmul_rn(f, 2048.0f));
}

// one thread per (block, channel): the block is at most 16 x 16 (b <= 16)
__global__ __launch_bounds__(64) void degrade_downsample_kernel(const uint8_t* __restrict__ src, const int32_t* __restrict__ levels,
                                                                uint8_t* __restrict__ dst, int n, int h, int w, int c, int b,
                                                                int by, int bx, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ch = (int)(i % c);
    long long t = i / c;
    const int bxi = (int)(t % bx); t /= bx;
    const int byi = (int)(t % by);
    const int f = (int)(t / by);
    const long long base = (((long long)f * h + (long long)byi * b) * w + (long long)bxi * b) * c + ch;
    const long long rs = (long long)w * c;
    const int lv = levels[((long long)f * by + byi) * bx + bxi];
    int fac = 1 << (lv < 0 ? 0 : (lv > 4 ? 4 : lv));
    if (fac <= 1) {
        for (int y = 0; y < b; ++y)
            for (int x = 0; x < b; ++x) dst[base + y * rs + (long long)x * c] = src[base + y * rs + (long long)x * c];
        return;
    }
    int s = b / fac;
    if (s < 1) { s = 1; fac = b; }
    uint8_t small[8][8];   // s <= b/2 <= 8
    const int area = fac * fac;
    const float inv = 1.0f / (float)area;
    for (int y = 0; y < s; ++y)
        for (int x = 0; x < s; ++x) {
            uint32_t sum = 0;
            for (int dy = 0; dy < fac; ++dy)
                for (int dx = 0; dx < fac; ++dx) sum += src[base + (long long)(y * fac + dy) * rs + (long long)(x * fac + dx) * c];
            uint32_t v = fac == 2 ? (sum + 2) >> 2 : (uint32_t)__float2int_rn(__fmul_rn((float)sum, inv));   // INTER_AREA u8 rules
            small[y][x] = (uint8_t)(v > 255 ? 255 : v);
        }
    // INTER_LINEAR back to b x b: horizontal pass in 11-bit fixed point, vertical pass (b0*(S0>>4)>>16 + b1*(S1>>4)>>16 + 2)>>2
    for (int y = 0; y < b; ++y) {
        int y0, b0, b1;
        linear_coef(y, s, b, y0, b0, b1);
        const int y1 = y0 + 1 < s ? y0 + 1 : y0;
        for (int x = 0; x < b; ++x) {
            int x0, a0, a1;
            linear_coef(x, s, b, x0, a0, a1);
            const int x1 = x0 + 1 < s ? x0 + 1 : x0;
            const int r0 = small[y0][x0] * a0 + small[y0][x1] * a1;
            const int r1 = small[y1][x0] * a0 + small[y1][x1] * a1;
            int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            dst[base + y * rs + (long long)x * c] = (uint8_t)v;
        }
    }
}

__device__ __forceinline__ int reflect101(int i, int n) {   // BORDER_REFLECT_101: -1 -> 1, n -> n-2 (n == 1: 0)
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

// one thread per (block, channel); b <= 16.  Per round: float32 horizontal pass, float32 vertical pass,
// round-half-even to uint8 (the kernel taps are passed in, computed once on the host: getGaussianKernel(5, 1)).
__global__ __launch_bounds__(64) void degrade_gaussian_kernel(const uint8_t* __restrict__ src, const int32_t* __restrict__ rounds,
                                                              uint8_t* __restrict__ dst, int n, int h, int w, int c, int b,
                                                              int by, int bx, float k0, float k1, float k2, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ch = (int)(i % c);
    long long t = i / c;
    const int bxi = (int)(t % bx); t /= bx;
    const int byi = (int)(t % by);
    const int f = (int)(t / by);
    const long long base = (((long long)f * h + (long long)byi * b) * w + (long long)bxi * b) * c + ch;
    const long long rs = (long long)w * c;
    int r = rounds[((long long)f * by + byi) * bx + bxi];
    r = r < 0 ? 0 : (r > 32 ? 32 : r);
    uint8_t cur[16][16];
    float tmp[16][16];
    for (int y = 0; y < b; ++y)
        for (int x = 0; x < b; ++x) cur[y][x] = src[base + y * rs + (long long)x * c];
    const float kk[5] = {k0, k1, k2, k1, k0};
    for (int it = 0; it < r; ++it) {
        for (int y = 0; y < b; ++y)
            for (int x = 0; x < b; ++x) {
                float a = 0.f;
                for (int d = 0; d < 5; ++d) a = __fadd_rn(a, __fmul_rn(kk[d], (float)cur[y][reflect101(x + d - 2, b)]));
                tmp[y][x] = a;
            }
        for (int y = 0; y < b; ++y)
            for (int x = 0; x < b; ++x) {
                float a = 0.f;
                for (int d = 0; d < 5; ++d) a = __fadd_rn(a, __fmul_rn(kk[d], tmp[reflect101(y + d - 2, b)][x]));
                int v = __float2int_rn(a);
                cur[y][x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
    }
    for (int y = 0; y < b; ++y)
        for (int x = 0; x < b; ++x) dst[base + y * rs + (long long)x * c] = cur[y][x];
}

// one thread per (8x8 block, channel).  basis[u][x] = C(u) cos((2x+1) u pi / 16) (f32, from the host);
// gain[level][u][v] = 2^(-level (u+v) / 14) (f32, from the host; [*][0][0] = 1).
__global__ __launch_bounds__(64) void degrade_dct_kernel(const uint8_t* __restrict__ src, const int32_t* __restrict__ levels,
                                                         uint8_t* __restrict__ dst, const float* __restrict__ basis,
                                                         const float* __restrict__ gain, int n_levels, int n, int h, int w, int c,
                                                         int by, int bx, long long total) {
    __shared__ float sb[64];
    if (threadIdx.x < 64) sb[threadIdx.x] = basis[threadIdx.x];
    __syncthreads();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ch = (int)(i % c);
    long long t = i / c;
    const int bxi = (int)(t % bx); t /= bx;
    const int byi = (int)(t % by);
    const int f = (int)(t / by);
    const long long base = (((long long)f * h + (long long)byi * 8) * w + (long long)bxi * 8) * c + ch;
    const long long rs = (long long)w * c;
    int lv = levels[((long long)f * by + byi) * bx + bxi];
    lv = lv < 0 ? 0 : (lv >= n_levels ? n_levels - 1 : lv);
    float X[8][8], Y[8][8];
    for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) X[y][x] = (float)src[base + y * rs + (long long)x * c];
    if (lv == 0) {
        for (int y = 0; y < 8; ++y)
            for (int x = 0; x < 8; ++x) dst[base + y * rs + (long long)x * c] = (uint8_t)X[y][x];
        return;
    }
    // Y = B X B^T (rows then columns), k ascending, plain float32 multiply-then-add
    for (int u = 0; u < 8; ++u)
        for (int x = 0; x < 8; ++x) {
            float a = 0.f;
            for (int k = 0; k < 8; ++k) a = __fadd_rn(a, __fmul_rn(sb[u * 8 + k], X[k][x]));
            Y[u][x] = a;
        }
    for (int u = 0; u < 8; ++u)
        for (int v = 0; v < 8; ++v) {
            float a = 0.f;
            for (int k = 0; k < 8; ++k) a = __fadd_rn(a, __fmul_rn(Y[u][k], sb[v * 8 + k]));
            X[u][v] = __fmul_rn(a, gain[(lv * 8 + u) * 8 + v]);
        }
    // x = B^T Y B
    for (int y = 0; y < 8; ++y)
        for (int v = 0; v < 8; ++v) {
            float a = 0.f;
            for (int k = 0; k < 8; ++k) a = __fadd_rn(a, __fmul_rn(sb[k * 8 + y], X[k][v]));
            Y[y][v] = a;
        }
    for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) {
            float a = 0.f;
            for (int k = 0; k < 8; ++k) a = __fadd_rn(a, __fmul_rn(Y[y][k], sb[k * 8 + x]));
            int v = __float2int_rn(a);
            dst[base + y * rs + (long long)x * c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
}

int check_common(const void* src, const void* map, const void* dst, int n, int h, int w, int c, int b, int by, int bx,
                 const char* what) {
    ELVIS_REQUIRE(src && map && dst, "%s: null pointer", what);
    ELVIS_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && b > 0, "%s: bad shape", what);
    ELVIS_REQUIRE(h % b == 0 && w % b == 0 && by == h / b && bx == w / b,
                  "%s: image %dx%d must be divisible by block_size %d and the map must be %dx%d", what, h, w, b, h / b, w / b);
    return ELVIS_OK;
}

}  // namespace

extern "C" int elvis_degrade_downsample_u8(const uint8_t* src, const int32_t* levels, uint8_t* dst, int n, int h, int w, int c,
                                           int block, int by, int bx, elvis_stream_t stream) {
    int rc = check_common(src, levels, dst, n, h, w, c, block, by, bx, "elvis_degrade_downsample_u8");
    if (rc) return rc;
    ELVIS_REQUIRE(block <= 16 && (block & (block - 1)) == 0, "elvis_degrade_downsample_u8: block_size must be a power of two <= 16");
    const long long total = (long long)n * by * bx * c;
    hipLaunchKernelGGL(degrade_downsample_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, (hipStream_t)stream, src,
                       levels, dst, n, h, w, c, block, by, bx, total);
    ELVIS_CHECK_LAUNCH("elvis_degrade_downsample_u8");
    return ELVIS_OK;
}

extern "C" int elvis_degrade_gaussian_u8(const uint8_t* src, const int32_t* rounds, uint8_t* dst, int n, int h, int w, int c,
                                         int block, int by, int bx, float tap0, float tap1, float tap2, elvis_stream_t stream) {
    int rc = check_common(src, rounds, dst, n, h, w, c, block, by, bx, "elvis_degrade_gaussian_u8");
    if (rc) return rc;
    ELVIS_REQUIRE(block <= 16, "elvis_degrade_gaussian_u8: block_size must be <= 16");
    const long long total = (long long)n * by * bx * c;
    hipLaunchKernelGGL(degrade_gaussian_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, (hipStream_t)stream, src,
                       rounds, dst, n, h, w, c, block, by, bx, tap0, tap1, tap2, total);
    ELVIS_CHECK_LAUNCH("elvis_degrade_gaussian_u8");
    return ELVIS_OK;
}

extern "C" int elvis_degrade_dct_u8(const uint8_t* src, const int32_t* levels, uint8_t* dst, const float* basis64,
                                    const float* gain, int n_levels, int n, int h, int w, int c, int by, int bx,
                                    elvis_stream_t stream) {
    int rc = check_common(src, levels, dst, n, h, w, c, 8, by, bx, "elvis_degrade_dct_u8");
    if (rc) return rc;
    ELVIS_REQUIRE(basis64 && gain && n_levels > 0, "elvis_degrade_dct_u8: basis / gain tables missing");
    const long long total = (long long)n * by * bx * c;
    hipLaunchKernelGGL(degrade_dct_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, (hipStream_t)stream, src, levels, dst,
                       basis64, gain, n_levels, n, h, w, c, by, bx, total);
    ELVIS_CHECK_LAUNCH("elvis_degrade_dct_u8");
    return ELVIS_OK;
}
