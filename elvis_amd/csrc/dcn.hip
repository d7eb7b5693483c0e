// DCNv2 (modulated deformable 3x3 convolution) and the temporal-window helpers of the
// LaplacianVCAR-style restorer (ELVIS v2 DCT slot; README.md:11-16 mentions only the upstream
// `ops/dcn` CUDA build, which is not in the reference - SURVEY.md F1/F2).
//
//   out[p][co] = b[co] + sum_{g,k} sum_{c in group g} W[co][c][k] * m[g][k][p] * bilinear(x[c], p + tap_k + d[g][k][p])
//
// with offsets laid out channel = (g*9 + k)*2 + {0: dy, 1: dx}, masks channel = g*9 + k (already
// passed through the sigmoid by the caller's choice: `mask_sigmoid`), zero contribution from
// out-of-image corners.
//
// Two kernels:
//  * dcnv2_tile_kernel (f16, one channel per deformable group, cin <= 8, cout <= 64 - the restorer's shape):
//    a workgroup owns an 8 x 32 pixel tile; the input window (tile + 8 pixels of halo) sits in LDS as fp32
//    planes, so the four bilinear corners of a tap are two ds_read2_b32 (offsets are bounded in practice;
//    a corner outside the window falls back to a bounds-checked global read); a thread reads its pixel's
//    offset/mask row with 16-byte loads, applies the sigmoid once per (group, tap), and packs the K = 9*cin
//    modulated samples (K 63 -> 64) as f16 into a swizzled [pixel][32 k] LDS image; phase 2 is
//    [64 cout] x [K] x [64 px] per wave on v_mfma_f32_16x16x32_f16 with fp32 accumulation, bias + ReLU and
//    32-byte-per-lane NHWC stores.  HBM-bound by design: 378 B of offsets/masks in, 128 B out per pixel.
//  * dcnv2_kernel (any dtype / grouping): the generic scalar form - the fp32 exact mode and odd shapes.
#include "common.h"
#include <type_traits>
#include <stdlib.h>
#include <mutex>

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

// ------------------------------------------------------------------------------------------------------------
constexpr int TY_ = 8, TX_ = 32, RH = 8;                     // tile, halo radius of the LDS input window
constexpr int WY_ = TY_ + 2 * RH, WX_ = TX_ + 2 * RH;        // 24 x 48 window
constexpr int WP_ = WY_ * WX_;
constexpr int XIN_BYTES = 8 * WP_ * 4;                       // fp32 planes [8][WY_][WX_]
constexpr int IMG_BYTES = 64 * 64;                           // one [64 rows][32 k] f16 image (rows = cout or pixels)

__device__ __forceinline__ int img_off(int row, int q) {     // 64-byte rows, 16-byte chunk q, XOR swizzle (as conv.hip)
    return row * 64 + ((q ^ (((row >> 2) & 1) << 1)) << 4);
}

template <int CIN>   // compile-time: the mask halfs start at 18*CIN and every register index below must be static
__global__ __launch_bounds__(256, 2) void dcnv2_tile_kernel(const half_t* __restrict__ x, const half_t* __restrict__ om,
                                                             const half_t* __restrict__ wt, const float* __restrict__ bias,
                                                             half_t* __restrict__ out, int n, int h, int w,
                                                             int x_pitch, int om_pitch, int mask_sigmoid, int cout,
                                                             int out_pitch, int act, int tiles_x, int tiles_y) {
    constexpr int cin = CIN, ksteps = (CIN * 9 + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* const xin = reinterpret_cast<float*>(smem_raw);
    char* const wimg = smem_raw + XIN_BYTES;                               // [ksteps] images of the weights
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* const col = wimg + ksteps * IMG_BYTES + wave * ksteps * IMG_BYTES;   // this wave's [ksteps] images of samples
    constexpr int K = cin * 9;

    int t = blockIdx.x;
    const int txi = t % tiles_x; t /= tiles_x;
    const int tyi = t % tiles_y;
    const int ni = t / tiles_y;
    const int y00 = tyi * TY_ - RH, x00 = txi * TX_ - RH;                  // image coordinates of window (0, 0)
    const half_t* const xb = x + (long long)ni * h * w * x_pitch;

    // ---- stage the input window (zero outside the image) and the weights (rows permuted so that a lane ends up
    //      with 16 contiguous output channels: image row fi*16 + 4q + r holds channel 16q + 4fi + r)
    for (int i = tid; i < WP_; i += 256) {
        const int wy = i / WX_, wx = i - wy * WX_;
        const int gy = y00 + wy, gx = x00 + wx;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (gy >= 0 && gy < h && gx >= 0 && gx < w) v = *reinterpret_cast<const uint4*>(xb + ((long long)gy * w + gx) * x_pitch);
        const half8 hv = __builtin_bit_cast(half8, v);
#pragma unroll
        for (int c = 0; c < 8; ++c) xin[c * WP_ + i] = (float)hv[c];
    }
    for (int i = tid; i < ksteps * IMG_BYTES / 4; i += 256) reinterpret_cast<unsigned*>(wimg)[i] = 0u;
    __syncthreads();
    {   // [cout][K] halfs, read as 16-byte vectors (the host checks the pointer's alignment) and scattered by element:
        // two vector loads per thread instead of sixteen 2-byte ones on every workgroup's start-up path
        auto put = [&](int i, half_t v) {
            const int co = i / K, k = i - co * K;
            const int row = ((co & 15) >> 2) * 16 + (co >> 4) * 4 + (co & 3);
            *reinterpret_cast<half_t*>(wimg + (k >> 5) * IMG_BYTES + img_off(row, (k & 31) >> 3) + (k & 7) * 2) = v;
        };
        const int nv = (cout * K) >> 3;
        for (int i = tid; i < nv; i += 256) {
            const half8 v = *reinterpret_cast<const half8*>(wt + 8 * i);
#pragma unroll
            for (int e = 0; e < 8; ++e) put(8 * i + e, v[e]);
        }
        for (int i = 8 * nv + tid; i < cout * K; i += 256) put(i, wt[i]);
    }

    // ---- phase 1: one pixel per thread
    const int ty = tid >> 5, tx = tid & 31;                                // wave w owns tile rows 2w, 2w + 1
    const int gy = tyi * TY_ + ty, gx = txi * TX_ + tx;
    const bool pok = gy < h && gx < w;
    const long long pix = ((long long)ni * h + (pok ? gy : 0)) * w + (pok ? gx : 0);
    // the pixel's 27*CIN offset / mask halfs (189 for CIN = 7, 216 for CIN = 8), sized by CIN: every offset dword
    // (index < 9*CIN) and every mask half (index 18*CIN .. 27*CIN-1) below must lie inside the row that was loaded
    constexpr int n16 = (27 * cin * 2 + 15) / 16;                          // 16-byte pieces that hold them (the host checks
    unsigned omw[4 * n16];                                                 //  om_pitch >= 8 * n16 halfs)
    {
        const uint4* op = reinterpret_cast<const uint4*>(om + pix * om_pitch);
#pragma unroll
        for (int i = 0; i < n16; ++i) {
            const uint4 v = op[i];
            omw[4 * i] = v.x; omw[4 * i + 1] = v.y; omw[4 * i + 2] = v.z; omw[4 * i + 3] = v.w;
        }
    }
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    const float fgy = (float)gy, fgx = (float)gx, fh2 = (float)h + 2.0f, fw2 = (float)w + 2.0f;
    constexpr int mask_h0 = 18 * cin;                                      // first mask half
    static_assert(((mask_h0 + 9 * cin - 1) >> 1) < 4 * n16 && 9 * cin <= 4 * n16, "offset / mask row does not cover every (group, tap)");
    const int prow = lane;                                                 // this pixel's row in the wave's images
    // Offsets of at most 6 pixels keep every corner of every tap inside the LDS window (|tap| <= 1, halo 8, the
    // window check below then always passes): a wave whose 64 pixels all satisfy that - every wave of the DCT slot,
    // whose offsets are a fraction of a pixel - runs the sample loop WITHOUT the window test and without the global-read
    // path.  That loop is a quarter of the other's code (the fully unrolled bounds-checked reads are ~8 k instructions,
    // more than the instruction cache holds) and saves eight VALU per sample.
    bool small = true;
    {
        // on the BIT PATTERNS of |dy|, |dx|: for non-negative halfs integer order is value order, and inf / NaN
        // (>= 0x7c00) compare as the largest values - they take the checked path, whose clamps handle them
        typedef unsigned short ushort2w __attribute__((ext_vector_type(2)));
        ushort2w mx = {0, 0};
#pragma unroll
        for (int kk = 0; kk < K; ++kk)
            mx = __builtin_elementwise_max(mx, __builtin_bit_cast(ushort2w, omw[kk] & 0x7fff7fffu));
        small = mx[0] <= 0x4600 && mx[1] <= 0x4600;   // 6.0 in f16
    }
    const bool wave_small = __builtin_amdgcn_ballot_w64(!small) == 0ull;
    auto sample_loop = [&](auto checked) {
        constexpr bool CHECK = decltype(checked)::value;
        unsigned pk[4];                                                    // eight packed samples
#pragma unroll
        for (int kk = 0; kk < ksteps * 32; ++kk) {                         // k = g*9 + tap (one channel per group)
            const int g = kk / 9, tap = kk - g * 9;
            float v = 0.f;
            if (kk < K) {
                const half2v d = __builtin_bit_cast(half2v, omw[kk]);      // (dy, dx) = halfs 2kk, 2kk + 1
                const int mh = mask_h0 + kk;
                const half2v mm = __builtin_bit_cast(half2v, omw[mh >> 1]);   // static index (kk < K: inside the row)
                float m = (float)mm[mh & 1];
                // (this kernel is VALU-bound - 63 samples per pixel - so the sigmoid is the 4-instruction exp2 / rcp form:
                //  ~1e-6 relative, far below the f16 storage of the samples; the clamps are one v_med3 each)
                if (mask_sigmoid) m = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * m));
                const float sy = (fgy + (float)(tap / 3 - 1)) + (float)d[0], sx = (fgx + (float)(tap % 3 - 1)) + (float)d[1];
                const float fy = floorf(sy), fx = floorf(sx);
                const float ly = sy - fy, lx = sx - fx;
                float v00, v01, v10, v11;
                if constexpr (CHECK) {
                    // clamp far-away samples (they contribute zero anyway) so the int conversion cannot overflow
                    const int y0 = (int)__builtin_amdgcn_fmed3f(fy, -4.0f, fh2), x0 = (int)__builtin_amdgcn_fmed3f(fx, -4.0f, fw2);
                    const int wy = y0 - y00, wx = x0 - x00;
                    if (wy >= 0 && wy < WY_ - 1 && wx >= 0 && wx < WX_ - 1) {  // all four corners inside the LDS window
                        const float* c0 = xin + g * WP_ + wy * WX_ + wx;
                        v00 = c0[0]; v01 = c0[1]; v10 = c0[WX_]; v11 = c0[WX_ + 1];
                    } else {                                                   // rare: bounds-checked global reads
                        auto tapg = [&](int yy, int xx) -> float {
                            return (yy >= 0 && yy < h && xx >= 0 && xx < w) ? (float)xb[((long long)yy * w + xx) * x_pitch + g] : 0.f;
                        };
                        v00 = tapg(y0, x0); v01 = tapg(y0, x0 + 1); v10 = tapg(y0 + 1, x0); v11 = tapg(y0 + 1, x0 + 1);
                    }
                } else {
                    const int wy = (int)fy - y00, wx = (int)fx - x00;          // inside the window by the offset bound
                    const float* c0 = xin + g * WP_ + wy * WX_ + wx;
                    v00 = c0[0]; v01 = c0[1]; v10 = c0[WX_]; v11 = c0[WX_ + 1];
                }
                // bilinear blend as three lerps (six instructions; the products-of-weights form was ten)
                const float a0 = fmaf(lx, v01 - v00, v00), a1 = fmaf(lx, v11 - v10, v10);
                v = fmaf(ly, a1 - a0, a0);
                v *= m;
                if (!pok) v = 0.f;
            }
            {   // pack two samples per dword, store eight per 16-byte chunk
                const unsigned hb = (unsigned)__builtin_bit_cast(unsigned short, (half_t)v);
                if ((kk & 1) == 0) pk[(kk & 7) >> 1] = hb; else pk[(kk & 7) >> 1] |= hb << 16;
                if ((kk & 7) == 7 && (kk >> 5) < ksteps)
                    *reinterpret_cast<uint4*>(col + (kk >> 5) * IMG_BYTES + img_off(prow, (kk & 31) >> 3)) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            }
        }
    };
    if (wave_small) sample_loop(std::false_type{}); else sample_loop(std::true_type{});
    __syncthreads();   // weights image complete (written by all threads); the wave's own samples are ordered by lgkmcnt

    // ---- phase 2: this wave's 64 pixels x 64 output channels on the matrix cores
    const int lq = lane >> 4, lr = lane & 15;
    float4v acc[4][4];
#pragma unroll
    for (int fi = 0; fi < 4; ++fi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = lq * 16 + fi * 4 + r;
            const float b = (bias && co < cout) ? bias[co] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[fi][j][r] = b;
        }
    for (int s = 0; s < ksteps; ++s) {
        half8 fa[4], fb[4];
#pragma unroll
        for (int fi = 0; fi < 4; ++fi) fa[fi] = *reinterpret_cast<const half8*>(wimg + s * IMG_BYTES + img_off(fi * 16 + lr, lq));
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const half8*>(col + s * IMG_BYTES + img_off(j * 16 + lr, lq));
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) acc[fi][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[fi], fb[j], acc[fi][j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pl = wave * 64 + j * 16 + lr;                            // pixel of the tile (row-major 8 x 32)
        const int oy = tyi * TY_ + (pl >> 5), ox = txi * TX_ + (pl & 31);
        if (oy >= h || ox >= w) continue;
        half_t* op = out + (((long long)ni * h + oy) * w + ox) * out_pitch + lq * 16;
        half_t o[16];
#pragma unroll
        for (int fi = 0; fi < 4; ++fi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[fi][j][r];
                if (act == 3) v = fmaxf(v, 0.f);
                o[fi * 4 + r] = (half_t)v;
            }
        if (lq * 16 + 16 <= cout && (out_pitch & 7) == 0) {
            *reinterpret_cast<uint4*>(op) = *reinterpret_cast<const uint4*>(o);
            *reinterpret_cast<uint4*>(op + 8) = *reinterpret_cast<const uint4*>(o + 8);
        } else {
            for (int e = 0; e < 16; ++e)
                if (lq * 16 + e < cout) op[e] = o[e];
        }
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
    // the tiled gather + MFMA kernel: f16, one channel per deformable group, up to 8 channels, up to 64 outputs
    static const bool no_tile = getenv("ELVIS_DCN_GENERIC") != nullptr;   // A/B switch
    if (dtype == ELVIS_F16 && !no_tile && deformable_groups == cin && (cin == 7 || cin == 8) && x_pitch == 8 && cout <= 64 &&
        om_pitch % 8 == 0 && om_pitch >= ((27 * cin * 2 + 15) / 16) * 8 && (((uintptr_t)x | (uintptr_t)offset_mask | (uintptr_t)out | (uintptr_t)weight) & 15) == 0) {
        const int ksteps = (K + 31) / 32;
        const int tiles_x = (w + TX_ - 1) / TX_, tiles_y = (h + TY_ - 1) / TY_;
        const size_t lds2 = XIN_BYTES + (size_t)ksteps * IMG_BYTES * 5;
        static std::mutex mu;
        static bool attr_set[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        {
            std::lock_guard<std::mutex> guard(mu);
            if (!attr_set[dev]) {
                hipError_t e = hipFuncSetAttribute((const void*)dcnv2_tile_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e == hipSuccess)
                    e = hipFuncSetAttribute((const void*)dcnv2_tile_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) {
                    elvis_set_error("elvis_dcnv2: cannot reserve LDS: %s", hipGetErrorString(e));
                    return ELVIS_E_RUNTIME;
                }
                attr_set[dev] = true;
            }
        }
        ELVIS_REQUIRE((long long)n * tiles_x * tiles_y < 0x7fffffffLL, "elvis_dcnv2: grid too large");
        if (cin == 7)
            hipLaunchKernelGGL(dcnv2_tile_kernel<7>, dim3((unsigned)(n * tiles_x * tiles_y)), dim3(256), lds2, (hipStream_t)stream,
                               (const half_t*)x, (const half_t*)offset_mask, (const half_t*)weight, bias, (half_t*)out, n, h, w,
                               x_pitch, om_pitch, mask_sigmoid, cout, out_pitch, act, tiles_x, tiles_y);
        else
            hipLaunchKernelGGL(dcnv2_tile_kernel<8>, dim3((unsigned)(n * tiles_x * tiles_y)), dim3(256), lds2, (hipStream_t)stream,
                               (const half_t*)x, (const half_t*)offset_mask, (const half_t*)weight, bias, (half_t*)out, n, h, w,
                               x_pitch, om_pitch, mask_sigmoid, cout, out_pitch, act, tiles_x, tiles_y);
        ELVIS_CHECK_LAUNCH("elvis_dcnv2(tile)");
        return ELVIS_OK;
    }
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
