// Implicit-GEMM convolution (3x3 / 1x1, stride 1/2, optional fused nearest-2x upsample of the
// input, optional virtual channel-concat of two inputs, optional fused GroupNorm-affine+SiLU
// prologue, bias / activation / residual epilogue) on the CDNA4 matrix cores.
//
//   D[co][px] = sum_{tap,ci} W[tap][ci][co] * X[px + tap][ci]
//
// MFMA operand roles: A = weights (rows = output channels), B = activations (cols = pixels), so
// each lane ends up holding 4 CONSECUTIVE output channels of one pixel -> one 8-byte (f16) or
// 16-byte (f32) NHWC store per 16x16 sub-tile.
//
// Both dtypes use the same LDS image: rows of 64 bytes (32 halfs / 16 floats of K), grouped in
// 16-row x 64-byte sub-tiles of 1 KiB, XOR-swizzled (chunk bit 1 ^= row bit 2) so that every
// ds_read_b128 lane group hits 16 distinct 16-byte slots for any 16 consecutive rows.  One ds_read_b128 per lane feeds
//   f16: one v_mfma_f32_16x16x32_f16   (lane (r,g) holds k = 8g..8g+7)
//   f32: four v_mfma_f32_16x16x4_f32   (lane (r,g) holds k = 4g..4g+3; MFMA e uses element e, so
//        the four instructions together cover the 16 k of the row - the k order is permuted
//        identically for A and B, which leaves the sum unchanged).
// f32 mode is exact IEEE fp32 FMA chains (parity mode); f16 mode accumulates in fp32.
#include "common.h"
#include <stdlib.h>
#include <mutex>

// Experiment hooks.  In the product build they are the identity / nothing.  Timing-only variants (no weight or
// halo staging, no barriers, no stores - all of which produce WRONG results - and the in-kernel phase stamps)
// are defined in tools/variants/conv_hooks.h and compiled only through tools/variants/conv_variant.hip
// (tools/build_variant.py), never into libelvis_amd.so.
#ifndef ELVIS_CONV_HOOKS
#define ELVIS_STAGE(x) x
#define ELVIS_STAGE_W(x) x
#define ELVIS_STAGE_H(x) x
#define ELVIS_SETPRIO(x)
#define ELVIS_BARRIER() __syncthreads()
#define ELVIS_HOOK_SKIP_STORE(tv)
#define ELVIS_HOOK_STAMP_BEGIN
#define ELVIS_HOOK_STAMP_LOOP
#define ELVIS_HOOK_STAMP_EPILOGUE
#define ELVIS_HOOK_STAMP_END
#endif

namespace {

struct ConvArgs {
    const void* x;
    const void* x2;
    const void* w;
    const float* bias;
    const void* res;
    const float* pa;
    const float* pb;
    void* out;
    int n, h, w_in;       // input dims (pre-upsample)
    int cin, cin_pitch, cin2, cin2_pitch;
    int cout, cout_pitch, res_pitch;
    int ksize, stride, pad, upsample;
    int ho, wo;
    int act, prologue;
    int nkc;              // K chunks per tap (over cin+cin2, each KC elements)
    int nkc1;             // chunks that belong to input 1
    int co_pad;           // padded cout in the packed weights
    long long M;          // n*ho*wo
    int n_co_tiles;
    long long n_px_tiles;
    // halo kernel only
    int two, tall;        // 256-thread two-workgroups-per-CU variant; its 16-row form
    float* stats;         // [n*tiles_y*tiles_x][cout][2] partial (sum, sumsq) or null
    int tiles_x, tiles_y;
    int sub, par_a, par_b;   // 2x2 modes (KS == 2): output pixel (ostr*y + par_a, ostr*x + par_b)
    int ostr, pad2y, pad2x;  // sub-pixel: ostr 2, pad 1 - par; space-to-depth stride-2 form: ostr 1, pad 0
    int s2d, istr, nkc_c, wfull;   // s2d: K chunk kc = phase (kc / nkc_c) of the full-res input (row stride wfull), istr = 2
    int strip;            // > 0: pixel tiles are walked in column strips of this many tiles (L2 reuse of halo rows)
    int strip_full;       // tiles_x / strip: whole strips per image row of tiles
    int x3;               // fp32 storage, error-compensated f16 MFMA (ELVIS_F32X3)
};

template <typename T> struct Frag;
template <> struct Frag<half_t> { typedef half8 type; };
template <> struct Frag<float> { typedef float4v type; };

__device__ __forceinline__ void mma_tile(float4v& acc, const half8& a, const half8& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_tile(float4v& acc, const float4v& a, const float4v& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
}

// X3 (ELVIS_F32X3): fp32 tensors, products on the f16 matrix pipe with the rounding error compensated.  Every operand
// is split v = hi + lo (hi = f16(v), lo = f16(v - hi): together ~22 significant bits; the MFMA does not flush f16
// subnormals - tools/probes/mfma_probe.hip) and hi*hi + hi*lo + lo*hi accumulate in fp32: errors ~1e-6 relative
// instead of f16's 5e-4.  The split happens ONCE per element, outside the MFMA loop: weights are packed and
// activations staged into LDS as (hi, lo) half pairs in the fp32 slot (x3_pair), so a lane's 16-byte fragment
// holds 4 channels x (hi, lo) = the eight K slots of one 16x16x32 MFMA.  With the activation fragment B used as
// it is, A1 = [a_hi, a_hi] x4 gives a_hi*(b_hi + b_lo) and A2 = [a_lo, 0] x4 adds a_lo*b_hi: two 16-cycle MFMAs
// (2.4x the fp32 MFMA's rate, measured) and 8 VALU per WEIGHT fragment, which 2*WPX MFMAs share.
__device__ __forceinline__ unsigned x3_pair(float v) {
    const half_t h = (half_t)v;
    const half_t l = (half_t)(v - (float)h);
    return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}
__device__ __forceinline__ uint4 x3_pair4(uint4 v) {
    return make_uint4(x3_pair(__builtin_bit_cast(float, v.x)), x3_pair(__builtin_bit_cast(float, v.y)),
                      x3_pair(__builtin_bit_cast(float, v.z)), x3_pair(__builtin_bit_cast(float, v.w)));
}
template <bool X3> __device__ __forceinline__ void mma_tile_x(float4v& acc, const half8& a, const half8& b) { mma_tile(acc, a, b); }
template <bool X3> __device__ __forceinline__ void mma_tile_x(float4v& acc, const float4v& a, const float4v& b) {
    if constexpr (X3) {
        const uint4 ap = __builtin_bit_cast(uint4, a);
        const uint4 a1 = make_uint4(__builtin_amdgcn_perm(ap.x, ap.x, 0x01000100u), __builtin_amdgcn_perm(ap.y, ap.y, 0x01000100u),
                                    __builtin_amdgcn_perm(ap.z, ap.z, 0x01000100u), __builtin_amdgcn_perm(ap.w, ap.w, 0x01000100u));
        const uint4 a2 = make_uint4(ap.x >> 16, ap.y >> 16, ap.z >> 16, ap.w >> 16);
        const half8 B = __builtin_bit_cast(half8, b);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a1), B, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a2), B, acc, 0, 0, 0);
    } else {
        mma_tile(acc, a, b);
    }
}

__device__ __forceinline__ int lds_row_off(int row, int q) {
    // byte offset of 16-byte chunk q of 64-byte `row` in the swizzled image: chunk bit 1 is XORed
    // with row bit 2.  Conflict-free for ds_read_b128 of ANY 16 consecutive rows (brute-forced
    // against the gfx950 lane-group table), which the halo kernel needs for its dx-shifted reads.
    return row * 64 + ((q ^ (((row >> 2) & 1) << 1)) << 4);
}

#ifndef ELVIS_G1_NST128
#define ELVIS_G1_NST128 3
#endif
#ifndef ELVIS_G1_NST64
#define ELVIS_G1_NST64 2   /* ring stages of the 1x1 GEMM path with a 64-channel tile */
#endif
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <typename T> struct PrologueVec { float a[DT<T>::VEC], b[DT<T>::VEC]; };
template <typename T> __device__ __forceinline__ uint4 prologue_apply(uint4 v, const float (&pa)[DT<T>::VEC], const float (&pb)[DT<T>::VEC]);
template <> __device__ __forceinline__ uint4 prologue_apply<float>(uint4 v, const float (&pa)[4], const float (&pb)[4]) {
    float* f = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float t = fmaf(f[i], pa[i], pb[i]);
        f[i] = t / (1.0f + expf(-t));
    }
    return v;
}
template <> __device__ __forceinline__ uint4 prologue_apply<half_t>(uint4 v, const float (&pa)[8], const float (&pb)[8]) {
    half_t* hv = reinterpret_cast<half_t*>(&v);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float t = fmaf((float)hv[i], pa[i], pb[i]);
        hv[i] = (half_t)(t * __builtin_amdgcn_rcpf(1.0f + __expf(-t)));
    }
    return v;
}

// 256-thread f16 kernels: the same GroupNorm-affine + SiLU on one dword (two halfs) in eleven VALU instructions.
// The table holds a' = -log2(e) a and b' = -log2(e) b, so u = fma(x, a', b') = -log2(e) t feeds v_exp_f32 directly;
// d = fma(exp2(u), K, K) = K (1 + e^-t) with K = -log2(e), and u * rcp(d) = t / (1 + e^-t).  hipcc's own code for
// prologue_apply<half_t> is ~44 issue cycles per element (separate cvt, packed-f32 pairs that need a re-pack, the
// -log2(e) multiply); this is 32, and the VALU block between the two barriers of a K chunk is what the prologue costs
// (15-19 % of the kernel: at two waves per SIMD the matrix and vector instructions of the pair do not overlap
// enough to hide it, tools/experiments/README.md).  (trans -> VALU forwarding: one independent instruction between.)
__device__ __forceinline__ unsigned prologue_dword_f16(unsigned w, float a0, float a1, float b0, float b1, float K) {
    float u0, u1, e1;   // (e0 lives in w's register once both halfs have been read)
    asm volatile(
        "v_fma_mix_f32 %1, %0, %4, %6 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %2, %0, %5, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_exp_f32 %0, %1\n\t"
        "v_exp_f32 %3, %2\n\t"
        "v_fma_f32 %0, %0, %8, %8\n\t"
        "v_fma_f32 %3, %3, %8, %8\n\t"
        "v_rcp_f32 %0, %0\n\t"
        "v_rcp_f32 %3, %3\n\t"
        "v_mul_f32 %1, %1, %0\n\t"
        "v_mul_f32 %2, %2, %3\n\t"
        "v_cvt_pk_f16_f32 %0, %1, %2"
        : "+v"(w), "=&v"(u0), "=&v"(u1), "=&v"(e1)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "s"(K));
    return w;
}

// Half-vector (8-byte) forms: the halo kernel spreads the prologue over the MFMA taps in pieces.
__device__ __forceinline__ uint2 prologue_apply_half(uint2 v, const float (&pa)[2], const float (&pb)[2], float) {
    float* f = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float t = fmaf(f[i], pa[i], pb[i]);
        f[i] = t / (1.0f + expf(-t));
    }
    return v;
}
__device__ __forceinline__ uint2 prologue_apply_half(uint2 v, const float (&pa)[4], const float (&pb)[4], half_t) {
    half_t* hv = reinterpret_cast<half_t*>(&v);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float t = fmaf((float)hv[i], pa[i], pb[i]);
        hv[i] = (half_t)(t * __builtin_amdgcn_rcpf(1.0f + __expf(-t)));
    }
    return v;
}

// WCO/WPX: 16x16 sub-tiles per wave along co / px.  NW_CO x NW_PX waves per workgroup.
template <typename T, int WCO, int WPX, int NW_CO, int NW_PX>
__global__ __launch_bounds__(64 * NW_CO * NW_PX) void conv_igemm_kernel(ConvArgs p) {
    constexpr int NT = 64 * NW_CO * NW_PX;
    constexpr int TCO = 16 * WCO * NW_CO;
    constexpr int TPX = 16 * WPX * NW_PX;
    constexpr int VEC = DT<T>::VEC;       // elements per 16 B
    constexpr int KC = 4 * VEC;           // elements per 64-byte LDS row
    constexpr int A_CHUNKS = TCO * 4;     // 16-byte chunks of the weight tile
    constexpr int B_CHUNKS = TPX * 4;
    constexpr int A_PER = (A_CHUNKS + NT - 1) / NT;
    constexpr int B_PER = (B_CHUNKS + NT - 1) / NT;
    constexpr int A_BYTES = TCO * 64, B_BYTES = TPX * 64;
    typedef typename Frag<T>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto lds_a = [&](int buf) -> char* { return smem + buf * (A_BYTES + B_BYTES); };
    auto lds_b = [&](int buf) -> char* { return smem + buf * (A_BYTES + B_BYTES) + A_BYTES; };

    // XCD-aware remap: blocks b and b+8 share an XCD (round-robin dispatch).  Give each XCD a
    // contiguous range of logical tiles so that the co-tiles of one pixel tile, and vertically
    // adjacent pixel tiles, hit the same L2.
    long long nblk = (long long)p.n_co_tiles * p.n_px_tiles;
    long long bid = blockIdx.x;
    {
        long long q = nblk / 8, r = nblk % 8;
        long long xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int co_tile = (int)(bid % p.n_co_tiles);
    const long long px_tile = bid / p.n_co_tiles;
    const int co0 = co_tile * TCO;
    const long long m0 = px_tile * TPX;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int w_co = wave / NW_PX, w_px = wave % NW_PX;

    // ---- per-thread staging state for the pixel rows this thread loads
    int b_row[B_PER], b_q[B_PER], b_oy[B_PER], b_ox[B_PER];
    long long b_nbase[B_PER];  // element offset of image n in input 1 (per pitch unit: pixels)
    bool b_ok[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        int chunk = tid + i * NT;
        b_row[i] = chunk >> 2;
        b_q[i] = chunk & 3;
        long long m = m0 + b_row[i];
        b_ok[i] = (chunk < B_CHUNKS) && (m < p.M);
        long long mm = b_ok[i] ? m : 0;
        int hw = p.ho * p.wo;
        int nimg = (int)(mm / hw);
        int rem = (int)(mm - (long long)nimg * hw);
        b_oy[i] = rem / p.wo;
        b_ox[i] = rem - b_oy[i] * p.wo;
        b_nbase[i] = (long long)nimg * p.h * p.w_in;
    }
    const int lh = p.upsample ? p.h * 2 : p.h;   // logical input size seen by the conv
    const int lw = p.upsample ? p.w_in * 2 : p.w_in;
    const int ntaps = p.ksize * p.ksize;
    const int nsteps = ntaps * p.nkc;

    uint4 ra[A_PER], rb[B_PER];

    auto load_step = [&](int s) {
        int tap = s / p.nkc, kc = s - tap * p.nkc;
        int ky = tap / p.ksize, kx = tap - ky * p.ksize;
        // weights: linear 64-byte rows
        const char* wsrc = (const char*)p.w + ((long long)(tap * p.nkc + kc) * p.co_pad + co0) * 64;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int chunk = tid + i * NT;
            if (A_CHUNKS % NT == 0 || chunk < A_CHUNKS) ra[i] = *reinterpret_cast<const uint4*>(wsrc + (long long)chunk * 16);
        }
        // activations: per-pixel gather of one 16-byte channel chunk
        const bool second = kc >= p.nkc1;
        const char* xsrc = (const char*)(second ? p.x2 : p.x);
        const int pitch = second ? p.cin2_pitch : p.cin_pitch;
        const int cvalid = second ? p.cin2 : p.cin;
        const int kcl = second ? kc - p.nkc1 : kc;
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            int iy = b_oy[i] * p.stride + ky - p.pad;
            int ix = b_ox[i] * p.stride + kx - p.pad;
            int c0 = kcl * KC + b_q[i] * VEC;
            // branch-free gather: out-of-image / out-of-range chunks read element 0 and are zeroed by a
            // select (a branch around each load costs a serialized vmcnt(0) round trip per element)
            const bool ok = b_ok[i] && iy >= 0 && iy < lh && ix >= 0 && ix < lw && c0 < pitch;
            int sy = p.upsample ? (iy >> 1) : iy, sx = p.upsample ? (ix >> 1) : ix;
            long long e = ok ? (b_nbase[i] + (long long)sy * p.w_in + sx) * pitch + c0 : 0;
            uint4 v = *reinterpret_cast<const uint4*>(xsrc + e * (long long)sizeof(T));
            {
                if (p.prologue) {
                    int nimg = (int)(b_nbase[i] / ((long long)p.h * p.w_in));
                    const float* pa = p.pa + (long long)nimg * (p.cin + p.cin2) + (second ? p.cin : 0) + c0;
                    const float* pb = p.pb + (long long)nimg * (p.cin + p.cin2) + (second ? p.cin : 0) + c0;
                    float la[VEC], lb[VEC];
#pragma unroll
                    for (int e2 = 0; e2 < VEC; ++e2) {
                        bool in = (c0 + e2) < cvalid;
                        la[e2] = in ? pa[e2] : 0.0f;
                        lb[e2] = in ? pb[e2] : 0.0f;
                    }
                    v = prologue_apply<T>(v, la, lb);
                }
            }
            v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
            rb[i] = v;
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int chunk = tid + i * NT;
            if (chunk < A_CHUNKS) *reinterpret_cast<uint4*>(lds_a(buf) + lds_row_off(chunk >> 2, chunk & 3)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            int chunk = tid + i * NT;
            if (chunk < B_CHUNKS) *reinterpret_cast<uint4*>(lds_b(buf) + lds_row_off(b_row[i], b_q[i])) = rb[i];
        }
    };

    float4v acc[WCO][WPX];
#pragma unroll
    for (int i = 0; i < WCO; ++i)
#pragma unroll
        for (int j = 0; j < WPX; ++j) acc[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};

    const int lane_off = lds_row_off(lane & 15, lane >> 4);

    load_step(0);
    store_step(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) load_step(s + 1);
        frag_t fa[WCO], fb[WPX];
#pragma unroll
        for (int i = 0; i < WCO; ++i)
            fa[i] = *reinterpret_cast<const frag_t*>(lds_a(buf) + (w_co * WCO + i) * 1024 + lane_off);
#pragma unroll
        for (int j = 0; j < WPX; ++j)
            fb[j] = *reinterpret_cast<const frag_t*>(lds_b(buf) + (w_px * WPX + j) * 1024 + lane_off);
#pragma unroll
        for (int i = 0; i < WCO; ++i)
#pragma unroll
            for (int j = 0; j < WPX; ++j) mma_tile(acc[i][j], fa[i], fb[j]);
        if (s + 1 < nsteps) store_step(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds co = cobase + (lane>>4)*4 + r (r=0..3) for pixel col = lane&15
    const int cgrp = (lane >> 4) * 4;
    if ((p.cout & 3) == 0 && (p.res_pitch & 3) == 0) {
        // whole 4-channel groups: branch-free (clamped addresses, exec-masked stores).  Per-element
        // branches around the bias / residual loads cost one serialized memory round trip each.
        float bv[WCO][4];
        bool co_ok[WCO];
#pragma unroll
        for (int i = 0; i < WCO; ++i) {
            const int co = co0 + w_co * (16 * WCO) + cgrp * WCO + i * 4;
            co_ok[i] = co < p.cout;
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[i][r] = 0.f;
            if (p.bias) {
                const float* bp = p.bias + (co_ok[i] ? co : 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[i][r] = bp[r];
            }
        }
        uint2 rv[WCO][WPX];
        if (p.res) {
#pragma unroll
            for (int j = 0; j < WPX; ++j) {
                const long long m = m0 + (w_px * WPX + j) * 16 + (lane & 15);
                const long long mc = m < p.M ? m : 0;
#pragma unroll
                for (int i = 0; i < WCO; ++i) {
                    const int co = co0 + w_co * (16 * WCO) + cgrp * WCO + i * 4;
                    const T* rp = (const T*)p.res + mc * p.res_pitch + (co_ok[i] ? co : 0);
                    if constexpr (sizeof(T) == 2) rv[i][j] = *reinterpret_cast<const uint2*>(rp);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < WPX; ++j) {
            const long long m = m0 + (w_px * WPX + j) * 16 + (lane & 15);
            const bool m_ok = m < p.M;
            const long long mc = m_ok ? m : 0;
#pragma unroll
            for (int i = 0; i < WCO; ++i) {
                const int co = co0 + w_co * (16 * WCO) + cgrp * WCO + i * 4;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bv[i][r];
                if (p.act == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_erf_f(v[r]);
                } else if (p.act == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + expf(-v[r]));
                } else if (p.act == 3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
                }
                if (p.res) {
                    if constexpr (sizeof(T) == 2) {
                        const half4 h4 = __builtin_bit_cast(half4, rv[i][j]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)h4[r];
                    } else {
                        const float4v f4 = *reinterpret_cast<const float4v*>((const T*)p.res + mc * p.res_pitch + (co_ok[i] ? co : 0));
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += f4[r];
                    }
                }
                T* op = (T*)p.out + mc * p.cout_pitch + (co_ok[i] ? co : 0);
                if (m_ok && co_ok[i]) {
                    if constexpr (sizeof(T) == 2) {
                        half4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                        *reinterpret_cast<half4*>(op) = hv;
                    } else {
                        *reinterpret_cast<float4v*>(op) = (float4v){v[0], v[1], v[2], v[3]};
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < WPX; ++j) {
        long long m = m0 + (w_px * WPX + j) * 16 + (lane & 15);
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < WCO; ++i) {
            int co = co0 + w_co * (16 * WCO) + cgrp * WCO + i * 4;
            if (co >= p.cout) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            int nv = p.cout - co < 4 ? p.cout - co : 4;
            if (p.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < nv) v[r] += p.bias[co + r];
            }
            if (p.act == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_erf_f(v[r]);
            } else if (p.act == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + expf(-v[r]));
            } else if (p.act == 3) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
            }
            if (p.res) {
                const T* rp = (const T*)p.res + m * p.res_pitch + co;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < nv) v[r] += to_f(rp[r]);
            }
            T* op = (T*)p.out + m * p.cout_pitch + co;
            if (nv == 4) {
                if constexpr (sizeof(T) == 2) {
                    half4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                    *reinterpret_cast<half4*>(op) = hv;
                } else {
                    *reinterpret_cast<float4v*>(op) = (float4v){v[0], v[1], v[2], v[3]};
                }
            } else {
                for (int r = 0; r < nv; ++r) op[r] = from_f<T>(v[r]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 "halo" kernel - the hot kernel of the SinSR path.
//
// A workgroup (4 waves) owns an 8 x 32 pixel output tile x TCO output channels.  Per 64-byte
// K chunk (32 halfs / 16 floats of input channels) the (8+2) x (32+2) input halo is staged ONCE
// in LDS - with the fused GroupNorm-affine + SiLU prologue applied on the way in - and the nine
// taps read dx/dy-shifted windows of it, so activation bytes cross L2->LDS 1.33x instead of 9x
// and the prologue math runs 1.33x instead of 9x per element.  Weights stream through a 3-slot LDS
// ring, one (tap, chunk) slice of TCO x 64 B per step, register-staged one step ahead.
// Epilogue: bias / activation / residual, NHWC store, and per-tile GroupNorm partial sums
// (sum, sum of squares per output channel) for the NEXT layer's normalisation - no atomics, the
// partials are reduced by elvis_gn_partials_to_sums.
// ACT = false: no epilogue activation compiled in (act == 0: bias/residual folded into the
// accumulator start).  The activation code (erf GELU, SiLU with a precise division) is large; keeping
// it out of the hot instantiations keeps prologue + loop + epilogue inside the instruction cache.
template <typename T, int TCO, int NT, int TY, bool PRO, int KS, bool ACT, bool X3>
__device__ __forceinline__ void conv3x3_halo_body(const ConvArgs& p) {
    // KS = 3: 3x3 / pad 1 (halo of one pixel).  KS = 1: 1x1 conv / linear layer - the same staging
    // pipeline with no halo and one tap per K chunk (HBM-bound: what matters is bytes in flight).
    // KS = 2: one parity of the sub-pixel decomposition of "nearest-2x upsample + 3x3 conv": the 3x3
    // kernel on the upsampled grid collapses to a 2x2 kernel (pre-summed weights) on the low-res
    // grid for each output parity (a,b), 16 instead of 36 taps per low-res pixel (2.25x fewer FLOPs);
    // top/left padding is 1-a / 1-b and the tile writes output pixels (2y+a, 2x+b).
    constexpr int TX = 32, HX = TX + KS - 1, HY = TY + KS - 1, HP = HX * HY;
    // NT = 256 ("TWO"): 4-wave workgroups sized so that TWO of them share a CU (<= 80 KB of LDS, <= 256
    // VGPRs): one workgroup's start-up loads / epilogue stores overlap the other's MFMA loop.  To fit,
    // the halo is single-buffered (next chunk staged in registers, written between two barriers) and
    // the weight ring has two row slots (slot = row step parity).
    constexpr bool TWO = NT == 256;
    constexpr int NSLOT = (TWO || KS == 2) ? 2 : 3;   // weight ring slots; row step r uses slot r % NSLOT
    constexpr int NHB = TWO ? 1 : 2;                  // halo buffers
    // G1 (1x1 conv / linear layer, 256 threads): a plain K-loop GEMM, HBM-bound.  Both operands go
    // global -> LDS by LDS-DMA through a ring of G1_NST stages ([pixel tile | weight slice] per
    // 64-byte K chunk), G1_NST-1 chunks in flight, one barrier per chunk, counted vmcnt.  Needs
    // cin (and cin2) to be whole K chunks: the DMA cannot zero-fill channel padding.
#ifndef ELVIS_COUNTED_W
#define ELVIS_COUNTED_W 1
#endif
#ifndef ELVIS_TWO_WDMA
#define ELVIS_TWO_WDMA 1
#endif
    // weights of the 256-thread kernels: LDS-DMA (no staging registers) or register staging
    constexpr bool WDMA = TWO && ELVIS_TWO_WDMA;
    constexpr bool G1 = TWO && KS == 1;
    // prologue in the hand-written form (prologue_dword_f16): table entries pre-scaled by -log2(e)
#ifndef ELVIS_ASM_PROLOGUE
#define ELVIS_ASM_PROLOGUE 1
#endif
    constexpr bool PSC = TWO && PRO && sizeof(T) == 2 && ELVIS_ASM_PROLOGUE;
    constexpr int G1_NST = TCO == 64 ? ELVIS_G1_NST64 : ELVIS_G1_NST128;
    constexpr int HCH = HP * 4;
    constexpr int H_PER = (HCH + NT - 1) / NT;
    constexpr int HALO_BYTES = HP * 64;
    constexpr int W_TAP_BYTES = TCO * 64;                 // one (tap, chunk) weight slice
    constexpr int W_BYTES = KS * W_TAP_BYTES;             // one LDS slot = the KS taps of a kernel row
    constexpr int W_CHUNKS = W_TAP_BYTES / 16;
    constexpr int W_PER = (W_CHUNKS + NT - 1) / NT;       // 16-byte chunks per thread per tap
    // TCO in {16, 32, 64, 128}: up to 64 output channels per wave (WCO 16-row MFMA tiles)
    constexpr int NW_CO = TCO >= 64 ? TCO / 64 : 1, NW_PX = (NT / 64) / NW_CO;
    constexpr int ROWS = TY / NW_PX;
    constexpr int WPX = ROWS * 2, WCO = TCO >= 64 ? 4 : TCO / 16;
    constexpr int VEC = DT<T>::VEC, KC = 4 * VEC;
    typedef typename Frag<T>::type frag_t;
    static_assert(ROWS * NW_PX == TY && ROWS >= 1, "tile rows must split evenly over the pixel waves");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo = smem;                     // 2 buffers of HALO_BYTES
    char* const wring = smem + NHB * HALO_BYTES;   // NSLOT slots of W_BYTES
    // prologue table: per K chunk and 16-byte slice q, VEC x a then VEC x b (f32), so a thread's
    // GroupNorm affine for the slice it stages is two LDS vector reads - no long-lived registers
    float* const ptab = reinterpret_cast<float*>(smem + NHB * HALO_BYTES + NSLOT * W_BYTES);

    // tile decode in 32-bit unsigned arithmetic (the host checks the tile count < 2^31): the 64-bit divisions this
    // replaces were ~700 scalar instructions on every workgroup's critical path, ahead of its first load
    const unsigned nblk = (unsigned)p.n_co_tiles * (unsigned)p.tiles_x * (unsigned)p.tiles_y * (unsigned)p.n;
    unsigned bid = blockIdx.x;
    {
        const unsigned q = nblk >> 3, r = nblk & 7u;
        const unsigned xcd = bid & 7u, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int co_tile = 0;
    unsigned t = bid;
    if (p.n_co_tiles > 1) {
        t = bid / (unsigned)p.n_co_tiles;
        co_tile = (int)(bid - t * (unsigned)p.n_co_tiles);
    }
    // Pixel-tile walk.  Row-major over the full image width puts vertical neighbours tiles_x tiles apart; column
    // strips of `strip` tiles, walked top to bottom, bring the two tiles that share halo rows `strip` tiles apart,
    // inside the set of workgroups resident on the XCD.  (Measured: no change in kernel time on any hot shape -
    // the halo re-reads are served by L2 / the Infinity Cache either way; kept for its lower fabric traffic.)
    int tx, ty, nimg;
    if (p.strip > 0) {
        const unsigned per_img = (unsigned)p.tiles_x * (unsigned)p.tiles_y;
        const unsigned ni = t / per_img;
        unsigned r = t - ni * per_img;
        const unsigned full = (unsigned)p.strip_full, strip_tiles = (unsigned)p.strip * (unsigned)p.tiles_y;
        unsigned s = r / strip_tiles;
        unsigned sw = (unsigned)p.strip;
        if (s >= full) { s = full; sw = (unsigned)p.tiles_x - full * (unsigned)p.strip; }
        r -= s * strip_tiles;
        const unsigned y = r / sw;
        nimg = (int)ni;
        ty = (int)y;
        tx = (int)(s * (unsigned)p.strip + (r - y * sw));
    } else {
        const unsigned row = t / (unsigned)p.tiles_x;
        tx = (int)(t - row * (unsigned)p.tiles_x);
        const unsigned ni = row / (unsigned)p.tiles_y;
        ty = (int)(row - ni * (unsigned)p.tiles_y);
        nimg = (int)ni;
    }
    const int oy0 = ty * TY, ox0 = tx * TX, co0 = co_tile * TCO;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    ELVIS_HOOK_STAMP_BEGIN
    const int w_co = wave / NW_PX, w_px = wave % NW_PX;
    const int lh = p.upsample ? p.h * 2 : p.h, lw = p.upsample ? p.w_in * 2 : p.w_in;

    const int pad_y = KS == 2 ? p.pad2y : KS / 2, pad_x = KS == 2 ? p.pad2x : KS / 2;
    // ---- halo staging plan: chunk = tid + NT*i -> (halo pixel, 16-byte channel slice q = tid&3).
    // Branch-free: out-of-image / out-of-range chunks load pixel 0 and are zeroed by a select, so
    // the compiler can keep counted (not vmcnt(0)) waits on the prefetch pipeline.
    int h_src[H_PER];
    unsigned h_ok = 0;
#pragma unroll
    for (int i = 0; i < H_PER; ++i) {
        int chunk = tid + i * NT;
        int pix = chunk >> 2;
        int hy = pix / HX, hx = pix - hy * HX;
        int gy = oy0 + hy - pad_y, gx = ox0 + hx - pad_x;
        bool ok = chunk < HCH && gy >= 0 && gy < lh && gx >= 0 && gx < lw;
        int sy = p.upsample ? (gy >> 1) : gy, sx = p.upsample ? (gx >> 1) : gx;
        h_src[i] = ok ? ((nimg * p.h + sy) * p.istr * p.w_in + sx) * p.istr : 0;   // istr 2: pixel (2sy, 2sx) of the full-res input
        h_ok |= (ok ? 1u : 0u) << i;
    }
    const int q4 = tid & 3;   // NT is a multiple of 4: every chunk of this thread has the same q
    const int nkc = p.nkc;
    const int nrows = nkc * KS;   // row steps: one kernel row (KS taps) of one K chunk per barrier

    // Two staging phases per K chunk keep only half of the halo registers live at a time:
    // phase A = chunk slots [0, HA) loaded at tap 0, stored at tap 2; phase B = [HA, H_PER)
    // loaded at tap 3, stored at tap 5.
    constexpr int HA = (H_PER + 1) / 2;
    uint4 hreg[H_PER];

    int wl_off[W_PER];
#pragma unroll
    for (int i = 0; i < W_PER; ++i) {
        int chunk = tid + i * NT;
        wl_off[i] = lds_row_off(chunk >> 2, chunk & 3);
    }

    // channel offset (and, in the space-to-depth form, the phase's pixel offset) of K chunk kc
    auto chunk_c0 = [&](int kc, int& pixoff) -> int {
        pixoff = 0;
        if (KS == 2 && p.s2d) {
            const int ph = kc / p.nkc_c;
            pixoff = (ph >> 1) * p.wfull + (ph & 1);
            return (kc - ph * p.nkc_c) * KC + q4 * VEC;
        }
        return (kc >= p.nkc1 ? kc - p.nkc1 : kc) * KC + q4 * VEC;
    };
    auto halo_load = [&](int kc, int i0, int i1) {
        const bool second = kc >= p.nkc1;
        const char* xsrc = (const char*)(second ? p.x2 : p.x);
        const int pitch = second ? p.cin2_pitch : p.cin_pitch;
        int pixoff;
        const int c0 = chunk_c0(kc, pixoff);
        const int c0c = c0 < pitch ? c0 : 0;
#pragma unroll
        for (int i = i0; i < i1; ++i)
            hreg[i] = *reinterpret_cast<const uint4*>(xsrc + ((long long)(h_src[i] + pixoff) * pitch + c0c) * (long long)sizeof(T));
    };
    auto halo_store = [&](int kc, int buf, int i0, int i1) {
        const bool second = kc >= p.nkc1;
        const int pitch = second ? p.cin2_pitch : p.cin_pitch;
        int pixoff_;
        const int c0 = chunk_c0(kc, pixoff_);
        const bool ch_ok = c0 < pitch;
        char* dst = halo + buf * HALO_BYTES;
        float la[VEC], lb[VEC];
        if (PRO) {
            const float* tp = ptab + (kc * 4 + q4) * 2 * VEC;
#pragma unroll
            for (int e = 0; e < VEC; e += 4) {
                float4v a4 = *reinterpret_cast<const float4v*>(tp + e);
                float4v b4 = *reinterpret_cast<const float4v*>(tp + VEC + e);
#pragma unroll
                for (int k = 0; k < 4; ++k) { la[e + k] = a4[k]; lb[e + k] = b4[k]; }
            }
        }
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            uint4 v = hreg[i];
            if constexpr (PSC) {
                const float K = -1.4426950408889634f;
                v.x = prologue_dword_f16(v.x, la[0], la[1], lb[0], lb[1], K);
                v.y = prologue_dword_f16(v.y, la[2], la[3], lb[2], lb[3], K);
                v.z = prologue_dword_f16(v.z, la[4], la[5], lb[4], lb[5], K);
                v.w = prologue_dword_f16(v.w, la[6], la[7], lb[6], lb[7], K);
            } else if (PRO) {
                v = prologue_apply<T>(v, la, lb);
            }
            if constexpr (X3) v = x3_pair4(v);
            const bool keep = ch_ok && ((h_ok >> i) & 1u);
            v.x = keep ? v.x : 0u; v.y = keep ? v.y : 0u; v.z = keep ? v.z : 0u; v.w = keep ? v.w : 0u;
            int chunk = tid + i * NT;
            if ((i + 1) * NT <= HCH || chunk < HCH)
                *reinterpret_cast<uint4*>(dst + lds_row_off(chunk >> 2, chunk & 3)) = v;
        }
    };
    // Fine-grained prologue (PRO, 3x3): one half of one staged 16-byte slot at a time, so the SiLU
    // VALU work can be spread between the MFMAs of six taps instead of forming one serial block.
    auto halo_act = [&](int kc, int slot, int half) {
        constexpr int HV = VEC / 2;
        const float* tp = ptab + (kc * 4 + q4) * 2 * VEC + half * HV;
        float a[HV], b[HV];
#pragma unroll
        for (int e = 0; e < HV; e += 2) {
            float2 a2 = *reinterpret_cast<const float2*>(tp + e);
            float2 b2 = *reinterpret_cast<const float2*>(tp + VEC + e);
            a[e] = a2.x; a[e + 1] = a2.y; b[e] = b2.x; b[e + 1] = b2.y;
        }
        uint2 v = half ? make_uint2(hreg[slot].z, hreg[slot].w) : make_uint2(hreg[slot].x, hreg[slot].y);
        v = prologue_apply_half(v, a, b, T());
        if (half) { hreg[slot].z = v.x; hreg[slot].w = v.y; } else { hreg[slot].x = v.x; hreg[slot].y = v.y; }
    };
    auto halo_write = [&](int kc, int buf, int slot) {
        const bool second = kc >= p.nkc1;
        const int pitch = second ? p.cin2_pitch : p.cin_pitch;
        int pixoff_;
        const int c0 = chunk_c0(kc, pixoff_);
        const bool keep = c0 < pitch && ((h_ok >> slot) & 1u);
        uint4 v = hreg[slot];
        if constexpr (X3) v = x3_pair4(v);
        v.x = keep ? v.x : 0u; v.y = keep ? v.y : 0u; v.z = keep ? v.z : 0u; v.w = keep ? v.w : 0u;
        int chunk = tid + slot * NT;
        if ((slot + 1) * NT <= HCH || chunk < HCH)
            *reinterpret_cast<uint4*>(halo + buf * HALO_BYTES + lds_row_off(chunk >> 2, chunk & 3)) = v;
    };
    // row step r = kc*3 + dy: the three taps (dy, 0..2) of K chunk kc
    // (named registers, not arrays passed by reference: those end up in scratch)
    static_assert(W_PER <= 2, "weight staging holds at most two 16-byte chunks per thread per tap");
    constexpr bool W_FULL0 = W_CHUNKS >= NT, W_FULL1 = W_PER == 2 && W_CHUNKS >= 2 * NT;
    const bool w_ok0 = W_FULL0 || tid < W_CHUNKS, w_ok1 = W_PER == 2 && (W_FULL1 || tid + NT < W_CHUNKS);
    uint4 wr00 = {}, wr01 = {}, wr10 = {}, wr11 = {}, wr20 = {}, wr21 = {};
    auto w_load = [&](int r) {
        r = r < nrows ? r : nrows - 1;   // tail rows re-load the last slice (never consumed)
        int kc = r / KS, dy = r - kc * KS;
        const long long tap_stride = (long long)nkc * p.co_pad * 64;
        const char* wsrc = (const char*)p.w + ((long long)(dy * KS * nkc + kc) * p.co_pad + co0) * 64;
        const int o0 = (w_ok0 ? tid : 0) * 16, o1 = (w_ok1 ? tid + NT : 0) * 16;   // clamped: branch-free
        wr00 = *reinterpret_cast<const uint4*>(wsrc + o0);
        if (W_PER == 2) wr01 = *reinterpret_cast<const uint4*>(wsrc + o1);
        if (KS >= 2) {
            wr10 = *reinterpret_cast<const uint4*>(wsrc + tap_stride + o0);
            if (W_PER == 2) wr11 = *reinterpret_cast<const uint4*>(wsrc + tap_stride + o1);
        }
        if (KS == 3) {
            wr20 = *reinterpret_cast<const uint4*>(wsrc + 2 * tap_stride + o0);
            if (W_PER == 2) wr21 = *reinterpret_cast<const uint4*>(wsrc + 2 * tap_stride + o1);
        }
    };
    auto w_store = [&](int slot) {
        char* dst = wring + slot * W_BYTES;
        if (w_ok0) *reinterpret_cast<uint4*>(dst + wl_off[0]) = wr00;
        if (W_PER == 2 && w_ok1) *reinterpret_cast<uint4*>(dst + wl_off[W_PER - 1]) = wr01;
        if (KS >= 2) {
            if (w_ok0) *reinterpret_cast<uint4*>(dst + W_TAP_BYTES + wl_off[0]) = wr10;
            if (W_PER == 2 && w_ok1) *reinterpret_cast<uint4*>(dst + W_TAP_BYTES + wl_off[W_PER - 1]) = wr11;
        }
        if (KS == 3) {
            if (w_ok0) *reinterpret_cast<uint4*>(dst + 2 * W_TAP_BYTES + wl_off[0]) = wr20;
            if (W_PER == 2 && w_ok1) *reinterpret_cast<uint4*>(dst + 2 * W_TAP_BYTES + wl_off[W_PER - 1]) = wr21;
        }
    };
    // TWO: weights go global -> LDS directly (LDS-DMA, no staging registers).  The DMA writes
    // lane-linear (wave base + lane*16), so the swizzle is applied to the per-lane SOURCE chunk:
    // LDS position P = (row P>>2, slot P&3) holds logical chunk q = slot ^ swz(row), whose source
    // offset is lds_row_off(row, P&3) (the swizzle is an involution) = wl_off[].
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_ptr_t)smem);
    auto w_glds = [&](int r, int slot) {
        r = r < nrows ? r : nrows - 1;
        int kc = r / KS, dy = r - kc * KS;
        const long long tap_stride = (long long)nkc * p.co_pad * 64;
        const char* wsrc = (const char*)p.w + ((long long)(dy * KS * nkc + kc) * p.co_pad + co0) * 64;
#pragma unroll
        for (int t = 0; t < KS; ++t)
#pragma unroll
            for (int i = 0; i < W_PER; ++i)
            {
                // inline asm keeps the DMA out of hipcc's waitcnt bookkeeping (a pending LDS-DMA makes it
                // emit lgkmcnt(0) for every fragment read); completion = the explicit vmcnt(0) that
                // precedes each row-step barrier (w_glds_wait)
                unsigned keep;
                const unsigned dst = lds_base + (unsigned)((wring - smem) + slot * W_BYTES + t * W_TAP_BYTES + (i * NT + wave_u * 64) * 16);
                const char* src = wsrc + t * tap_stride + wl_off[i];
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
            }
    };
    auto w_glds_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    float4v acc[WCO][WPX];
#pragma unroll
    for (int i = 0; i < WCO; ++i)
#pragma unroll
        for (int j = 0; j < WPX; ++j) acc[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};

    const int lane_off = lds_row_off(lane & 15, lane >> 4);
    const int lq = lane >> 4, lr = lane & 15;

    // The first halo chunk and weight row are requested BEFORE the residual: the two HBM latencies then
    // overlap instead of adding up (the start-up phase is not hidden by anything but the CU's other
    // workgroup).
    constexpr bool EARLY = !G1 && !(NT == 256 && TCO == 64 && TY == 16) && !(NT == 256 && TCO == 128 && KS == 2);   // (these two have no registers to spare)
    if constexpr (EARLY) {
        halo_load(0, 0, H_PER);
        if constexpr (WDMA) w_glds(0, 0); else w_load(0);
    }
    // With no epilogue activation, y = conv + bias + residual: start the accumulators from
    // bias + residual so the residual's HBM latency hides under the start-up instead of sitting
    // on the epilogue's critical path.
    const bool fold = !ACT;   // host guarantees ACT == (act != 0)
    // vec4: every lane's 4-channel group is whole (cout % 4 == 0) -> branch-free bias / residual /
    // store code with clamped addresses.  (Per-element branches around loads make hipcc wait
    // vmcnt(0) per load: 24 serialized round trips per tile, measured 30k cycles.)
    const bool vec4 = (p.cout & 3) == 0 && (p.res_pitch & 3) == 0;
    bool co_ok[WCO];
#pragma unroll
    for (int i = 0; i < WCO; ++i) co_ok[i] = co0 + w_co * (16 * WCO) + lq * (4 * WCO) + i * 4 < p.cout;
    auto load_bias = [&](float (&bv)[WCO][4]) {   // vec4 only: 4 unconditional loads per channel group
#pragma unroll
        for (int i = 0; i < WCO; ++i) {
            const int co = co0 + w_co * (16 * WCO) + lq * (4 * WCO) + i * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[i][r] = 0.f;
            if (p.bias) {
                const float* bp = p.bias + (co_ok[i] ? co : 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[i][r] = bp[r];
            }
        }
    };
    // wide16: a lane's 4*WCO = 16 channels of a pixel are contiguous (the packed-weight row permutation):
    // residual loads and output stores move them as two 16-byte accesses on whole 128-byte lines
    constexpr bool WIDE16 = WCO == 4 && sizeof(T) == 2;
    const bool wide16 = WIDE16 && vec4 && (p.cout & 15) == 0 && (p.res_pitch & 7) == 0 && (p.cout_pitch & 7) == 0 &&
                        (((uintptr_t)p.res | (uintptr_t)p.out) & 15) == 0;
    if (fold && vec4) {
        float bv[WCO][4];
        load_bias(bv);
        if (p.res) {
#pragma unroll
            for (int j = 0; j < WPX; ++j) {
                const int oy = oy0 + w_px * ROWS + (j >> 1), ox = ox0 + (j & 1) * 16 + lr;
                const bool pix_ok = KS == 2 ? (oy < p.h && ox < p.w_in) : (oy < p.ho && ox < p.wo);
                const long long m = KS == 2 ? ((long long)nimg * p.ho + p.ostr * oy + p.par_a) * p.wo + p.ostr * ox + p.par_b
                                            : ((long long)nimg * p.ho + oy) * p.wo + ox;
                const long long mc = pix_ok ? m : 0;
                if constexpr (WIDE16) {
                    if (wide16) {   // the lane's 16 channels of this pixel are 32 contiguous bytes
                        const T* rp = (const T*)p.res + mc * p.res_pitch + (co_ok[0] ? co0 + w_co * 64 + lq * 16 : 0);
                        // parked raw in the accumulator registers; converted by res_convert() AFTER the
                        // prologue table / first-chunk activation, whose work hides this HBM latency
                        acc[0][j] = __builtin_bit_cast(float4v, *reinterpret_cast<const uint4*>(rp));
                        acc[2][j] = __builtin_bit_cast(float4v, *reinterpret_cast<const uint4*>(rp + 8));
                        continue;
                    }
                }
#pragma unroll
                for (int i = 0; i < WCO; ++i) {
                    const int co = co0 + w_co * (16 * WCO) + lq * (4 * WCO) + i * 4;
                    const T* rp = (const T*)p.res + mc * p.res_pitch + (co_ok[i] ? co : 0);
                    if constexpr (sizeof(T) == 2) {
                        half4 rv = *reinterpret_cast<const half4*>(rp);
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][j][r] = bv[i][r] + (float)rv[r];
                    } else {
                        float4v rv = *reinterpret_cast<const float4v*>(rp);
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][j][r] = bv[i][r] + rv[r];
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < WPX; ++j)
#pragma unroll
                for (int i = 0; i < WCO; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = bv[i][r];
        }
    } else if (fold && (p.res || p.bias)) {
#pragma unroll
        for (int j = 0; j < WPX; ++j) {
            const int oy = oy0 + w_px * ROWS + (j >> 1), ox = ox0 + (j & 1) * 16 + lr;
            const bool pix_ok = KS == 2 ? (oy < p.h && ox < p.w_in) : (oy < p.ho && ox < p.wo);
            const long long m = KS == 2 ? ((long long)nimg * p.ho + p.ostr * oy + p.par_a) * p.wo + p.ostr * ox + p.par_b
                                        : ((long long)nimg * p.ho + oy) * p.wo + ox;
#pragma unroll
            for (int i = 0; i < WCO; ++i) {
                const int co = co0 + w_co * (16 * WCO) + lq * (4 * WCO) + i * 4;
                if (!pix_ok || co >= p.cout) continue;
                const int nv = p.cout - co < 4 ? p.cout - co : 4;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r < nv) v[r] = p.bias[co + r];
                }
                if (p.res) {
                    const T* rp = (const T*)p.res + m * p.res_pitch + co;
                    if (nv == 4) {   // one 8-byte (f16) / 16-byte (f32) load per sub-tile
                        if constexpr (sizeof(T) == 2) {
                            half4 rv = *reinterpret_cast<const half4*>(rp);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
                        } else {
                            float4v rv = *reinterpret_cast<const float4v*>(rp);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += rv[r];
                        }
                    } else {
                        for (int r = 0; r < nv; ++r) v[r] += to_f(rp[r]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = v[r];
            }
        }
    }

    auto res_convert = [&]() {
        if constexpr (WIDE16) {
            if (fold && wide16 && p.res) {
                float bv[WCO][4];
                load_bias(bv);
#pragma unroll
                for (int j = 0; j < WPX; ++j) {
                    const half8 hlo = __builtin_bit_cast(half8, acc[0][j]), hhi = __builtin_bit_cast(half8, acc[2][j]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc[0][j][r] = bv[0][r] + (float)hlo[r];
                        acc[1][j][r] = bv[1][r] + (float)hlo[4 + r];
                        acc[2][j][r] = bv[2][r] + (float)hhi[r];
                        acc[3][j][r] = bv[3][r] + (float)hhi[4 + r];
                    }
                }
            }
        }
    };
    if constexpr (G1) {
        constexpr int STAGE = HALO_BYTES + W_TAP_BYTES;
        constexpr int L = H_PER + W_PER;   // LDS-DMA instructions per thread per stage
        static_assert(HCH % NT == 0 && W_CHUNKS % NT == 0, "G1 stages are whole 1 KiB wave pieces");
        const int swz_t = ((tid >> 4) & 1) << 1;   // (pix >> 2) & 1 of every piece this thread stages
        const int wave_u1 = __builtin_amdgcn_readfirstlane(tid >> 6);
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_ptr_t)smem);
        auto g1_dma = [&](const char* src, unsigned dst) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        };
        auto g1_issue = [&](int kc) {
            const int slot = kc % G1_NST;
            const bool second = kc >= p.nkc1;
            const char* xsrc = (const char*)(second ? p.x2 : p.x);
            const long long pitch = second ? p.cin2_pitch : p.cin_pitch;
            const int c0 = (second ? kc - p.nkc1 : kc) * KC + ((q4 ^ swz_t) * VEC);   // source chunk of LDS slot q4
#pragma unroll
            for (int i = 0; i < H_PER; ++i)
                g1_dma(xsrc + ((long long)h_src[i] * pitch + c0) * (long long)sizeof(T),
                       lds0 + (unsigned)(slot * STAGE + (i * NT + wave_u1 * 64) * 16));
            const char* wsrc = (const char*)p.w + ((long long)kc * p.co_pad + co0) * 64;
#pragma unroll
            for (int i = 0; i < W_PER; ++i)
                g1_dma(wsrc + wl_off[i], lds0 + (unsigned)(slot * STAGE + HALO_BYTES + (i * NT + wave_u1 * 64) * 16));
        };
#pragma unroll
        for (int s = 0; s < G1_NST - 1; ++s)
            if (s < nkc) g1_issue(s);
        res_convert();
        const int pp0 = (w_px * ROWS) * HX + lr;
        int bb[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int carry = ((pp0 & 3) + r) >> 2;
                int bit = ((pp0 >> 2) & 1) ^ carry ^ e;
                bb[r][e] = pp0 * 64 + ((lq ^ (bit << 1)) << 4);
            }
        const int a_off = HALO_BYTES + w_co * WCO * 1024 + lane_off;
        ELVIS_HOOK_STAMP_LOOP
        for (int kc = 0; kc < nkc; ++kc) {
            // this thread's pieces of chunk kc have landed when at most the younger chunks' DMAs remain
            if (kc + G1_NST - 2 < nkc) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((G1_NST - 2) * L) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();   // everyone's pieces landed; everyone is past its reads of chunk kc-1
            if (kc + G1_NST - 1 < nkc) g1_issue(kc + G1_NST - 1);
            const int sb = (kc % G1_NST) * STAGE;
            frag_t fa[WCO];
#pragma unroll
            for (int i = 0; i < WCO; ++i) fa[i] = *reinterpret_cast<const frag_t*>(smem + sb + a_off + i * 1024);
#pragma unroll
            for (int j = 0; j < WPX; ++j) {
                const int C = (j >> 1) * HX + (j & 1) * 16;
                frag_t fb = *reinterpret_cast<const frag_t*>(smem + sb + bb[C & 3][(C >> 2) & 1] + C * 64);
#pragma unroll
                for (int i = 0; i < WCO; ++i) mma_tile_x<X3>(acc[i][j], fa[i], fb);
            }
        }
        __syncthreads();   // the epilogue reuses LDS for the statistics reduction
    } else {
    // ---- prologue: halo(0) and weight row 0 into LDS; weight row 1 in flight in registers.
    // The first global loads are issued before the prologue table is built so their latency
    // overlaps it (one workgroup per CU: nothing else hides a workgroup's start-up).
    if constexpr (!EARLY && !G1) {
        halo_load(0, 0, H_PER);
        if constexpr (WDMA) w_glds(0, 0); else w_load(0);
    }
    if (PRO) {
        // table entry t = (kc*4 + q)*2*VEC + {0..VEC-1: a, VEC..2VEC-1: b}; channels past the
        // logical count get a = b = 0 (silu(0) = 0 keeps zero padding exact)
        const int ctot = p.cin + p.cin2;
        for (int t2 = tid; t2 < nkc * 4 * VEC; t2 += NT) {
            int kc = t2 / (4 * VEC), r = t2 - kc * 4 * VEC;       // r = q*VEC + e
            bool second = kc >= p.nkc1;
            int cl = (second ? kc - p.nkc1 : kc) * KC + r;         // channel within its input
            bool in = cl < (second ? p.cin2 : p.cin);
            long long g = (long long)nimg * ctot + (second ? p.cin : 0) + (in ? cl : 0);
            int q = r / VEC, e = r - q * VEC;
            const float psc = PSC ? -1.4426950408889634f : 1.0f;
            ptab[(kc * 4 + q) * 2 * VEC + e] = in ? psc * p.pa[g] : 0.0f;
            ptab[(kc * 4 + q) * 2 * VEC + VEC + e] = in ? psc * p.pb[g] : 0.0f;
        }
        __syncthreads();
    }
    halo_store(0, 0, 0, H_PER);
    res_convert();
    if constexpr (WDMA) {
        w_glds_wait();
    } else {
        w_store(0);
        w_load(1);
    }
    __syncthreads();

    ELVIS_HOOK_STAMP_LOOP
    // B-fragment addressing with ZERO per-read VALU.  A lane reads halo pixel x + C (x = its pixel at
    // tap (0,0) of the wave's first sub-tile, C a compile-time pixel offset) at byte
    //   (x+C)*64 + ((lq ^ (swz(x+C) << 1)) << 4),   swz(v) = (v >> 2) & 1.
    // With C = 4*Cq + Cr:  swz(x+C) = swz(x) ^ (Cq & 1) ^ carry(x & 3, Cr), so eight per-lane bases
    // bbase[Cr][Cq & 1] = x*64 + ((lq ^ ((swz(x) ^ carry(x&3,Cr) ^ (Cq&1)) << 1)) << 4) turn every read
    // into `ds_read_b128 v, bbase[..] offset:C*64`.
    const int pp0 = (w_px * ROWS) * HX + lr;
    int bb[4][2];   // bases into halo buffer 0; toggled by +-HALO_BYTES as the K chunks alternate
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int carry = ((pp0 & 3) + r) >> 2;
            int bit = ((pp0 >> 2) & 1) ^ carry ^ e;
            bb[r][e] = pp0 * 64 + ((lq ^ (bit << 1)) << 4);
        }
    const int a_off = w_co * WCO * 1024 + lane_off;

    // Row step r = kc*3 + dy reads LDS weight slot r%3 = dy and halo buffer kc&1.  At its top the
    // register set (holding row r+1, loaded one full row step = 3 x WCO*WPX MFMAs per wave ago) is
    // written to slot (dy+1)%3 and re-armed with the load of row r+2.  One barrier per row step.
    // The next K chunk's halo is staged in two register phases: A loaded at dy=0 / stored after the
    // dy=0 MFMAs, B loaded at dy=1 / stored after the dy=1 MFMAs.
#define ELVIS_ROW_STEP(DY)                                                                             \
    {                                                                                                  \
        const int rslot = TWO ? ((r0 + DY) & 1) : DY;                                                  \
        if constexpr (WDMA) {                                                                          \
            ELVIS_STAGE_W(w_glds(r0 + DY + 1, rslot ^ 1);)                                             \
        } else {                                                                                       \
            ELVIS_STAGE(w_store(TWO ? (rslot ^ 1) : (DY + 1) % NSLOT);)                                \
            ELVIS_STAGE(w_load(r0 + DY + 2);)                                                          \
        }                                                                                              \
        ELVIS_STAGE_H(if (DY == 0) halo_load(kcn, 0, PRO ? H_PER : HA);)                               \
        ELVIS_STAGE_H(if (DY == 1 && !PRO) halo_load(kcn, HA, H_PER);)                                 \
        if constexpr (TWO) {                                                                           \
            /* software-pipelined fragment reads: the B fragment of step s+1 and the A fragments of   \
               the next tap are in flight while the MFMAs of step s issue (counted lgkmcnt waits) */  \
            const char* wsb = wring + rslot * W_BYTES + a_off;                                         \
            frag_t fa[2][WCO], fb[2];                                                                  \
            _Pragma("unroll") for (int i = 0; i < WCO; ++i)                                            \
                fa[0][i] = *reinterpret_cast<const frag_t*>(wsb + i * 1024);                           \
            {                                                                                          \
                const int C = DY * HX;                                                                 \
                fb[0] = *reinterpret_cast<const frag_t*>(smem + bb[C & 3][(C >> 2) & 1] + C * 64);     \
            }                                                                                          \
            ELVIS_SETPRIO(1);                                                                          \
            _Pragma("unroll") for (int s = 0; s < KS * WPX; ++s) {                                     \
                const int dx = s / WPX, j = s - dx * WPX;                                              \
                if (s + 1 < KS * WPX) {                                                                \
                    const int ndx = (s + 1) / WPX, nj = (s + 1) - ndx * WPX;                           \
                    const int C = ((nj >> 1) + DY) * HX + (nj & 1) * 16 + ndx;                         \
                    fb[(s + 1) & 1] = *reinterpret_cast<const frag_t*>(smem + bb[C & 3][(C >> 2) & 1] + C * 64); \
                }                                                                                      \
                if (dx + 1 < KS && j < WCO)                                                            \
                    fa[(dx + 1) & 1][j] = *reinterpret_cast<const frag_t*>(wsb + (dx + 1) * W_TAP_BYTES + j * 1024); \
                _Pragma("unroll") for (int i = 0; i < WCO; ++i) mma_tile_x<X3>(acc[i][j], fa[dx & 1][i], fb[s & 1]); \
            }                                                                                          \
            ELVIS_SETPRIO(0);                                                                          \
            __builtin_amdgcn_sched_group_barrier(0x100, WCO + 1, 0);                                   \
            _Pragma("unroll") for (int s = 0; s < KS * WPX; ++s) {                                     \
                if ((s % WPX) < WCO && s / WPX + 1 < KS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); \
                else if (s + 1 < KS * WPX) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          \
                __builtin_amdgcn_sched_group_barrier(0x008, WCO, 0);                                   \
            }                                                                                          \
        } else {                                                                                       \
        _Pragma("unroll") for (int dx = 0; dx < KS; ++dx) {                                            \
            const char* ws = wring + rslot * W_BYTES + dx * W_TAP_BYTES + a_off;                       \
            frag_t fa[WCO];                                                                            \
            _Pragma("unroll") for (int i = 0; i < WCO; ++i)                                            \
                fa[i] = *reinterpret_cast<const frag_t*>(ws + i * 1024);                               \
            _Pragma("unroll") for (int j = 0; j < WPX; ++j) {                                          \
                const int C = ((j >> 1) + DY) * HX + (j & 1) * 16 + dx;                                \
                frag_t fb = *reinterpret_cast<const frag_t*>(smem + bb[C & 3][(C >> 2) & 1] + C * 64); \
                _Pragma("unroll") for (int i = 0; i < WCO; ++i) mma_tile_x<X3>(acc[i][j], fa[i], fb);        \
                /* PRO: this tap's share of the next chunk's prologue, placed mid-tap */               \
                if (PRO && DY >= 1 && j == WPX / 2 - 1) {                                              \
                    _Pragma("unroll") for (int pi = 0; pi < 2 * H_PER; ++pi)                           \
                        if (pi * 6 / (2 * H_PER) == (DY - 1) * 3 + dx) {                               \
                            ELVIS_STAGE(halo_act(kcn, pi / 2, pi & 1);)                                \
                            if ((pi & 1) && !TWO) { ELVIS_STAGE(halo_write(kcn, (kc + 1) & 1, pi / 2);) } \
                        }                                                                              \
                }                                                                                      \
            }                                                                                          \
            /* interleave the prologue VALU with this tap's MFMAs (1 MFMA : 3 VALU) */                \
            if (PRO && DY >= 1 && !TWO) {                                                              \
                _Pragma("unroll") for (int g = 0; g < WCO * WPX; ++g) {                                \
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                 \
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                 \
                }                                                                                      \
            }                                                                                          \
        }                                                                                              \
        }                                                                                              \
        ELVIS_STAGE(if (DY == 0 && !PRO && !TWO) halo_store(kcn, (kc + 1) & 1, 0, HA);)                \
        ELVIS_STAGE(if (DY == 1 && !PRO && !TWO) halo_store(kcn, (kc + 1) & 1, HA, H_PER);)            \
        if constexpr (WDMA) {                                                                          \
            /* retire the weight DMAs only: the halo loads of this row were issued AFTER them (younger), \
               so a counted wait leaves those HBM loads in flight across the barrier */                \
            constexpr int YOUNGER = ELVIS_COUNTED_W ? (DY == 0 ? (PRO ? H_PER : HA) : (DY == 1 && !PRO) ? H_PER - HA : 0) : 0; \
            ELVIS_STAGE_W(asm volatile("s_waitcnt vmcnt(%0)" :: "n"(YOUNGER) : "memory");)            \
        }                                                                                              \
        ELVIS_BARRIER();                                                                               \
    }
    for (int kc = 0; kc < nkc; ++kc) {
        const int r0 = kc * KS;
        const int kcn = kc + 1 < nkc ? kc + 1 : kc;   // last chunk re-stages itself into the idle buffer
        if constexpr (KS == 3) {
            ELVIS_ROW_STEP(0)
            ELVIS_ROW_STEP(1)
            ELVIS_ROW_STEP(2)
        } else if constexpr (KS == 2) {
            ELVIS_ROW_STEP(0)
            ELVIS_ROW_STEP(1)
        } else {
            // 1x1: one tap per chunk; weight slot kc % 3 (runtime), whole tile staged in one phase
            const int slot = kc % 3;
            w_store((kc + 1) % 3);
            w_load(kc + 2);
            halo_load(kcn, 0, H_PER);
            const char* ws = wring + slot * W_BYTES + a_off;
            frag_t fa[WCO];
#pragma unroll
            for (int i = 0; i < WCO; ++i) fa[i] = *reinterpret_cast<const frag_t*>(ws + i * 1024);
#pragma unroll
            for (int j = 0; j < WPX; ++j) {
                const int C = (j >> 1) * HX + (j & 1) * 16;
                frag_t fb = *reinterpret_cast<const frag_t*>(smem + bb[C & 3][(C >> 2) & 1] + C * 64);
#pragma unroll
                for (int i = 0; i < WCO; ++i) mma_tile_x<X3>(acc[i][j], fa[i], fb);
            }
            halo_store(kcn, (kc + 1) & 1, 0, H_PER);
            __syncthreads();
        }
        if constexpr (TWO) {
            // every wave is past its last read of this chunk's halo (barrier of the last row step):
            // overwrite the single buffer with the register-staged next chunk
            ELVIS_STAGE_H(if (kc + 1 < nkc) {
                // (with the prologue: GroupNorm-affine + SiLU applied here, in one block between the two
                // barriers - keeping its LDS table reads out of the row steps leaves their fragment
                // pipeline intact; the CU's other workgroup runs MFMAs meanwhile)
                halo_store(kcn, 0, 0, H_PER);
            })
            __syncthreads();
        } else {
            const int delta = (kc & 1) ? -HALO_BYTES : HALO_BYTES;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bb[r][0] += delta;
                bb[r][1] += delta;
            }
        }
    }
#undef ELVIS_ROW_STEP
    }   // !G1
    ELVIS_HOOK_STAMP_EPILOGUE

    // ---- epilogue
    float st[WCO][4], sq[WCO][4];
#pragma unroll
    for (int i = 0; i < WCO; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[i][r] = sq[i][r] = 0.f;
    const int cgrp = lq * 4;
    if (vec4) {
        // branch-free: clamped addresses, exec-masked stores, no per-element waits
        float bv[WCO][4];
        if constexpr (ACT) load_bias(bv);
#pragma unroll
        for (int j = 0; j < WPX; ++j) {
            const int oy = oy0 + w_px * ROWS + (j >> 1), ox = ox0 + (j & 1) * 16 + lr;
            const bool pix_ok = KS == 2 ? (oy < p.h && ox < p.w_in) : (oy < p.ho && ox < p.wo);
            const long long m = KS == 2 ? ((long long)nimg * p.ho + p.ostr * oy + p.par_a) * p.wo + p.ostr * ox + p.par_b
                                        : ((long long)nimg * p.ho + oy) * p.wo + ox;
            const long long mc = pix_ok ? m : 0;
            T wv[WCO][4];
#pragma unroll
            for (int i = 0; i < WCO; ++i) {
                const int co = co0 + w_co * (16 * WCO) + cgrp * WCO + i * 4;
                const bool ok = pix_ok && co_ok[i];
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if constexpr (ACT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += bv[i][r];
                    if (p.act == 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gelu_erf_f(v[r]);
                    } else if (p.act == 2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + expf(-v[r]));
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
                    }
                    if (p.res) {
                        const T* rp = (const T*)p.res + mc * p.res_pitch + (co_ok[i] ? co : 0);
                        if constexpr (sizeof(T) == 2) {
                            half4 rv = *reinterpret_cast<const half4*>(rp);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
                        } else {
                            float4v rv = *reinterpret_cast<const float4v*>(rp);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += rv[r];
                        }
                    }
                }
                T tv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) tv[r] = from_f<T>(v[r]);
                T* op = (T*)p.out + m * p.cout_pitch + co;
                ELVIS_HOOK_SKIP_STORE(tv)
                if (ok && !wide16) {
                    if constexpr (sizeof(T) == 2) {
                        half4 hv = {tv[0], tv[1], tv[2], tv[3]};
                        *reinterpret_cast<half4*>(op) = hv;
                    } else {
                        *reinterpret_cast<float4v*>(op) = (float4v){tv[0], tv[1], tv[2], tv[3]};
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    wv[i][r] = tv[r];
                    const float f = ok ? to_f(tv[r]) : 0.f;   // statistics of the STORED value
                    st[i][r] += f;
                    sq[i][r] = fmaf(f, f, sq[i][r]);
                }
            }
            if constexpr (WIDE16) {
                if (wide16 && pix_ok && co_ok[0]) {   // 16 contiguous channels: two 16-byte stores
                    half8 lo = {wv[0][0], wv[0][1], wv[0][2], wv[0][3], wv[1][0], wv[1][1], wv[1][2], wv[1][3]};
                    half8 hi = {wv[2][0], wv[2][1], wv[2][2], wv[2][3], wv[3][0], wv[3][1], wv[3][2], wv[3][3]};
                    T* op = (T*)p.out + m * p.cout_pitch + co0 + w_co * 64 + lq * 16;
                    *reinterpret_cast<half8*>(op) = lo;
                    *reinterpret_cast<half8*>(op + 8) = hi;
                }
            }
        }
    } else
#pragma unroll
    for (int j = 0; j < WPX; ++j) {
        const int oy = oy0 + w_px * ROWS + (j >> 1), ox = ox0 + (j & 1) * 16 + lr;
        const bool pix_ok = KS == 2 ? (oy < p.h && ox < p.w_in) : (oy < p.ho && ox < p.wo);
        const long long m = KS == 2 ? ((long long)nimg * p.ho + p.ostr * oy + p.par_a) * p.wo + p.ostr * ox + p.par_b
                                    : ((long long)nimg * p.ho + oy) * p.wo + ox;
#pragma unroll
        for (int i = 0; i < WCO; ++i) {
            const int co = co0 + w_co * (16 * WCO) + cgrp * WCO + i * 4;
            if (!pix_ok || co >= p.cout) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            const int nv = p.cout - co < 4 ? p.cout - co : 4;
            if (p.bias && !fold) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < nv) v[r] += p.bias[co + r];
            }
            if constexpr (ACT) {
                if (p.act == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_erf_f(v[r]);
                } else if (p.act == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + expf(-v[r]));
                } else if (p.act == 3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
                }
            }
            if (p.res && !fold) {
                const T* rp = (const T*)p.res + m * p.res_pitch + co;
                if (nv == 4) {
                    if constexpr (sizeof(T) == 2) {
                        half4 rv = *reinterpret_cast<const half4*>(rp);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
                    } else {
                        float4v rv = *reinterpret_cast<const float4v*>(rp);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += rv[r];
                    }
                } else {
                    for (int r = 0; r < nv; ++r) v[r] += to_f(rp[r]);
                }
            }
            T* op = (T*)p.out + m * p.cout_pitch + co;
            T tv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) tv[r] = from_f<T>(v[r]);
            ELVIS_HOOK_SKIP_STORE(tv)
            if (nv == 4) {
                if constexpr (sizeof(T) == 2) {
                    half4 hv = {tv[0], tv[1], tv[2], tv[3]};
                    *reinterpret_cast<half4*>(op) = hv;
                } else {
                    *reinterpret_cast<float4v*>(op) = (float4v){tv[0], tv[1], tv[2], tv[3]};
                }
            } else {
                for (int r = 0; r < nv; ++r) op[r] = tv[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r < nv) {
                    float f = to_f(tv[r]);   // statistics of the STORED value
                    st[i][r] += f;
                    sq[i][r] = fmaf(f, f, sq[i][r]);
                }
            }
        }
    }
    if (p.stats) {
        // reduce over the 16 pixel lanes that share lq, then over the NW_PX pixel waves via LDS
        float* red = reinterpret_cast<float*>(smem);  // [NW_PX][TCO][2]; all LDS reads are done
#pragma unroll
        for (int i = 0; i < WCO; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // the 16 pixel lanes of a channel group are one DPP row: rotate-and-add, no LDS traffic
                const float a = row16_sum(st[i][r]), b = row16_sum(sq[i][r]);
                if (lr == 0) {
                    int cl = w_co * (16 * WCO) + cgrp * WCO + i * 4 + r;
                    red[(w_px * TCO + cl) * 2 + 0] = a;
                    red[(w_px * TCO + cl) * 2 + 1] = b;
                }
            }
        __syncthreads();
        if (tid < TCO && co0 + tid < p.cout) {   // TCO <= 128 < NT
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < NW_PX; ++w) {
                a += red[(w * TCO + tid) * 2 + 0];
                b += red[(w * TCO + tid) * 2 + 1];
            }
            long long tile = ((long long)nimg * p.tiles_y + ty) * p.tiles_x + tx;   // (sub-pixel: p.stats is pre-offset per parity)
            float* dst = p.stats + (tile * p.cout + co0 + tid) * 2;
            dst[0] = a;
            dst[1] = b;
        }
    }
    ELVIS_HOOK_STAMP_END
}

template <typename T, int TCO, int NT, int TY, bool PRO, int KS = 3, bool ACT = false>
__global__ __launch_bounds__(NT, 2) void conv3x3_halo_kernel(ConvArgs p) {
    conv3x3_halo_body<T, TCO, NT, TY, PRO, KS, ACT, false>(p);
}
// fp32 storage, error-compensated f16 MFMA (ELVIS_F32X3; mma_tile_x): 512-thread kernels only
template <int TCO, int TY, bool PRO, int KS = 3, bool ACT = false>
__global__ __launch_bounds__(512, 2) void conv3x3_halo_x3_kernel(ConvArgs p) {
    conv3x3_halo_body<float, TCO, 512, TY, PRO, KS, ACT, true>(p);
}

// tile configuration chosen from cout (shared by pack + launch)
struct TileCfg {
    int tco, tpx, id;
};
inline TileCfg choose_tile(int cout) {
    if (cout % 128 == 0) return {128, 128, 0};
    if (cout >= 64) return {64, 128, 1};   // co padded up to a multiple of 64 (e.g. 160 -> 192)
    if (cout <= 16) return {16, 256, 3};
    return {32, 256, 2};
}
// the halo kernel handles 3x3 / stride 1 / pad 1 with a 128- or 64-channel output tile
inline bool halo_eligible(const elvis_conv_desc* d) {
    if (d->ksize == 2) return true;   // validated: sub-pixel parity conv
    bool same = d->ho == (d->upsample ? 2 * d->h : d->h) && d->wo == (d->upsample ? 2 * d->w : d->w);
    if (d->ksize == 3) return d->stride == 1 && d->pad_before == 1 && same;
    return d->ksize == 1 && d->stride == 1 && d->pad_before == 0 && same && !d->prologue &&
           (long long)d->ho * d->wo >= 256;   // tiny GEMMs (load-time embeddings) stay on the generic kernel; the choice
                                              // must not depend on n: the two kernels round differently (bias first / last)
}
// 512-thread workgroups.  Without the fused prologue: 16 x 32 pixel tile, 64co x 128px per wave
// (248 VGPRs).  With it: 8 x 32 tile, 64co x 64px per wave, leaving registers for the SiLU math.
constexpr int HALO_TY = 16, HALO_TY_PRO = 8, HALO_TY_PRO128 = 12, HALO_TX = 32;
#ifndef ELVIS_G1_TY128
#define ELVIS_G1_TY128 8
#endif
#ifndef ELVIS_G1_TY64
#define ELVIS_G1_TY64 8
#endif
// 256-thread variant, two or three workgroups per CU (f16; 3x3 and sub-pixel 2x2; 128- or 64-channel tile):
// 6 x 32 pixels x 128 channels (96px x 64co per wave) or 8 x 32 x 64 (64px x 64co per wave)
// 8 x 32 pixel tiles; the 64-channel tile with the fused prologue uses 16 rows on large images (128 px x
// 64 cout per wave: better fragment reuse), 8 rows where a taller tile would leave CUs without work
constexpr int HALO_TY2 = 8, HALO_TY2_TALL = 16;
inline bool halo_tall(const elvis_conv_desc* d) {
    return d->ksize == 3 && d->prologue && choose_tile(d->cout).tco == 64 &&
           (long long)d->ho * d->wo >= 128 * 1024;   // per image, NOT per batch: the tile shape sets the statistics' summation order
}
inline int halo_ty2(const elvis_conv_desc* d) { return halo_tall(d) ? HALO_TY2_TALL : HALO_TY2; }
// 1x1 convs are HBM/latency-bound: the 8-row tile halves LDS and registers so two workgroups fit a CU
// the fused-prologue kernel with a 128-channel tile uses 12 rows (64co x 96px per wave, ~210 VGPRs)
inline int kc_elems(int dtype) { return dtype == ELVIS_F16 ? 32 : 16; }
// two-workgroups-per-CU variant: f16, 3x3, 128-channel tile, LDS footprint <= 80 KB
inline bool halo_two(const elvis_conv_desc* d) {
    static const int mode = getenv("ELVIS_HALO2") ? atoi(getenv("ELVIS_HALO2")) : 1;   // 0 disables (A/B runs)
    const int tco = choose_tile(d->cout).tco, ks = d->ksize;
    if (!mode || (ks != 3 && ks != 2) || d->dtype != ELVIS_F16 || tco < 64) return false;
    if (mode == 2 && (ks != 3 || tco != 128)) return false;
    int nkc = (d->cin + 31) / 32 + (d->cin2 > 0 ? (d->cin2 + 31) / 32 : 0);
    size_t lds = (size_t)(halo_ty2(d) + ks - 1) * (HALO_TX + ks - 1) * 64 + 2 * ks * (size_t)tco * 64 + (d->prologue ? (size_t)nkc * 256 : 0);
    return lds <= 80 * 1024;
}
// 1x1 GEMM path with LDS-DMA staging: f16, whole 32-channel K chunks, 64/128-channel tile
inline bool halo_g1(const elvis_conv_desc* d) {
    static const int on = getenv("ELVIS_G1") ? atoi(getenv("ELVIS_G1")) : 1;   // 0 disables (A/B runs)
    return on && d->ksize == 1 && d->dtype == ELVIS_F16 && choose_tile(d->cout).tco >= 64 && d->cin % 32 == 0 &&
           d->cin2 % 32 == 0;
}
inline int halo_ty(const elvis_conv_desc* d) {
    if (halo_two(d)) return halo_ty2(d);
    if (halo_g1(d)) return choose_tile(d->cout).tco == 128 ? ELVIS_G1_TY128 : ELVIS_G1_TY64;
    if (d->ksize == 3 && d->prologue && choose_tile(d->cout).tco == 128) return HALO_TY_PRO128;
    return (d->prologue || d->ksize == 1) ? HALO_TY_PRO : HALO_TY;
}

int validate(const elvis_conv_desc* d) {
    ELVIS_REQUIRE(d, "conv: null descriptor");
    ELVIS_REQUIRE(d->dtype == ELVIS_F32 || d->dtype == ELVIS_F16 || d->dtype == ELVIS_F32X3, "conv: bad dtype %d", d->dtype);
    ELVIS_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0 && d->ho > 0 && d->wo > 0,
                  "conv: bad shape n=%d h=%d w=%d cin=%d cout=%d ho=%d wo=%d", d->n, d->h, d->w, d->cin, d->cout, d->ho, d->wo);
    ELVIS_REQUIRE(d->ksize == 1 || d->ksize == 3 || (d->ksize == 2 && d->subpixel >= 1 && d->subpixel <= 5),
                  "conv: ksize must be 1 or 3, or 2 with subpixel = 1 + parity / 5 (got ksize %d, subpixel %d)", d->ksize, d->subpixel);
    const bool s2d = d->ksize == 2 && d->subpixel == ELVIS_CONV_S2D;
    if (d->ksize == 2 && !s2d)
        ELVIS_REQUIRE(d->stride == 1 && !d->upsample && !d->prologue && d->ho == 2 * d->h && d->wo == 2 * d->w,
                      "conv: a sub-pixel parity conv maps h x w to 2h x 2w, stride 1, no prologue");
    if (s2d)
        ELVIS_REQUIRE(!d->upsample && !d->prologue && d->cin2 == 0 && d->h == 2 * d->ho && d->w == 2 * d->wo && d->cin % 128 == 0 &&
                          d->dtype == ELVIS_F16 && d->cout >= 64,
                      "conv: the space-to-depth stride-2 form maps 2ho x 2wo to ho x wo, f16, cin = 4 x (a multiple of 32), cout >= 64");
    ELVIS_REQUIRE(d->stride == 1 || d->stride == 2, "conv: stride must be 1 or 2");
    ELVIS_REQUIRE(d->cin_pitch >= (s2d ? d->cin / 4 : d->cin) && d->cin_pitch % 8 == 0, "conv: cin_pitch %d must be >= cin %d and a multiple of 8", d->cin_pitch, d->cin);
    ELVIS_REQUIRE(d->cout_pitch >= d->cout && d->cout_pitch % 4 == 0, "conv: cout_pitch %d invalid for cout %d", d->cout_pitch, d->cout);
    int kc = kc_elems(d->dtype);
    if (d->cin2 > 0) {
        ELVIS_REQUIRE(d->cin % kc == 0, "conv: with a second input, cin (%d) must be a multiple of %d", d->cin, kc);
        ELVIS_REQUIRE(d->cin2_pitch >= d->cin2 && d->cin2_pitch % 8 == 0, "conv: bad cin2_pitch");
    }
    int lh = d->upsample ? 2 * d->h : d->h, lw = d->upsample ? 2 * d->w : d->w;
    // last tap of the last output must start inside [-(pad), l+2): loose sanity bound
    ELVIS_REQUIRE(d->ksize == 2 ||
                  ((long long)(d->ho - 1) * d->stride - d->pad_before < lh && (long long)(d->wo - 1) * d->stride - d->pad_before < lw),
                  "conv: output %dx%d does not fit input %dx%d (stride %d)", d->ho, d->wo, lh, lw, d->stride);
    return ELVIS_OK;
}

template <typename T, int WCO, int WPX, int NW_CO, int NW_PX>
int launch(const ConvArgs& a, hipStream_t stream) {
    constexpr int TCO = 16 * WCO * NW_CO, TPX = 16 * WPX * NW_PX;
    size_t lds = 2 * (size_t)(TCO + TPX) * 64;
    long long nblk = (long long)a.n_co_tiles * a.n_px_tiles;
    ELVIS_REQUIRE(nblk < 0x7fffffffLL, "conv: grid too large");
    hipLaunchKernelGGL((conv_igemm_kernel<T, WCO, WPX, NW_CO, NW_PX>), dim3((unsigned)nblk), dim3(64 * NW_CO * NW_PX),
                       lds, stream, a);
    ELVIS_CHECK_LAUNCH("elvis_conv2d");
    return ELVIS_OK;
}

template <typename T, int TCO, bool PRO, int KS, int NT = 512, bool ACT = false, int TY2 = HALO_TY2, bool X3 = false>
int launch_halo_p(const ConvArgs& a, hipStream_t stream) {
    if constexpr (!ACT) {
        if (a.act != 0) return launch_halo_p<T, TCO, PRO, KS, NT, true, TY2, X3>(a, stream);
    }
    // fp32 storage with the compensated f16 MFMA: the 512-thread kernels with a 64- / 128-channel tile
    // (narrower layers stay on the exact fp32 MFMA)
    if constexpr (!X3 && sizeof(T) == 4 && NT == 512 && TCO >= 64) {
        if (a.x3) return launch_halo_p<T, TCO, PRO, KS, NT, ACT, TY2, true>(a, stream);
    }
    constexpr bool TWO = NT == 256;
    constexpr int TY = (TWO && KS == 1) ? (TCO == 128 ? ELVIS_G1_TY128 : ELVIS_G1_TY64) : TWO ? TY2 : (KS == 3 && PRO && TCO == 128) ? HALO_TY_PRO128 : ((PRO || KS == 1) ? HALO_TY_PRO : HALO_TY);
    const size_t lds_fixed = (TWO && KS == 1) ? (TCO == 64 ? ELVIS_G1_NST64 : ELVIS_G1_NST128) * ((size_t)TY * HALO_TX * 64 + (size_t)TCO * 64)
                           : (TWO ? 1 : 2) * (size_t)((TY + KS - 1) * (HALO_TX + KS - 1) * 64) + ((TWO || KS == 2) ? 2 : 3) * KS * (size_t)TCO * 64;
    const size_t lds = lds_fixed + (PRO ? (size_t)a.nkc * 4 * 2 * DT<T>::VEC * sizeof(float) : 0);
    ELVIS_REQUIRE(lds <= 160 * 1024, "conv3x3_halo: %zu bytes of LDS needed (too many input channels)", lds);
    {   // the 160 KB opt-in is a per-device function attribute: set it once per (instantiation, device),
        // under a lock - P2 calls this from pool threads, one per device (elvis.py:342-346)
        static std::mutex mu;
        static bool attr_set[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        std::lock_guard<std::mutex> guard(mu);
        if (!attr_set[dev]) {
            const void* fn = nullptr;
            if constexpr (X3) fn = (const void*)conv3x3_halo_x3_kernel<TCO, TY, PRO, KS, ACT>;
            else fn = (const void*)conv3x3_halo_kernel<T, TCO, NT, TY, PRO, KS, ACT>;
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) {
                elvis_set_error("conv3x3_halo: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
                return ELVIS_E_RUNTIME;
            }
            attr_set[dev] = true;
        }
    }
    long long nblk = (long long)a.n_co_tiles * a.tiles_x * a.tiles_y * a.n;
    ELVIS_REQUIRE(nblk < 0x7fffffffLL, "conv: grid too large");
    if constexpr (X3) hipLaunchKernelGGL((conv3x3_halo_x3_kernel<TCO, TY, PRO, KS, ACT>), dim3((unsigned)nblk), dim3(NT), lds, stream, a);
    else hipLaunchKernelGGL((conv3x3_halo_kernel<T, TCO, NT, TY, PRO, KS, ACT>), dim3((unsigned)nblk), dim3(NT), lds, stream, a);
    ELVIS_CHECK_LAUNCH("elvis_conv2d(halo)");
    return ELVIS_OK;
}

template <typename T, int TCO> int launch_halo(const ConvArgs& a, hipStream_t stream) {
    if constexpr (TCO >= 64 && sizeof(T) == 2) {
        if (a.two && a.ksize == 1) return launch_halo_p<T, TCO, false, 1, 256>(a, stream);
    }
    if (a.ksize == 1) return launch_halo_p<T, TCO, false, 1>(a, stream);
    if constexpr (TCO >= 64 && sizeof(T) == 2) {
        if (a.two && a.ksize == 2) return launch_halo_p<T, TCO, false, 2, 256>(a, stream);
        if constexpr (TCO == 64) {
            if (a.two && a.prologue && a.tall) return launch_halo_p<T, TCO, true, 3, 256, false, HALO_TY2_TALL>(a, stream);
        }
        if (a.two) return a.prologue ? launch_halo_p<T, TCO, true, 3, 256>(a, stream) : launch_halo_p<T, TCO, false, 3, 256>(a, stream);
    }
    if (a.ksize == 2) return launch_halo_p<T, TCO, false, 2>(a, stream);
    return a.prologue ? launch_halo_p<T, TCO, true, 3>(a, stream) : launch_halo_p<T, TCO, false, 3>(a, stream);
}

template <typename T> int dispatch(const ConvArgs& a, int id, hipStream_t stream) {
    switch (id) {
        case 0: return launch<T, 4, 4, 2, 2>(a, stream);
        case 1: return launch<T, 4, 2, 1, 4>(a, stream);   // 64 cout per wave, like the halo kernels (shared weight permutation)
        case 2: return launch<T, 2, 4, 1, 4>(a, stream);
        default: return launch<T, 1, 4, 1, 4>(a, stream);
    }
}

// ---- weight packing: OIHW f32 -> [tap][kc][co_pad][KC] (T)
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ out, int cout, int ctot, int ks,
                                    int nkc, int co_pad, int KC, long long total, int wco, int pairs = 0) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int k = (int)(i % KC);
    long long t = i / KC;
    // Packed row -> output channel.  Within a wave's group of G = 16*wco channels the rows are permuted
    // so that MFMA fragment i, row 4*lq + r (what lane-quarter lq holds in accumulator register r of
    // sub-tile i) is channel lq*4*wco + i*4 + r: a lane then owns 4*wco CONTIGUOUS channels of a pixel
    // and the epilogue moves them with 16-byte accesses on whole cache lines.
    int row = (int)(t % co_pad);
    const int G = 16 * wco;
    const int rho = row % G, fi = rho / 16, q = (rho % 16) / 4, rr = rho % 4;
    int co = (row / G) * G + q * 4 * wco + fi * 4 + rr;
    t /= co_pad;
    int kc = (int)(t % nkc);
    int tap = (int)(t / nkc);
    int ci = kc * KC + k;
    float v = 0.f;
    if (co < cout && ci < ctot) v = w[((long long)co * ctot + ci) * ks * ks + tap];
    if constexpr (sizeof(T) == 4) {
        if (pairs) {   // ELVIS_F32X3: (hi, lo) half pair in the fp32 slot
            reinterpret_cast<unsigned*>(out)[i] = x3_pair(v);
            return;
        }
    }
    out[i] = from_f<T>(v);
}

}  // namespace

static void conv_geom(const elvis_conv_desc* d, int* nkc1, int* nkc, int* co_pad) {
    int kc = kc_elems(d->dtype);
    *nkc1 = (d->cin + kc - 1) / kc;
    *nkc = *nkc1 + (d->cin2 > 0 ? (d->cin2 + kc - 1) / kc : 0);
    TileCfg t = choose_tile(d->cout);
    *co_pad = ((d->cout + t.tco - 1) / t.tco) * t.tco;
}

extern "C" size_t elvis_conv_packed_weight_bytes(const elvis_conv_desc* d) {
    if (!d || d->cin <= 0 || d->cout <= 0) return 0;
    int nkc1, nkc, co_pad;
    conv_geom(d, &nkc1, &nkc, &co_pad);
    return (size_t)d->ksize * d->ksize * nkc * co_pad * 64;
}

extern "C" int elvis_conv_pack_weights(const elvis_conv_desc* d, const float* w_oihw, void* packed,
                                       elvis_stream_t stream) {
    int rc = validate(d);
    if (rc) return rc;
    ELVIS_REQUIRE(w_oihw && packed, "elvis_conv_pack_weights: null pointer");
    int nkc1, nkc, co_pad;
    conv_geom(d, &nkc1, &nkc, &co_pad);
    int KC = kc_elems(d->dtype);
    // with two inputs the packed K axis is [cin padded to nkc1*KC | cin2]; cin % KC == 0 is
    // enforced in that case so the source channel index is simply ci.
    long long total = (long long)d->ksize * d->ksize * nkc * co_pad * KC;
    int grid = (int)((total + 255) / 256);
    int ctot = d->cin + d->cin2;
    const int tco_ = choose_tile(d->cout).tco, wco = tco_ >= 64 ? 4 : tco_ / 16;   // 16-row fragments per wave (all kernels agree)
    if (d->dtype == ELVIS_F16)
        hipLaunchKernelGGL(pack_weights_kernel<half_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw,
                           (half_t*)packed, d->cout, ctot, d->ksize, nkc, co_pad, KC, total, wco);
    else
        hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw,
                           (float*)packed, d->cout, ctot, d->ksize, nkc, co_pad, KC, total, wco, d->dtype == ELVIS_F32X3 ? 1 : 0);
    ELVIS_CHECK_LAUNCH("elvis_conv_pack_weights");
    return ELVIS_OK;
}

extern "C" int elvis_conv_stats_tiles(const elvis_conv_desc* d) {
    if (!d || !halo_eligible(d)) return 0;
    int ty = halo_ty(d);
    if (d->ksize == 2 && d->subpixel != ELVIS_CONV_S2D) return d->n * ((d->h + ty - 1) / ty) * ((d->w + HALO_TX - 1) / HALO_TX);   // per parity launch
    return d->n * ((d->ho + ty - 1) / ty) * ((d->wo + HALO_TX - 1) / HALO_TX);
}

// Does a descriptor with dtype ELVIS_F32X3 run on a compensated-f16 kernel?  (Its weights are packed as (hi, lo)
// half pairs, which only those kernels read: a host keeps the ELVIS_F32 packing for everything else.)
static bool x3_eligible(const elvis_conv_desc* d) {
    return d->dtype == ELVIS_F32X3 && halo_eligible(d) && !getenv("ELVIS_NO_HALO") && choose_tile(d->cout).tco >= 64;
}
extern "C" int elvis_conv_x3_eligible(const elvis_conv_desc* d) {
    if (!d || validate(d)) return 0;
    return x3_eligible(d) ? 1 : 0;
}

extern "C" int elvis_conv_kernel_name(const elvis_conv_desc* d, char* buf, size_t n) {
    int rc = validate(d);
    if (rc) return rc;
    ELVIS_REQUIRE(buf && n > 0, "elvis_conv_kernel_name: null buffer");
    const char* t = d->dtype == ELVIS_F16 ? "half" : "float";
    TileCfg c = choose_tile(d->cout);
    if (halo_eligible(d) && !getenv("ELVIS_NO_HALO")) {
        const bool pro = d->ksize == 3 && d->prologue;
        if (d->dtype == ELVIS_F32X3 && c.tco >= 64)
            snprintf(buf, n, "conv3x3_halo_x3_kernel<%d,%d,%s,%d,%s>", c.tco, halo_ty(d), pro ? "true" : "false", d->ksize,
                     d->act ? "true" : "false");
        else
        snprintf(buf, n, "conv3x3_halo_kernel<%s,%d,%d,%d,%s,%d,%s>", t, c.tco, (halo_two(d) || halo_g1(d)) ? 256 : 512, halo_ty(d),
                 pro ? "true" : "false", d->ksize, d->act ? "true" : "false");
    } else {
        static const int cfg[4][4] = {{4, 4, 2, 2}, {4, 2, 1, 4}, {2, 4, 1, 4}, {1, 4, 1, 4}};
        snprintf(buf, n, "conv_igemm_kernel<%s,%d,%d,%d,%d>", t, cfg[c.id][0], cfg[c.id][1], cfg[c.id][2], cfg[c.id][3]);
    }
    return ELVIS_OK;
}

extern "C" int elvis_conv2d(const elvis_conv_desc* d, const void* x, const void* x2, const void* w_packed,
                            const float* bias, const void* residual, int residual_pitch, const float* pa,
                            const float* pb, void* out, float* stats, elvis_stream_t stream) {
    int rc = validate(d);
    if (rc) return rc;
    ELVIS_REQUIRE(x && w_packed && out, "elvis_conv2d: null pointer");
    ELVIS_REQUIRE(d->cin2 == 0 || x2, "elvis_conv2d: cin2 > 0 but x2 is null");
    ELVIS_REQUIRE(!d->prologue || (pa && pb), "elvis_conv2d: prologue requested without pa/pb");
    ELVIS_REQUIRE(d->dtype != ELVIS_F32X3 || x3_eligible(d),
                  "elvis_conv2d: this shape has no compensated-f16 kernel (elvis_conv_x3_eligible): run it as ELVIS_F32");
    ELVIS_REQUIRE(!residual || residual_pitch >= d->cout, "elvis_conv2d: bad residual pitch");
    ELVIS_REQUIRE(((uintptr_t)x | (uintptr_t)(x2 ? x2 : x) | (uintptr_t)w_packed | (uintptr_t)out) % 16 == 0,
                  "elvis_conv2d: pointers must be 16-byte aligned");
    ConvArgs a;
    a.x = x; a.x2 = x2; a.w = w_packed; a.bias = bias; a.res = residual; a.pa = pa; a.pb = pb; a.out = out;
    a.n = d->n; a.h = d->h; a.w_in = d->w;
    a.cin = d->cin; a.cin_pitch = d->cin_pitch; a.cin2 = d->cin2; a.cin2_pitch = d->cin2_pitch;
    a.cout = d->cout; a.cout_pitch = d->cout_pitch; a.res_pitch = residual_pitch;
    a.ksize = d->ksize; a.stride = d->stride; a.pad = d->pad_before; a.upsample = d->upsample;
    a.ho = d->ho; a.wo = d->wo; a.act = d->act; a.prologue = d->prologue;
    conv_geom(d, &a.nkc1, &a.nkc, &a.co_pad);
    TileCfg t = choose_tile(d->cout);
    a.M = (long long)d->n * d->ho * d->wo;
    a.n_co_tiles = a.co_pad / t.tco;
    a.n_px_tiles = (a.M + t.tpx - 1) / t.tpx;
    a.stats = stats;
    const bool s2d = d->ksize == 2 && d->subpixel == ELVIS_CONV_S2D;
    const bool subpix = d->ksize == 2 && !s2d;
    a.sub = d->ksize == 2;
    a.par_a = subpix ? (d->subpixel - 1) >> 1 : 0;
    a.par_b = subpix ? (d->subpixel - 1) & 1 : 0;
    a.ostr = subpix ? 2 : 1;
    a.pad2y = subpix ? 1 - a.par_a : 0;
    a.pad2x = subpix ? 1 - a.par_b : 0;
    a.s2d = s2d ? 1 : 0;
    a.istr = s2d ? 2 : 1;
    a.nkc_c = s2d ? d->cin / 4 / 32 : 0x3fffffff;
    a.wfull = d->w;
    if (s2d) {   // the kernel sees the phase image: ho x wo pixels of 4C channels
        a.h = d->ho;
        a.w_in = d->wo;
    }
    a.tiles_x = ((subpix ? d->w : d->wo) + HALO_TX - 1) / HALO_TX;
    {
        const char* e = getenv("ELVIS_STRIP");   // A/B switch (read per call); 0 = row-major walk
        a.strip = e ? atoi(e) : 8;
        if (a.strip < 0 || a.strip >= a.tiles_x) a.strip = 0;
        a.strip_full = a.strip > 0 ? a.tiles_x / a.strip : 0;
    }
    const int tyv = halo_ty(d);
    a.x3 = d->dtype == ELVIS_F32X3 ? 1 : 0;
    a.two = (halo_two(d) || halo_g1(d)) ? 1 : 0;
    a.tall = (halo_two(d) && halo_tall(d)) ? 1 : 0;
    a.tiles_y = ((subpix ? d->h : d->ho) + tyv - 1) / tyv;
    if (halo_eligible(d) && !getenv("ELVIS_NO_HALO")) {
        ELVIS_REQUIRE((long long)d->n * d->h * d->w < 0x7fffffffLL, "conv: input too large for 32-bit pixel indices");
        hipStream_t st = (hipStream_t)stream;
        if (d->dtype == ELVIS_F16) {
            switch (t.tco) {
                case 128: return launch_halo<half_t, 128>(a, st);
                case 64: return launch_halo<half_t, 64>(a, st);
                case 32: return launch_halo<half_t, 32>(a, st);
                default: return launch_halo<half_t, 16>(a, st);
            }
        }
        switch (t.tco) {
            case 128: return launch_halo<float, 128>(a, st);
            case 64: return launch_halo<float, 64>(a, st);
            case 32: return launch_halo<float, 32>(a, st);
            default: return launch_halo<float, 16>(a, st);
        }
    }
    ELVIS_REQUIRE(!stats, "elvis_conv2d: fused statistics need a 3x3/stride-1 conv with cout >= 64 (query elvis_conv_stats_tiles)");
    if (d->dtype == ELVIS_F16) return dispatch<half_t>(a, t.id, (hipStream_t)stream);
    return dispatch<float>(a, t.id, (hipStream_t)stream);
}
