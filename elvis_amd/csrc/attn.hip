// Swin (shifted-)window attention on an NHWC token image: 64 tokens x head_dim 32 per (window,
// head).  QK^T + relative-position bias + shift mask -> softmax -> PV.  f16 tensors run on the
// matrix cores (window_attention_mfma_kernel below); the exact-parity f32 mode runs in fp32 VALU on
// LDS-resident tiles; the cyclic shift / window partition / reverse are pure index math
// (tokens never leave image order in HBM).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int WS = 8, NTOK = 64, HD = 32;
constexpr int QP = HD + 1;    // padded row pitch (floats) for Q/K/V tiles
constexpr int PP = NTOK + 1;  // padded row pitch for P

template <typename T>
__global__ __launch_bounds__(256) void window_attention_kernel(const T* __restrict__ qkv, T* __restrict__ out, int h,
                                                               int w, int heads, int shift, int qkv_pitch,
                                                               int out_pitch, const float* __restrict__ bias_table,
                                                               float scale) {
    __shared__ float sq[NTOK * QP], sk[NTOK * QP], sv[NTOK * QP], sp[NTOK * PP];
    __shared__ int s_pos[NTOK];     // original pixel index of each window token
    __shared__ int s_region[NTOK];  // shift-mask region id
    const int E = heads * HD;
    const int nwx = w / WS, nwy = h / WS;
    int bid = blockIdx.x;
    const int head = bid % heads;
    bid /= heads;
    const int wx = bid % nwx;
    bid /= nwx;
    const int wy = bid % nwy;
    const int n = bid / nwy;
    const int tid = threadIdx.x;

    if (tid < NTOK) {
        int ty = tid / WS, tx = tid % WS;
        int yr = wy * WS + ty, xr = wx * WS + tx;  // coordinates in the rolled image
        int y = yr + shift, x = xr + shift;
        if (y >= h) y -= h;
        if (x >= w) x -= w;
        s_pos[tid] = (n * h + y) * w + x;
        int rh = 0, rw = 0;
        if (shift) {
            rh = yr < h - WS ? 0 : (yr < h - shift ? 1 : 2);
            rw = xr < w - WS ? 0 : (xr < w - shift ? 1 : 2);
        }
        s_region[tid] = rh * 3 + rw;
    }
    __syncthreads();
    // stage q (pre-scaled), k, v: 64 tokens x 32 dims each
    for (int i = tid; i < NTOK * HD; i += 256) {
        int t = i / HD, d = i % HD;
        const T* src = qkv + (long long)s_pos[t] * qkv_pitch + head * HD + d;
        sq[t * QP + d] = to_f(src[0]) * scale;
        sk[t * QP + d] = to_f(src[E]);
        sv[t * QP + d] = to_f(src[2 * E]);
    }
    __syncthreads();
    const int row = tid >> 2, part = tid & 3;
    float q[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) q[d] = sq[row * QP + d];
    float s[16];
    const int yi = row / WS, xi = row % WS;
    const int reg_i = s_region[row];
    float mx = -3.0e38f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        int j = part * 16 + jj;
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) a = fmaf(q[d], sk[j * QP + d], a);
        int yj = j / WS, xj = j % WS;
        a += bias_table[((yi - yj + WS - 1) * (2 * WS - 1) + (xi - xj + WS - 1)) * heads + head];
        if (shift && s_region[j] != reg_i) a += -100.0f;
        s[jj] = a;
        mx = fmaxf(mx, a);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    float sum = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        s[jj] = expf(s[jj] - mx);
        sum += s[jj];
    }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    float inv = 1.0f / sum;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) sp[row * PP + part * 16 + jj] = s[jj] * inv;
    __syncthreads();
    float o[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) o[d] = 0.f;
    for (int j = 0; j < NTOK; ++j) {
        float pj = sp[row * PP + j];
#pragma unroll
        for (int d = 0; d < 8; ++d) o[d] = fmaf(pj, sv[j * QP + part * 8 + d], o[d]);
    }
    T* dst = out + (long long)s_pos[row] * out_pitch + head * HD + part * 8;
#pragma unroll
    for (int d = 0; d < 8; ++d) dst[d] = from_f<T>(o[d]);
}


// ------------------------------------------------------------------------------------------
// MFMA form (f16 tensors): one wave per (window, head).
//   S[i][j] = sum_d Q[i][d] K[j][d]      16 x v_mfma_f32_16x16x32_f16, Q/K fragments straight from
//                                        global memory (a token's 32-d head slice is 64 contiguous bytes)
//   P = softmax_j(S*scale + bias + mask) in registers: a row lives in 4 registers x 16 lanes,
//                                        reduced with DPP/shuffle xor 1,2,4,8
//   O[i][d] = sum_j P[i][j] V[j][d]      16 MFMAs; P goes through a per-wave LDS tile to become an A
//                                        operand (row = token, k = key contiguous), V through a
//                                        LDS tile stored transposed ([d][key]): one ds_read_b128 per fragment
constexpr int P_PITCH = NTOK + 8;   // halfs; 144-byte rows keep ds_read_b128 16-byte aligned
constexpr int V_PITCH = NTOK + 8;   // halfs; V is kept TRANSPOSED ([d][key], 144-byte rows) so a B fragment is one ds_read_b128

#ifndef ELVIS_ATT_WAVES
#define ELVIS_ATT_WAVES 2   /* waves per workgroup (= heads in flight per window): 2 -> 30 KB of LDS, five workgroups per CU (-26 % vs 4) */
#endif
constexpr int ATT_NW = ELVIS_ATT_WAVES;
__global__ __launch_bounds__(64 * ATT_NW) void window_attention_mfma_kernel(const half_t* __restrict__ qkv,
                                                                    half_t* __restrict__ out, int h, int w,
                                                                    int heads, int shift, int qkv_pitch,
                                                                    int out_pitch,
                                                                    const float* __restrict__ bias_table,
                                                                    float scale) {
    __shared__ __attribute__((aligned(16))) half_t sP[ATT_NW][NTOK * P_PITCH];
    __shared__ __attribute__((aligned(16))) half_t sV[ATT_NW][HD * V_PITCH];
    __shared__ float sBias[ATT_NW][(2 * WS - 1) * (2 * WS - 1)];
    __shared__ int s_pos[NTOK];
    __shared__ int s_region[NTOK];
    const int E = heads * HD;
    const int nwx = w / WS, nwy = h / WS;
    int bid = blockIdx.x;
    const int wx = bid % nwx;
    bid /= nwx;
    const int wy = bid % nwy;
    const int n = bid / nwy;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    if (tid < NTOK) {
        int ty = tid / WS, tx = tid % WS;
        int yr = wy * WS + ty, xr = wx * WS + tx;
        int y = yr + shift, x = xr + shift;
        if (y >= h) y -= h;
        if (x >= w) x -= w;
        s_pos[tid] = (n * h + y) * w + x;
        int rh = 0, rw = 0;
        if (shift) {
            rh = yr < h - WS ? 0 : (yr < h - shift ? 1 : 2);
            rw = xr < w - WS ? 0 : (xr < w - shift ? 1 : 2);
        }
        s_region[tid] = rh * 3 + rw;
    }
    __syncthreads();

    const int lr = lane & 15, lq = lane >> 4;
    half_t* myP = sP[wave];
    half_t* myV = sV[wave];
    float* myB = sBias[wave];

    for (int head = wave; head < heads; head += ATT_NW) {
        // relative-position bias column of this head -> LDS
        for (int t = lane; t < (2 * WS - 1) * (2 * WS - 1); t += 64) myB[t] = bias_table[t * heads + head];
        // V tile (64 keys x 32 dims) -> LDS, one 16-byte chunk per lane-iteration
        for (int t = lane; t < NTOK * 4; t += 64) {
            int j = t >> 2, c = t & 3;
            half8 v8 = *reinterpret_cast<const half8*>(qkv + (long long)s_pos[j] * qkv_pitch + 2 * E + head * HD + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) myV[(c * 8 + e) * V_PITCH + j] = v8[e];
        }
        // Q (A operand) and K (B operand) fragments straight from global memory
        half8 fq[4], fk[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int tok = t * 16 + lr;
            fq[t] = *reinterpret_cast<const half8*>(qkv + (long long)s_pos[tok] * qkv_pitch + head * HD + lq * 8);
            fk[t] = *reinterpret_cast<const half8*>(qkv + (long long)s_pos[tok] * qkv_pitch + E + head * HD + lq * 8);
        }
        float4v sacc[4][4];
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                sacc[it][jt] = (float4v){0.f, 0.f, 0.f, 0.f};
                sacc[it][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fq[it], fk[jt], sacc[it][jt], 0, 0, 0);
            }
        // softmax over j for each row i = it*16 + lq*4 + r ; this lane holds columns j = jt*16 + lr
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = it * 16 + lq * 4 + r;
                const int yi = i / WS, xi = i % WS;
                const int reg_i = s_region[i];
                float v[4];
                float mx = -3.0e38f;
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {
                    const int j = jt * 16 + lr;
                    const int yj = j / WS, xj = j % WS;
                    float a = sacc[it][jt][r] * scale + myB[(yi - yj + WS - 1) * (2 * WS - 1) + (xi - xj + WS - 1)];
                    if (shift && s_region[j] != reg_i) a += -100.0f;
                    v[jt] = a;
                    mx = fmaxf(mx, a);
                }
                mx = row16_max(mx);   // the 16 lanes holding a row's columns are one DPP row
                float sum = 0.f;
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {
                    v[jt] = __expf(v[jt] - mx);
                    sum += v[jt];
                }
                sum = row16_sum(sum);
                const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) myP[i * P_PITCH + jt * 16 + lr] = (half_t)(v[jt] * inv);
            }
        // (wave-local LDS tile: no workgroup barrier needed, only the LDS writes to land)
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        // O = P V : A = P[i][j] (rows on lr, k contiguous), B[k=j][col=d] from the V tile
        float4v oacc[4][2];
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) oacc[it][dt] = (float4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 fv[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                fv[dt] = *reinterpret_cast<const half8*>(myV + (dt * 16 + lr) * V_PITCH + ks * 32 + lq * 8);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                half8 fp = *reinterpret_cast<const half8*>(myP + (it * 16 + lr) * P_PITCH + ks * 32 + lq * 8);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    oacc[it][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fp, fv[dt], oacc[it][dt], 0, 0, 0);
            }
        }
        // store: lane holds O[i = it*16 + lq*4 + r][d = dt*16 + lr].  Transpose through the (now idle) P
        // tile so that every token's 64-byte head slice leaves as four 16-byte stores, not 32 2-byte ones.
        __builtin_amdgcn_s_waitcnt(0xC07F);   // this wave's P reads are done
        __builtin_amdgcn_wave_barrier();
        constexpr int O_PITCH = HD + 8;       // halfs; 80-byte rows: 16-byte aligned chunks
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = it * 16 + lq * 4 + r;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) myP[i * O_PITCH + dt * 16 + lr] = (half_t)oacc[it][dt][r];
            }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pc = lane + 64 * k, tok = pc >> 2, c8 = pc & 3;
            *reinterpret_cast<half8*>(out + (long long)s_pos[tok] * out_pitch + head * HD + c8 * 8) =
                *reinterpret_cast<const half8*>(myP + tok * O_PITCH + c8 * 8);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();   // the next head reuses this wave's LDS tiles
    }
}

// ------------------------------------------------------------------------------------------
// Round 3: the same attention with P and O kept in REGISTERS (window_attention_mfma_kernel above moves P, V and the
// output tile through LDS in 2-byte pieces: 0.32 of its LDS cycles are bank conflicts, 0.34 of the HBM peak).
//   S^T[key][token] = K Q^T        (A = K, B = Q, both straight from global memory): a lane holds, for its token column,
//                                   16 keys in registers and the rest in the lanes 16 / 32 / 48 away - the softmax over
//                                   keys is 16 in-lane values + two cross-lane steps, no DPP row reductions
//   P^T as B operand               two stacked 16-key accumulator tiles are one 32-deep B fragment (the key order inside
//                                   it is a permutation; the same one is applied to V's rows when its fragments are read)
//   O^T[d][token] = V^T P^T        A = V^T through ONE LDS tile stored row-major ([key][32 d], 16-byte writes) and read with
//                                   ds_read_b64_tr_b16 (hardware transpose); the block columns a lane group reads are
//                                   d = 8p + 4dt + (0..3), so accumulator rows 4q + r of the two d tiles are d = 8q .. 8q + 7:
//                                   a token's 64-byte head slice leaves as four 16-byte stores straight from registers
// The 8-byte halves of a V row's 16-byte chunks are swapped on rows with bit 2 set: the two 4-row blocks a 32-lane half
// reads (keys 4 apart) then fall on disjoint banks.
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ __launch_bounds__(64 * ATT_NW) void window_attention_tr_kernel(const half_t* __restrict__ qkv,
                                                                  half_t* __restrict__ out, int h, int w,
                                                                  int heads, int shift, int qkv_pitch,
                                                                  int out_pitch,
                                                                  const float* __restrict__ bias_table,
                                                                  float scale) {
    __shared__ __attribute__((aligned(16))) half_t sV[ATT_NW][NTOK * HD];
    __shared__ float sBias[ATT_NW][(2 * WS - 1) * (2 * WS - 1)];
    __shared__ int s_pos[NTOK];
    __shared__ int s_region[NTOK];
    const int E = heads * HD;
    const int nwx = w / WS, nwy = h / WS;
    int bid = blockIdx.x;
    const int wx = bid % nwx;
    bid /= nwx;
    const int wy = bid % nwy;
    const int n = bid / nwy;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    if (tid < NTOK) {
        int ty = tid / WS, tx = tid % WS;
        int yr = wy * WS + ty, xr = wx * WS + tx;
        int y = yr + shift, x = xr + shift;
        if (y >= h) y -= h;
        if (x >= w) x -= w;
        s_pos[tid] = (n * h + y) * w + x;
        int rh = 0, rw = 0;
        if (shift) {
            rh = yr < h - WS ? 0 : (yr < h - shift ? 1 : 2);
            rw = xr < w - WS ? 0 : (xr < w - shift ? 1 : 2);
        }
        s_region[tid] = rh * 3 + rw;
    }
    __syncthreads();

    const int lr = lane & 15, lq = lane >> 4;
    // shifted windows: only the last row / column of windows holds tokens of more than one region (every other window's
    // tokens are all region 0): the mask compare + select per score runs for those windows alone
    const bool need_mask = shift && (wy == nwy - 1 || wx == nwx - 1);
    half_t* myV = sV[wave];
    float* myB = sBias[wave];
    // transposed-read addressing: lane 4 qq + p of a 16-lane group supplies block row qq, columns 4p .. 4p+3
    const int qq = (lane & 15) >> 2, pp = lane & 3;
    const int swz = lq & 1;                                 // bit 2 of the key rows 32 m + 16 s + 4 lq + qq this lane addresses

    for (int head = wave; head < heads; head += ATT_NW) {
        // the softmax runs in the exp2 domain: the bias column is stored times log2(e), the scale carries the same factor
        for (int t = lane; t < (2 * WS - 1) * (2 * WS - 1); t += 64) myB[t] = bias_table[t * heads + head] * 1.4426950408889634f;
        // V tile (64 keys x 32 dims) -> LDS, row-major, 16-byte writes
        for (int t = lane; t < NTOK * 4; t += 64) {
            const int j = t >> 2, c = t & 3;
            uint4 v = *reinterpret_cast<const uint4*>(qkv + (long long)s_pos[j] * qkv_pitch + 2 * E + head * HD + c * 8);
            if (j & 4) v = make_uint4(v.z, v.w, v.x, v.y);
            *reinterpret_cast<uint4*>(myV + j * HD + c * 8) = v;
        }
        half8 fq[4], fk[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int tok = t * 16 + lr;
            fq[t] = *reinterpret_cast<const half8*>(qkv + (long long)s_pos[tok] * qkv_pitch + head * HD + lq * 8);
            fk[t] = *reinterpret_cast<const half8*>(qkv + (long long)s_pos[tok] * qkv_pitch + E + head * HD + lq * 8);
        }
        float4v sacc[4][4];   // [key tile][token tile]: rows = keys jt*16 + 4 lq + r, column = token it*16 + lr
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                sacc[jt][it] = (float4v){0.f, 0.f, 0.f, 0.f};
                sacc[jt][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fk[jt], fq[it], sacc[jt][it], 0, 0, 0);
            }
        __builtin_amdgcn_s_waitcnt(0xC07F);   // the bias column and the V tile have landed (wave-local LDS)
        __builtin_amdgcn_wave_barrier();
        half8 pb[4][2];       // P^T as B fragments: [token tile][32-key block] - UNNORMALISED exp2(a - max) in (0, 1]; the 1 / sum
        float inv[4];         // of a token scales its 8 output values instead of its 64 probabilities
        const float scale2 = scale * 1.4426950408889634f;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = it * 16 + lr;
            const int yi = i / WS, xi = i % WS;
            const int reg_i = s_region[i];
            float v[4][4];
            float mx = -3.0e38f;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = jt * 16 + lq * 4 + r;
                    const int yj = j / WS, xj = j % WS;
                    float a = fmaf(sacc[jt][it][r], scale2, myB[(yi - yj + WS - 1) * (2 * WS - 1) + (xi - xj + WS - 1)]);
                    if (need_mask && s_region[j] != reg_i) a += -144.26950408889634f;   // -100 log2(e)
                    v[jt][r] = a;
                    mx = fmaxf(mx, a);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[jt][r] = __builtin_amdgcn_exp2f(v[jt][r] - mx);
                    sum += v[jt][r];
                }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            inv[it] = __builtin_amdgcn_rcpf(sum);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) pb[it][m][jj] = (half_t)v[2 * m + (jj >> 2)][jj & 3];
        }
        float4v oacc[2][4];   // [d tile][token tile]: rows 4 lq + r of d tile dt = dims 8 lq + 4 dt + r
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int it = 0; it < 4; ++it) oacc[dt][it] = (float4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                // element jj of the A fragment: key 32 m + 16 (jj >> 2) + 4 lq + (jj & 3), dim 8 p + 4 dt + e of column 4 p + e
                const half_t* base = myV + (32 * m + 4 * lq + qq) * HD + 8 * pp + 4 * (dt ^ swz);
                const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)base);
                const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(base + 16 * HD));
                half8 vt;
#pragma unroll
                for (int e = 0; e < 4; ++e) { vt[e] = (half_t)lo[e]; vt[4 + e] = (half_t)hi[e]; }
#pragma unroll
                for (int it = 0; it < 4; ++it) oacc[dt][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vt, pb[it][m], oacc[dt][it], 0, 0, 0);
            }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            half8 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) { o[r] = (half_t)(oacc[0][it][r] * inv[it]); o[4 + r] = (half_t)(oacc[1][it][r] * inv[it]); }
            *reinterpret_cast<half8*>(out + (long long)s_pos[it * 16 + lr] * out_pitch + head * HD + lq * 8) = o;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();   // the next head reuses this wave's LDS tiles
    }
}

}  // namespace

extern "C" int elvis_window_attention(const void* qkv, void* out, int dtype, int n, int h, int w, int heads,
                                      int head_dim, int ws, int shift, int qkv_pitch, int out_pitch,
                                      const float* bias_table, float scale, elvis_stream_t stream) {
    ELVIS_REQUIRE(qkv && out && bias_table, "elvis_window_attention: null pointer");
    ELVIS_REQUIRE(ws == WS && head_dim == HD, "elvis_window_attention: only window 8 / head_dim 32 are built (got %d/%d)", ws, head_dim);
    ELVIS_REQUIRE(n > 0 && h > 0 && w > 0 && heads > 0 && h % ws == 0 && w % ws == 0, "elvis_window_attention: H,W (%d,%d) must be multiples of the window %d", h, w, ws);
    ELVIS_REQUIRE(shift >= 0 && shift < ws, "elvis_window_attention: bad shift %d", shift);
    ELVIS_REQUIRE(qkv_pitch >= 3 * heads * head_dim && out_pitch >= heads * head_dim, "elvis_window_attention: bad pitch");
    long long blocks = (long long)n * (h / ws) * (w / ws) * heads;
    ELVIS_REQUIRE(blocks < 0x7fffffffLL, "elvis_window_attention: grid too large");
    static const bool attn_valu = getenv("ELVIS_ATTN_VALU") != nullptr;   // A/B switch, read once
    if (dtype == ELVIS_F16 && !attn_valu) {
        // MFMA path: one workgroup per window, waves loop over heads
        long long wblocks = (long long)n * (h / ws) * (w / ws);
        static const bool lds_form = getenv("ELVIS_ATTN_LDS") != nullptr;   // A/B switch: round 2's P / O through LDS
        if (lds_form)
            hipLaunchKernelGGL(window_attention_mfma_kernel, dim3((unsigned)wblocks), dim3(64 * ATT_NW), 0, (hipStream_t)stream,
                               (const half_t*)qkv, (half_t*)out, h, w, heads, shift, qkv_pitch, out_pitch, bias_table, scale);
        else
            hipLaunchKernelGGL(window_attention_tr_kernel, dim3((unsigned)wblocks), dim3(64 * ATT_NW), 0, (hipStream_t)stream,
                               (const half_t*)qkv, (half_t*)out, h, w, heads, shift, qkv_pitch, out_pitch, bias_table, scale);
    } else if (dtype == ELVIS_F16)
        hipLaunchKernelGGL(window_attention_kernel<half_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           (const half_t*)qkv, (half_t*)out, h, w, heads, shift, qkv_pitch, out_pitch, bias_table, scale);
    else if (dtype == ELVIS_F32)
        hipLaunchKernelGGL(window_attention_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           (const float*)qkv, (float*)out, h, w, heads, shift, qkv_pitch, out_pitch, bias_table, scale);
    else
        ELVIS_REQUIRE(false, "elvis_window_attention: bad dtype");
    ELVIS_CHECK_LAUNCH("elvis_window_attention");
    return ELVIS_OK;
}
