// Fused per-token blocks of the Swin layers (SinSR Swin-UNet, ELVIS v2 Blur deblurrer), f16 storage, fp32 accumulate:
//
//   MLP    : out = x + fc2( GELU( fc1( LayerNorm(x) ) ) )          (one launch instead of LN, fc1, fc2: the 4C-wide hidden
//                                                                   tensor never leaves the CU - it was written and read
//                                                                   back through HBM, 10 C bytes per token of 17 C)
//   LINEAR : out = W . LayerNorm(x) + b                             (LN fused into the qkv projection)
//   PROJ   : y' = y + Wp . a + bp ;  out = y' + fc2( GELU( fc1( LayerNorm(y') ) ) )
//            (the attention output projection folded in front of the MLP: y' is born in the accumulator layout, LayerNorm
//             runs on the accumulators, and stacked accumulator tiles are fc1's B operand - the same k permutation, now
//             applied to fc1's packed columns; y' never reaches HBM: 3 C bytes per token instead of 6 C)
//
// A workgroup (8 waves) owns 128*PXT consecutive tokens; a wave owns 16*PXT of them and keeps their C channels in
// registers for the whole kernel as MFMA B fragments (token = column): LayerNorm is two register passes plus two
// cross-lane adds, the normalised tile never touches memory.  The first GEMM's weights stream through LDS in chunks of
// 64 output rows (LDS-DMA, two stages, issued one chunk ahead): S = W1[chunk] . xn is a [64 x 16*PXT] accumulator
// tile per wave.  MLP: bias + erf-GELU in registers, and the accumulator tile IS the next MFMA's B operand - lane
// (token, q) of a 16x16x32 C tile holds rows 4q..4q+3, so two stacked tiles give it eight k values; which eight is a
// permutation of k that is applied to fc2's packed columns at load time (a sum does not care about its order).  The
// second GEMM accumulates out[C x 16*PXT] over all hidden chunks in registers; the epilogue adds bias and the residual
// and stores.  Output rows are permuted at packing time (as in the conv kernels) so that a lane owns 16 contiguous
// channels of a token: 16-byte stores and residual loads.
// The MLP is paced by the GELU's VALU (16 instructions per hidden value) rather than by its MFMAs or by HBM; waves 4-7 run
// half a chunk behind waves 0-3 so that one SIMD partner's GELU overlaps the other's matrix part.
#include "common.h"
#include <mutex>
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct SwinArgs {
    const half_t* y;      // PROJ: the residual stream [M, y_pitch]
    const float* bp;      // PROJ: the projection's bias [C]
    int y_pitch;
    const half_t* x;      // [M, x_pitch]  (PROJ: the attention output)
    half_t* out;          // [M, out_pitch]
    const half_t* w;      // packed weights: per chunk [W1 chunk | W2 chunk]
    const float* b1;      // [n1]
    const float* b2;      // [C] (MLP)
    const float* gamma;   // [C]
    const float* beta;    // [C]
    long long M;
    int x_pitch, out_pitch;
    int n1;               // fc1 outputs = hidden (MLP) or the projection's outputs (LINEAR); multiple of 64
    float eps;
};

__device__ __forceinline__ int srow_off(int row, int q) { return row * 64 + ((q ^ (((row >> 2) & 1) << 1)) << 4); }
template <typename F> __device__ __forceinline__ F lds_frag16(const char* p) {
    return *reinterpret_cast<const F*>(__builtin_assume_aligned(p, 16));
}

template <int C, int PXT, int MODE, bool STAG>   // MODE 0: LINEAR, 1: MLP, 2: PROJ (+ MLP); STAG: waves 4-7 half a chunk behind
__global__ __launch_bounds__(512, 2) void swin_fused_kernel(SwinArgs p) {
    constexpr bool MLP = MODE >= 1, PROJ = MODE == 2;
    constexpr int NP = PROJ ? C / 64 : 0;               // 64-row chunks of the projection in front of the MLP chunks
    constexpr int KB = C / 32;                          // 32-channel k blocks of the first GEMM
    constexpr int CT = C / 16;                          // 16-row output tiles of the second GEMM
    constexpr int W1B = KB * 4096;                      // one chunk of W1: [KB][64 rows][64 B]
    constexpr int W2B = MLP ? 2 * C * 64 : 0;           // one chunk of W2: [2 k blocks][C rows][64 B]
    constexpr int STAGE = W1B + W2B;
    constexpr int PIECES = STAGE / 16 / 512;            // LDS-DMA pieces per thread per stage
    constexpr int PIECES1 = W1B / 16 / 512;             // ... per projection chunk (a W1-shaped block)
    static_assert(C % 64 == 0 && (STAGE / 16) % 512 == 0 && 2 * STAGE <= 160 * 1024, "shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const long long tok0 = ((long long)blockIdx.x * 8 + wave) * (16 * PXT);
    const int nchunks = p.n1 / 64;

    // ---- weight stream: stage s <- chunk hc, linear copy with the row swizzle on the per-lane SOURCE offset
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_ptr_t)smem);
    int src_off[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int c = tid + i * 512;
        src_off[i] = srow_off(c >> 2, c & 3);
    }
    // global chunk index g: NP projection chunks (W1-shaped blocks), then the MLP / LINEAR chunks (whole stages)
    auto w_issue = [&](int g, int s) {
        g = g < NP + nchunks ? g : NP + nchunks - 1;    // the chunk after the last re-loads it (never read): uniform vmcnt
        const bool proj = g < NP;
        const char* base = (const char*)p.w + (proj ? (long long)g * W1B : (long long)NP * W1B + (long long)(g - NP) * STAGE);
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            if (PROJ && proj && i >= PIECES1) break;    // (wave-uniform) a projection chunk is the W1-shaped head of a stage
            unsigned keep;
            const unsigned dst = lds_base + (unsigned)(s * STAGE + (i * 512 + wave_u * 64) * 16);
            const char* src = base + src_off[i];
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
    };
    w_issue(0, 0);

    // ---- the wave's tokens: load (B-fragment shaped); LINEAR / MLP: LayerNorm in registers, pack to f16
    half8 xb[PXT][KB];
#pragma unroll
    for (int t = 0; t < PXT; ++t) {
        const long long tok = tok0 + 16 * t + lr;
        const half_t* src = p.x + (tok < p.M ? tok : 0) * p.x_pitch + 8 * lq;
#pragma unroll
        for (int k = 0; k < KB; ++k) xb[t][k] = *reinterpret_cast<const half8*>(src + 32 * k);
    }
    if constexpr (!PROJ) {
        float mean[PXT], rstd[PXT];
#pragma unroll
        for (int t = 0; t < PXT; ++t) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < KB; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) s += (float)xb[t][k][e];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            mean[t] = s / (float)C;
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < KB; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = (float)xb[t][k][e] - mean[t];
                    q = fmaf(d, d, q);
                }
            q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            rstd[t] = 1.0f / sqrtf(q / (float)C + p.eps);
        }
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const float4v g0 = *reinterpret_cast<const float4v*>(p.gamma + 32 * k + 8 * lq), g1 = *reinterpret_cast<const float4v*>(p.gamma + 32 * k + 8 * lq + 4);
            const float4v h0 = *reinterpret_cast<const float4v*>(p.beta + 32 * k + 8 * lq), h1 = *reinterpret_cast<const float4v*>(p.beta + 32 * k + 8 * lq + 4);
#pragma unroll
            for (int t = 0; t < PXT; ++t)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float g = e < 4 ? g0[e & 3] : g1[e & 3], h = e < 4 ? h0[e & 3] : h1[e & 3];
                    xb[t][k][e] = (half_t)(((float)xb[t][k][e] - mean[t]) * rstd[t] * g + h);
                }
        }
    }

    float4v acc[MLP ? CT : 1][PXT];
    if constexpr (MLP && !PROJ) {
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int t = 0; t < PXT; ++t) acc[c][t] = (float4v){0.f, 0.f, 0.f, 0.f};
    }
    const int a_off = srow_off(lr, lq);

    if constexpr (PROJ) {
        // ---- y' = y + Wp . a + bp, chunk pc = output channels 64 pc .. 64 pc + 63 = accumulator tiles 4 pc .. 4 pc + 3
        //      (the lane's rows of tile 4 pc + i are channels 64 pc + 16 lq + 4 i + r: the layout the epilogue stores from)
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) {
            const int s = pc & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            w_issue(pc + 1, s ^ 1);
            const char* w1 = smem + s * STAGE + a_off;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4v b = *reinterpret_cast<const float4v*>(p.bp + 64 * pc + 16 * lq + 4 * i);
#pragma unroll
                for (int t = 0; t < PXT; ++t) acc[4 * pc + i][t] = b;
            }
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                half8 a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = lds_frag16<half8>(w1 + k * 4096 + i * 1024);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < PXT; ++t)
                        acc[4 * pc + i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], xb[t][k], acc[4 * pc + i][t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < PXT; ++t) {
                const long long tok = tok0 + 16 * t + lr;
                const half_t* yp = p.y + (tok < p.M ? tok : 0) * p.y_pitch + 64 * pc + 16 * lq;
                const half8 r0 = *reinterpret_cast<const half8*>(yp), r1 = *reinterpret_cast<const half8*>(yp + 8);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[4 * pc + i][t][r] += (float)(i < 2 ? r0[4 * i + r] : r1[4 * (i - 2) + r]);
            }
        }
        // ---- LayerNorm of y' on the accumulators (a token's C values: CT x 4 in this lane, the rest in the lanes 16 / 32
        //      / 48 away), written as fc1's B fragments: element j of k block m is tile 2 m + (j >> 2), row 4 lq + (j & 3)
        //      = channel 64 (m >> 1) + 16 lq + 8 (m & 1) + j - fc1's columns are packed in that order
#pragma unroll
        for (int t = 0; t < PXT; ++t) {
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) sum += acc[c][t][r];
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum / (float)C;
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = acc[c][t][r] - mean;
                    q = fmaf(d, d, q);
                }
            q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            const float rstd = 1.0f / sqrtf(q / (float)C + p.eps);
#pragma unroll
            for (int m = 0; m < KB; ++m) {
                const int ch = 64 * (m >> 1) + 16 * lq + 8 * (m & 1);
                const float4v g0 = *reinterpret_cast<const float4v*>(p.gamma + ch), g1 = *reinterpret_cast<const float4v*>(p.gamma + ch + 4);
                const float4v h0 = *reinterpret_cast<const float4v*>(p.beta + ch), h1 = *reinterpret_cast<const float4v*>(p.beta + ch + 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float g = j < 4 ? g0[j & 3] : g1[j & 3], h = j < 4 ? h0[j & 3] : h1[j & 3];
                    xb[t][m][j] = (half_t)((acc[2 * m + (j >> 2)][t][j & 3] - mean) * rstd * g + h);
                }
            }
        }
    }

    if constexpr (MLP && !STAG) {
        // ---- the MLP loop, all eight waves in lockstep: one barrier per chunk
        for (int hc = 0; hc < nchunks; ++hc) {
            const int s = (NP + hc) & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            w_issue(NP + hc + 1, s ^ 1);
            const char* w1 = smem + s * STAGE + a_off;
            float4v S[4][PXT];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4v b = *reinterpret_cast<const float4v*>(p.b1 + 64 * hc + 16 * i + 4 * lq);
#pragma unroll
                for (int t = 0; t < PXT; ++t) S[i][t] = b;
            }
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                half8 a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = lds_frag16<half8>(w1 + k * 4096 + i * 1024);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < PXT; ++t) S[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], xb[t][k], S[i][t], 0, 0, 0);
            }
            half8 H[PXT][2];
#pragma unroll
            for (int t = 0; t < PXT; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int j = 0; j < 8; ++j) H[t][m][j] = (half_t)gelu_erf_f(S[2 * m + (j >> 2)][t][j & 3]);
            const char* w2 = smem + s * STAGE + W1B + a_off;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const half8 a = lds_frag16<half8>(w2 + m * (C * 64) + c * 1024);
#pragma unroll
                    for (int t = 0; t < PXT; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, H[t][m], acc[c][t], 0, 0, 0);
                }
        }
    } else if constexpr (MLP) {
        // ---- the MLP loop, STAGGERED: a chunk is two half-steps separated by workgroup barriers,
        //        X(c) = fc1 chunk c (48 MFMAs) + GELU of its first 32 rows,   Y(c) = GELU of the other 32 rows + fc2 chunk c (48 MFMAs),
        //      and waves 4-7 run one half-step behind waves 0-3.  Waves w and w + 4 share a SIMD, so while one of them is in
        //      its matrix part the other is in its GELU part: the GELU's VALU (longer than the MFMAs) no longer serializes
        //      with them as it does when all eight waves run in lockstep (MI355X_MICROARCH.md, two waves per SIMD, item 9).
        //      Chunk c sits in stage (NP + c) & 1 during half-steps 2c .. 2c+2; its successor in that stage, chunk c + 2, is
        //      requested at half-step 2c + 3 and waited for (vmcnt(0) + barrier) at 2c + 4.
        const int grp = wave >> 2;
        float4v S[4][PXT];
        half8 H[PXT][2];
        for (int tau = 0; tau <= 2 * nchunks; ++tau) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tau & 1) {
                const int c = (tau + 1) >> 1;
                if (c < nchunks) w_issue(NP + c, (NP + c) & 1);
            }
            const int u = tau - grp;                       // this wave's own half-step
            if (u < 0 || u >= 2 * nchunks) continue;
            const int hc = u >> 1, s = (NP + hc) & 1;
            if ((u & 1) == 0) {
                const char* w1 = smem + s * STAGE + a_off;
#pragma unroll
                for (int i = 0; i < 4; ++i) {               // bias: the lane's rows of tile i are 16 i + 4 lq + r
                    const float4v b = *reinterpret_cast<const float4v*>(p.b1 + 64 * hc + 16 * i + 4 * lq);
#pragma unroll
                    for (int t = 0; t < PXT; ++t) S[i][t] = b;
                }
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                    half8 a[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[i] = lds_frag16<half8>(w1 + k * 4096 + i * 1024);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int t = 0; t < PXT; ++t) S[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], xb[t][k], S[i][t], 0, 0, 0);
                }
                // GELU; two stacked 16-row tiles are one 32-deep B fragment: element j of lane (token, q) is hidden row
                // 32 m + 16 (j >> 2) + 4 q + (j & 3) of the chunk - fc2's columns are packed in that order
#pragma unroll
                for (int t = 0; t < PXT; ++t)
#pragma unroll
                    for (int j = 0; j < 8; ++j) H[t][0][j] = (half_t)gelu_erf_f(S[j >> 2][t][j & 3]);
            } else {
#pragma unroll
                for (int t = 0; t < PXT; ++t)
#pragma unroll
                    for (int j = 0; j < 8; ++j) H[t][1][j] = (half_t)gelu_erf_f(S[2 + (j >> 2)][t][j & 3]);
                const char* w2 = smem + s * STAGE + W1B + a_off;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        const half8 a = lds_frag16<half8>(w2 + m * (C * 64) + c * 1024);
#pragma unroll
                        for (int t = 0; t < PXT; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, H[t][m], acc[c][t], 0, 0, 0);
                    }
            }
        }
    } else {
        for (int hc = 0; hc < nchunks; ++hc) {
            const int s = hc & 1;
            // chunk hc has landed (this thread's pieces: the only DMAs in flight), and everyone is past chunk hc-1's reads
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            w_issue(hc + 1, s ^ 1);
            const char* w1 = smem + s * STAGE + a_off;
            float4v S[4][PXT];
#pragma unroll
            for (int i = 0; i < 4; ++i) {                   // bias: the lane's channels of tile i are 16 lq + 4 i + r
                const float4v b = *reinterpret_cast<const float4v*>(p.b1 + 64 * hc + 16 * lq + 4 * i);
#pragma unroll
                for (int t = 0; t < PXT; ++t) S[i][t] = b;
            }
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                half8 a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = lds_frag16<half8>(w1 + k * 4096 + i * 1024);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < PXT; ++t) S[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], xb[t][k], S[i][t], 0, 0, 0);
            }
            // the chunk's 64 output channels of each token: the lane's 16 contiguous ones as two 16-byte stores
#pragma unroll
            for (int t = 0; t < PXT; ++t) {
                const long long tok = tok0 + 16 * t + lr;
                if (tok < p.M) {
                    half8 lo, hi;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        lo[r] = (half_t)S[0][t][r]; lo[4 + r] = (half_t)S[1][t][r];
                        hi[r] = (half_t)S[2][t][r]; hi[4 + r] = (half_t)S[3][t][r];
                    }
                    half_t* dst = p.out + tok * p.out_pitch + 64 * hc + 16 * lq;
                    *reinterpret_cast<half8*>(dst) = lo;
                    *reinterpret_cast<half8*>(dst + 8) = hi;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail DMA must have landed before the workgroup's LDS is released
    if constexpr (MLP) {
        // out = x + b2 + acc: output tile c = 4 g + i, accumulator row 4 lq + r  <->  channel 64 g + 16 lq + 4 i + r
#pragma unroll
        for (int t = 0; t < PXT; ++t) {
            const long long tok = tok0 + 16 * t + lr;
            const long long tc = tok < p.M ? tok : 0;
#pragma unroll
            for (int g = 0; g < C / 64; ++g) {
                const int ch = 64 * g + 16 * lq;
                half8 r0 = {}, r1 = {};   // (PROJ: the accumulators started from y + proj: nothing to add)
                if constexpr (!PROJ) {
                    r0 = *reinterpret_cast<const half8*>(p.x + tc * p.x_pitch + ch);
                    r1 = *reinterpret_cast<const half8*>(p.x + tc * p.x_pitch + ch + 8);
                }
                half8 lo, hi;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4v b = *reinterpret_cast<const float4v*>(p.b2 + ch + 4 * i);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float res = PROJ ? 0.f : (float)(i < 2 ? r0[4 * i + r] : r1[4 * (i - 2) + r]);
                        const half_t v = (half_t)(acc[4 * g + i][t][r] + b[r] + res);
                        if (i < 2) lo[4 * i + r] = v; else hi[4 * (i - 2) + r] = v;
                    }
                }
                if (tok < p.M) {
                    half_t* dst = p.out + tok * p.out_pitch + ch;
                    *reinterpret_cast<half8*>(dst) = lo;
                    *reinterpret_cast<half8*>(dst + 8) = hi;
                }
            }
        }
    }
}

template <int C, int PXT, int MODE, bool STAG> int launch_swin(const SwinArgs& a, hipStream_t stream) {
    constexpr int STAGE = (C / 32) * 4096 + (MODE >= 1 ? 2 * C * 64 : 0);
    const size_t lds = 2 * (size_t)STAGE;
    {
        static std::mutex mu;
        static bool attr_set[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        std::lock_guard<std::mutex> guard(mu);
        if (!attr_set[dev]) {
            hipError_t e = hipFuncSetAttribute((const void*)swin_fused_kernel<C, PXT, MODE, STAG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) {
                elvis_set_error("elvis_swin: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
                return ELVIS_E_RUNTIME;
            }
            attr_set[dev] = true;
        }
    }
    const long long per = 128LL * PXT;
    const long long blocks = (a.M + per - 1) / per;
    ELVIS_REQUIRE(blocks < 0x7fffffffLL, "elvis_swin: grid too large");
    hipLaunchKernelGGL((swin_fused_kernel<C, PXT, MODE, STAG>), dim3((unsigned)blocks), dim3(512), lds, stream, a);
    ELVIS_CHECK_LAUNCH("elvis_swin");
    return ELVIS_OK;
}

// Stagger (waves 4-7 half a chunk behind) against all eight waves in lockstep, measured per shape in one process
// (tools/swin_bench.py, us per launch, lockstep -> staggered): C = 192 MLP 1137 -> 1064 but projection + MLP 1315 -> 1477
// (the combined kernel spills once the half-steps split its loop); C = 64 487 -> 548 and 559 -> 740; C = 128 369 -> 392 and
// 434 -> 438; C = 256 314 -> 276 and 339 -> 328.  So: staggered for C = 256 and for the C = 192 MLP, lockstep otherwise.
// ELVIS_SWIN_STAGGER=0/1 forces it off / on for A/B runs (read once).
static int stagger_mode() {
    static const int v = getenv("ELVIS_SWIN_STAGGER") ? atoi(getenv("ELVIS_SWIN_STAGGER")) : -1;
    return v;
}
template <int C, int PXT, int MODE> int launch_swin_s(const SwinArgs& a, hipStream_t stream, bool default_stag) {
    if constexpr (MODE == 0) return launch_swin<C, PXT, MODE, false>(a, stream);
    const int m = stagger_mode();
    const bool st = m < 0 ? default_stag : m != 0;
    return st ? launch_swin<C, PXT, MODE, true>(a, stream) : launch_swin<C, PXT, MODE, false>(a, stream);
}
template <int MODE> int dispatch_swin(int c, const SwinArgs& a, hipStream_t stream) {
    switch (c) {
        case 64: return launch_swin_s<64, 2, MODE>(a, stream, false);
        case 128: return launch_swin_s<128, 2, MODE>(a, stream, false);
        case 192: return launch_swin_s<192, 2, MODE>(a, stream, MODE == 1);
        case 256: return launch_swin_s<256, 1, MODE>(a, stream, true);
        default: break;
    }
    elvis_set_error("elvis_swin: channels must be 64, 128, 192 or 256 (got %d)", c);
    return ELVIS_E_INVALID;
}

static int check_common(const void* x, const void* out, const void* w, const float* gamma, const float* beta, long long tokens, int c,
                        int x_pitch, int out_pitch, int n_out) {
    ELVIS_REQUIRE(x && out && w && gamma && beta, "elvis_swin: null pointer");
    ELVIS_REQUIRE(tokens > 0 && c > 0 && c % 64 == 0 && x_pitch >= c && x_pitch % 8 == 0 && out_pitch >= n_out && out_pitch % 8 == 0,
                  "elvis_swin: bad shape (tokens %lld, c %d, pitches %d / %d)", tokens, c, x_pitch, out_pitch);
    ELVIS_REQUIRE((((uintptr_t)x | (uintptr_t)out | (uintptr_t)w | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "elvis_swin: pointers must be 16-byte aligned");
    return ELVIS_OK;
}

}  // namespace

extern "C" size_t elvis_swin_packed_bytes(int c, int n1, int mode) {
    if (c <= 0 || c % 64 || n1 <= 0 || n1 % 64 || mode < 0 || mode > 2) return 0;
    const size_t w1b = (size_t)(c / 32) * 4096;
    return (mode == 2 ? (size_t)(c / 64) * w1b : 0) + (size_t)(n1 / 64) * (w1b + (mode ? 2 * (size_t)c * 64 : 0));
}

// OIHW-style fp32 weights -> the kernels' packed f16 stream: per 64-row chunk hc of the first matrix
//   [k block][64 rows][32 halfs] of W1 (rows in natural order for the MLP, in the 16-contiguous-channels-per-lane order
//   for LINEAR), then for the MLP [2 k blocks][C rows][32 halfs] of W2's columns 64 hc .. 64 hc + 63 (rows in the
//   16-contiguous order, columns in the accumulator-as-operand order, swin_fused_kernel).
__global__ void swin_pack_kernel(const float* __restrict__ wp, const float* __restrict__ w1, const float* __restrict__ w2,
                                 half_t* __restrict__ out, int c, int n1, int mode, long long total) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long i0 = i;
    const int kb_n = c / 32;
    const long long w1h = (long long)kb_n * 2048;                                 // halfs of one W1-shaped block
    if (mode == 2) {                                                              // the projection's chunks come first
        const long long ph = (long long)(c / 64) * w1h;
        if (i < ph) {
            const int pc = (int)(i / w1h);
            const long long r = i - (long long)pc * w1h;
            const int kk = (int)(r & 31), row = (int)((r >> 5) & 63), kb = (int)(r >> 11);
            const int i4 = row >> 4, q = (row >> 2) & 3, rr = row & 3;
            out[i0] = (half_t)wp[(long long)(64 * pc + 16 * q + 4 * i4 + rr) * c + 32 * kb + kk];
            return;
        }
        i -= ph;
    }
    const long long stage = w1h + (mode ? 2LL * c * 32 : 0);                      // halfs per chunk
    const int hc = (int)(i / stage);
    long long r = i - (long long)hc * stage;
    float v;
    if (r < w1h) {
        const int kk = (int)(r & 31), row = (int)((r >> 5) & 63), kb = (int)(r >> 11);
        int o = 64 * hc + row;                                        // MLP: hidden row, natural order
        if (mode == 0) {
            const int i4 = row >> 4, q = (row >> 2) & 3, rr = row & 3;
            o = 64 * hc + 16 * q + 4 * i4 + rr;
        }
        int col = 32 * kb + kk;
        if (mode == 2)                                                // fc1 reads LayerNorm(y') out of the accumulator tiles
            col = 64 * (kb >> 1) + 16 * (kk >> 3) + 8 * (kb & 1) + (kk & 7);
        v = w1[(long long)o * c + col];
    } else {
        r -= w1h;
        const int kk = (int)(r & 31);
        const int row = (int)((r >> 5) % c), m = (int)((r >> 5) / c);
        const int g = row >> 6, i4 = (row >> 4) & 3, q = (row >> 2) & 3, rr = row & 3;
        const int co = 64 * g + 16 * q + 4 * i4 + rr;
        const int lqk = kk >> 3, j = kk & 7;
        const int hid = 64 * hc + 32 * m + 16 * (j >> 2) + 4 * lqk + (j & 3);
        v = w2[(long long)co * n1 + hid];
    }
    out[i0] = (half_t)v;
}

static int pack_common(const float* wp, const float* w1, const float* w2, void* packed, int c, int n1, int mode, hipStream_t stream) {
    const size_t bytes = elvis_swin_packed_bytes(c, n1, mode);
    ELVIS_REQUIRE(bytes > 0, "elvis_swin_pack_weights: c (%d) and n1 (%d) must be positive multiples of 64", c, n1);
    const long long total = (long long)(bytes / 2);
    hipLaunchKernelGGL(swin_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, wp, w1, w2, (half_t*)packed, c, n1,
                       mode, total);
    ELVIS_CHECK_LAUNCH("elvis_swin_pack_weights");
    return ELVIS_OK;
}

extern "C" int elvis_swin_pack_weights(const float* w1, const float* w2, void* packed, int c, int n1, int mlp, elvis_stream_t stream) {
    ELVIS_REQUIRE(w1 && packed && (!mlp || w2), "elvis_swin_pack_weights: null pointer");
    return pack_common(nullptr, w1, w2, packed, c, n1, mlp ? 1 : 0, (hipStream_t)stream);
}

extern "C" int elvis_swin_pack_proj_mlp(const float* wp, const float* w1, const float* w2, void* packed, int c, int hidden,
                                        elvis_stream_t stream) {
    ELVIS_REQUIRE(wp && w1 && w2 && packed, "elvis_swin_pack_proj_mlp: null pointer");
    return pack_common(wp, w1, w2, packed, c, hidden, 2, (hipStream_t)stream);
}

extern "C" int elvis_swin_mlp(const void* x, void* out, const void* packed, const float* b1, const float* b2, const float* gamma,
                              const float* beta, long long tokens, int c, int hidden, int x_pitch, int out_pitch, float eps,
                              elvis_stream_t stream) {
    int rc = check_common(x, out, packed, gamma, beta, tokens, c, x_pitch, out_pitch, c);
    if (rc) return rc;
    ELVIS_REQUIRE(b1 && b2 && hidden > 0 && hidden % 64 == 0, "elvis_swin_mlp: hidden (%d) must be a positive multiple of 64", hidden);
    ELVIS_REQUIRE((((uintptr_t)b1 | (uintptr_t)b2) & 15) == 0, "elvis_swin_mlp: biases must be 16-byte aligned");
    SwinArgs a{nullptr, nullptr, 0, (const half_t*)x, (half_t*)out, (const half_t*)packed, b1, b2, gamma, beta, tokens, x_pitch, out_pitch, hidden, eps};
    return dispatch_swin<1>(c, a, (hipStream_t)stream);
}

extern "C" int elvis_swin_ln_linear(const void* x, void* out, const void* packed, const float* bias, const float* gamma, const float* beta,
                                    long long tokens, int c, int n_out, int x_pitch, int out_pitch, float eps, elvis_stream_t stream) {
    int rc = check_common(x, out, packed, gamma, beta, tokens, c, x_pitch, out_pitch, n_out);
    if (rc) return rc;
    ELVIS_REQUIRE(bias && n_out > 0 && n_out % 64 == 0 && (((uintptr_t)bias) & 15) == 0, "elvis_swin_ln_linear: n_out (%d) must be a positive multiple of 64, bias 16-byte aligned", n_out);
    SwinArgs a{nullptr, nullptr, 0, (const half_t*)x, (half_t*)out, (const half_t*)packed, bias, nullptr, gamma, beta, tokens, x_pitch, out_pitch, n_out, eps};
    return dispatch_swin<0>(c, a, (hipStream_t)stream);
}

extern "C" int elvis_swin_proj_mlp(const void* attn, const void* y, void* out, const void* packed, const float* bp, const float* b1,
                                   const float* b2, const float* gamma, const float* beta, long long tokens, int c, int hidden,
                                   int attn_pitch, int y_pitch, int out_pitch, float eps, elvis_stream_t stream) {
    int rc = check_common(attn, out, packed, gamma, beta, tokens, c, attn_pitch, out_pitch, c);
    if (rc) return rc;
    ELVIS_REQUIRE(y && bp && b1 && b2 && hidden > 0 && hidden % 64 == 0 && y_pitch >= c && y_pitch % 8 == 0,
                  "elvis_swin_proj_mlp: hidden (%d) must be a positive multiple of 64, y_pitch (%d) >= c and a multiple of 8", hidden, y_pitch);
    ELVIS_REQUIRE((((uintptr_t)y | (uintptr_t)bp | (uintptr_t)b1 | (uintptr_t)b2) & 15) == 0, "elvis_swin_proj_mlp: pointers must be 16-byte aligned");
    SwinArgs a{(const half_t*)y, bp, y_pitch, (const half_t*)attn, (half_t*)out, (const half_t*)packed, b1, b2, gamma, beta, tokens, attn_pitch,
               out_pitch, hidden, eps};
    return dispatch_swin<2>(c, a, (hipStream_t)stream);
}
