// Swin (shifted-)window attention on an NHWC token image.  One workgroup per (window, head):
// 64 tokens x head_dim 32.  QK^T + relative-position bias + shift mask -> softmax -> PV, all in
// fp32 on LDS-resident tiles; the cyclic shift / window partition / reverse are pure index math
// (tokens never leave image order in HBM).
#include "common.h"

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
    if (dtype == ELVIS_F16)
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
