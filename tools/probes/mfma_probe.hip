// Probe (not product code): rate of the f16 / f32 MFMA variants on gfx950 and whether f16 subnormal inputs survive.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_probe.hip -o gpurun_out/mfma_probe && gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <int MODE> __global__ __launch_bounds__(256) void rate(float* out, int iters) {
    float4v acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (float4v){0, 0, 0, 0};
    half4 a4 = {(_Float16)1.0f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)2.0f}, b4 = a4;
    half8 a8 = {1, 1, 1, 1, 1, 1, 1, 1}, b8 = a8;
    float af = 1.0f + threadIdx.x * 1e-3f, bf = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
            if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0);
            if (MODE == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void subnormal(float* out) {
    // A[row][k]: every entry 2^-20 (an f16 subnormal); B: 2^10.  16 terms -> 16 * 2^-10 if subnormals are kept, 0 if flushed
    half4 a = {(_Float16)9.5367431640625e-07f, (_Float16)9.5367431640625e-07f, (_Float16)9.5367431640625e-07f, (_Float16)9.5367431640625e-07f};
    half4 b = {(_Float16)1024.f, (_Float16)1024.f, (_Float16)1024.f, (_Float16)1024.f};
    float4v acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, acc, 0, 0, 0);
    half8 a8 = {a[0], a[0], a[0], a[0], a[0], a[0], a[0], a[0]}, b8 = {b[0], b[0], b[0], b[0], b[0], b[0], b[0], b[0]};
    float4v acc2 = {0, 0, 0, 0};
    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc2, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = acc2[0]; }
}

int main() {
    float* d; hipMalloc(&d, 1024 * 256 * 4 + 64);
    const int iters = 20000, grid = 1024;
    const double flop[3] = {16. * 16 * 32 * 2, 16. * 16 * 16 * 2, 16. * 16 * 4 * 2};
    const char* nm[3] = {"16x16x32_f16", "16x16x16_f16", "16x16x4_f32"};
    for (int m = 0; m < 3; ++m) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (m == 0) rate<0><<<grid, 256>>>(d, iters);
            if (m == 1) rate<1><<<grid, 256>>>(d, iters);
            if (m == 2) rate<2><<<grid, 256>>>(d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double tf = flop[m] * 8 * iters * 4.0 * grid / (ms * 1e-3) / 1e12;   // 4 waves per block
        printf("%-14s %8.3f ms  %8.1f TFLOP/s\n", nm[m], ms, tf);
    }
    subnormal<<<1, 64>>>(d);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("subnormal probe: 16x16x16 -> %g (kept: %g), 16x16x32 -> %g (kept: %g)\n", h[0], 16 * 9.765625e-4, h[1], 32 * 9.765625e-4);
    return 0;
}
