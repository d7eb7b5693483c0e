// The fp32 (exact MFMA) and compensated-f16 (ELVIS_F32X3) instantiations of the conv kernels: a translation unit of
// its own so that it compiles in parallel with conv.hip (which holds the API and the f16 instantiations).
#include "conv_kernels.inc"
#include "conv_x3p.inc"

__attribute__((visibility("hidden"))) int elvis_conv_launch_f32_(const void* conv_args, int halo, int tco, int id, hipStream_t stream) {
    const ConvArgs& a = *static_cast<const ConvArgs*>(conv_args);   // same definition in both translation units
    if (halo && a.x3 == 2) return tco == 128 ? launch_x3p<128>(a, stream) : launch_x3p<64>(a, stream);
    if (halo) {
        switch (tco) {
            case 128: return launch_halo<float, 128>(a, stream);
            case 64: return launch_halo<float, 64>(a, stream);
            case 32: return launch_halo<float, 32>(a, stream);
            default: return launch_halo<float, 16>(a, stream);
        }
    }
    return dispatch<float>(a, id, stream);
}

__attribute__((visibility("hidden"))) int elvis_conv_pack_x3p_(const float* w_oihw, void* packed, int cout, int ctot, int nkc, int n_co_tiles,
                                                              int tco, int taps, hipStream_t stream) {
    const long long total = (long long)taps * nkc * n_co_tiles * 2 * tco * 32;
    hipLaunchKernelGGL(pack_weights_x3p_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w_oihw, (half_t*)packed,
                       cout, ctot, nkc, n_co_tiles, tco, taps, total);
    ELVIS_CHECK_LAUNCH("elvis_conv_pack_weights(x3p)");
    return ELVIS_OK;
}
