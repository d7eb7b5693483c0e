// The fp32 (exact MFMA) and compensated-f16 (ELVIS_F32X3) instantiations of the conv kernels: a translation unit of
// its own so that it compiles in parallel with conv.hip (which holds the API and the f16 instantiations).
#include "conv_kernels.inc"

__attribute__((visibility("hidden"))) int elvis_conv_launch_f32_(const void* conv_args, int halo, int tco, int id, hipStream_t stream) {
    const ConvArgs& a = *static_cast<const ConvArgs*>(conv_args);   // same definition in both translation units
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
