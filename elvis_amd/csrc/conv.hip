// C ABI of the conv kernels (elvis_conv2d and friends) + the f16 instantiations; the fp32 / compensated-f16
// instantiations live in conv_f32.hip.  Kernels, launch helpers and dispatch rules: conv_kernels.inc.
#include "conv_kernels.inc"
#include "conv_ws.inc"
#include <string.h>

// conv_f32.hip: halo != 0 -> launch_halo<float, tco>, else dispatch<float>(id)
__attribute__((visibility("hidden"))) int elvis_conv_launch_f32_(const void* conv_args, int halo, int tco, int id, hipStream_t stream);
// conv_f32.hip: the planar compensated form's weight packing (conv_x3p.inc)
__attribute__((visibility("hidden"))) int elvis_conv_pack_x3p_(const float* w_oihw, void* packed, int cout, int ctot, int nkc, int n_co_tiles,
                                                              int tco, int taps, hipStream_t stream);

// A/B switches of the experiment tools: the environment is read ONCE per process, never on the per-call path;
// elvis_conv_debug_set() flips the same switches at run time (tests compare the halo kernels with the generic one).
#include <atomic>
static std::atomic<int> g_no_halo{-1};
static bool no_halo() {
    static const bool env = getenv("ELVIS_NO_HALO") != nullptr;
    const int v = g_no_halo.load(std::memory_order_relaxed);
    return v < 0 ? env : v != 0;
}
extern "C" int elvis_conv_debug_set(const char* key, int value) {
    ELVIS_REQUIRE(key, "elvis_conv_debug_set: null key");
    if (!strcmp(key, "no_halo")) { g_no_halo.store(value, std::memory_order_relaxed); return ELVIS_OK; }   // -1: back to the environment
    ELVIS_REQUIRE(false, "elvis_conv_debug_set: unknown key '%s'", key);
}
static int strip_width() {
    static const int v = getenv("ELVIS_STRIP") ? atoi(getenv("ELVIS_STRIP")) : 8;   // 0 = row-major tile walk
    return v;
}

static void conv_geom(const elvis_conv_desc* d, int* nkc1, int* nkc, int* co_pad) {
    int kc = kc_of(d);
    *nkc1 = (d->cin + kc - 1) / kc;
    *nkc = *nkc1 + (d->cin2 > 0 ? (d->cin2 + kc - 1) / kc : 0);
    TileCfg t = choose_tile(d->cout);
    *co_pad = ((d->cout + t.tco - 1) / t.tco) * t.tco;
}

extern "C" size_t elvis_conv_packed_weight_bytes(const elvis_conv_desc* d) {
    if (!d || d->cin <= 0 || d->cout <= 0) return 0;
    int nkc1, nkc, co_pad;
    conv_geom(d, &nkc1, &nkc, &co_pad);
    return (size_t)d->ksize * d->ksize * nkc * co_pad * (x3_planar_fmt(d) ? 128 : 64);   // planar: a hi and a lo row per 32 channels
}

extern "C" int elvis_conv_pack_weights(const elvis_conv_desc* d, const float* w_oihw, void* packed,
                                       elvis_stream_t stream) {
    int rc = validate(d);
    if (rc) return rc;
    ELVIS_REQUIRE(w_oihw && packed, "elvis_conv_pack_weights: null pointer");
    int nkc1, nkc, co_pad;
    conv_geom(d, &nkc1, &nkc, &co_pad);
    if (x3_planar_fmt(d)) {
        const int tco = choose_tile(d->cout).tco;
        return elvis_conv_pack_x3p_(w_oihw, packed, d->cout, d->cin + d->cin2, nkc, co_pad / tco, tco, d->ksize * d->ksize, (hipStream_t)stream);
    }
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
    if (!(d->dtype == ELVIS_F32X3 && halo_eligible(d) && !no_halo() && choose_tile(d->cout).tco >= 64)) return false;
    // a layer packed in the planar format runs on the planar kernel only (plain 3x3 / stride 1 / pad 1 geometry)
    return !x3_planar_fmt(d) || x3_planar_run(d);
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
    if (ws_shape_ok(d) && !no_halo()) {   // (a call with a residual or a statistics buffer falls back to the halo kernel below)
        const int nkc = (d->cin + 31) / 32 + (d->cin2 > 0 ? (d->cin2 + 31) / 32 : 0);
        snprintf(buf, n, "conv3x3_ws_kernel<%d,%d,%s>", nkc, c.tco, ws_stagger(nkc, c.tco) ? "true" : "false");
    } else if (halo_eligible(d) && !no_halo()) {
        const bool pro = d->ksize == 3 && d->prologue;
        if (x3_planar_run(d))
            snprintf(buf, n, "conv3x3_x3p_kernel<%d,%d,%d,%s,%s>", c.tco, halo_ty(d), d->ksize, pro ? "true" : "false", d->act ? "true" : "false");
        else if (d->dtype == ELVIS_F32X3 && c.tco >= 64)
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
    a.pad2y = subpix ? 1 - a.par_a : (s2d ? d->pad_before : 0);   // space-to-depth: pad (0,1,0,1) form 0, pad 1 form 1
    a.pad2x = subpix ? 1 - a.par_b : (s2d ? d->pad_before : 0);
    a.s2d = s2d ? 1 : 0;
    a.istr = s2d ? 2 : 1;
    a.nkc_c = s2d ? d->cin / 4 / kc_of(d) : 0x3fffffff;
    a.wfull = d->w;
    if (s2d) {   // the kernel sees the phase image: ho x wo pixels of 4C channels
        a.h = d->ho;
        a.w_in = d->wo;
    }
    a.tiles_x = ((subpix ? d->w : d->wo) + HALO_TX - 1) / HALO_TX;
    {
        a.strip = strip_width();
        if (a.strip < 0 || a.strip >= a.tiles_x) a.strip = 0;
        a.strip_full = a.strip > 0 ? a.tiles_x / a.strip : 0;
    }
    const int tyv = halo_ty(d);
    a.x3 = d->dtype == ELVIS_F32X3 ? (x3_planar_run(d) ? 2 : 1) : 0;
    a.two = (halo_two(d) || halo_g1(d)) ? 1 : 0;
    a.tall = (halo_two(d) && halo_tall(d)) ? 1 : 0;
    a.tiles_y = ((subpix ? d->h : d->ho) + tyv - 1) / tyv;
    if (ws_shape_ok(d) && !residual && !stats && !no_halo()) {   // narrow layer, large image: persistent weight-stationary kernel
        ELVIS_REQUIRE((long long)d->n * d->h * d->w < 0x7fffffffLL, "conv: input too large for 32-bit pixel indices");
        ConvArgs b = a;
        b.tiles_x = (d->wo + 31) / 32;
        b.tiles_y = (d->ho + 7) / 8;
        return dispatch_ws(b, t.tco, (hipStream_t)stream);
    }
    if (halo_eligible(d) && !no_halo()) {
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
        return elvis_conv_launch_f32_(&a, 1, t.tco, t.id, st);
    }
    ELVIS_REQUIRE(!stats, "elvis_conv2d: fused statistics need a 3x3/stride-1 conv with cout >= 64 (query elvis_conv_stats_tiles)");
    if (d->dtype == ELVIS_F16) return dispatch<half_t>(a, t.id, (hipStream_t)stream);
    return elvis_conv_launch_f32_(&a, 0, t.tco, t.id, (hipStream_t)stream);
}
