// Block-map glue kernels (uint8 NHWC frames): recompose / area downscale / blend /
// per-level select / feathered tile accumulate / normalise / SSE.
// All HBM-bound byte work: 16-byte vector accesses, no LDS needed (no reuse), one pass.
#include "common.h"

// ---------------------------------------------------------------------------------------
// Position decoding for a flat byte index over (n,h,w,c) frames.
struct BytePos {
    int f, y, x, ch;
};
__device__ __forceinline__ BytePos decode_pos(long long g, int h, int w, int c) {
    BytePos p;
    long long px = g / c;
    p.ch = (int)(g - px * c);
    long long row = px / w;
    p.x = (int)(px - row * w);
    p.f = (int)(row / h);
    p.y = (int)(row - (long long)p.f * h);
    return p;
}
__device__ __forceinline__ void advance_pos(BytePos& p, int h, int w, int c) {
    if (++p.ch == c) {
        p.ch = 0;
        if (++p.x == w) {
            p.x = 0;
            if (++p.y == h) {
                p.y = 0;
                ++p.f;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// recompose: out = (map <= thr) ? a : b.  Each lane owns 16 contiguous bytes; it evaluates the
// predicate per byte (a 16-byte vector spans <= 2 blocks for block*c >= 16) and loads only the
// source(s) it actually needs, so HBM traffic is ~2/3 of the algorithmic 3 streams.
template <bool POW2>   // block_size a power of two: the block index is a shift
__global__ __launch_bounds__(256) void recompose_u8_kernel(const uint8_t* __restrict__ a,
                                                           const uint8_t* __restrict__ b,
                                                           const int32_t* __restrict__ map,
                                                           uint8_t* __restrict__ out, int n, int h, int w,
                                                           int c, int block, int bshift, int by, int bx, int thr,
                                                           long long total) {
    long long vec = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long nvec = (total + 15) >> 4;
    for (; vec < nvec; vec += (long long)gridDim.x * blockDim.x) {
        long long g0 = vec << 4;
        BytePos p = decode_pos(g0, h, w, c);
        int nb = (int)((total - g0) < 16 ? (total - g0) : 16);
        uint32_t mask = 0;  // bit k set -> take a
        int last_key = -1;
        bool last_pred = false, pred = false;
        for (int k = 0; k < nb; ++k) {
            if (k == 0 || p.ch == 0) {   // the predicate changes at pixel boundaries only (round 1 evaluated two
                                         // integer divides per BYTE: 0.24 of the HBM peak; now per pixel, shifts)
                const int byi = POW2 ? (p.y >> bshift) : p.y / block, bxi = POW2 ? (p.x >> bshift) : p.x / block;
                pred = false;
                if (byi < by && bxi < bx) {
                    int key = (p.f * by + byi) * bx + bxi;
                    if (key != last_key) {
                        last_key = key;
                        last_pred = map[key] <= thr;
                    }
                    pred = last_pred;
                }
            }
            mask |= (pred ? 1u : 0u) << k;
            advance_pos(p, h, w, c);
        }
        uint32_t full = nb == 16 ? 0xFFFFu : ((1u << nb) - 1u);
        if (nb == 16) {
            uint4 va = make_uint4(0, 0, 0, 0), vb = make_uint4(0, 0, 0, 0);
            if (mask != 0) va = *reinterpret_cast<const uint4*>(a + g0);
            if (mask != full) vb = *reinterpret_cast<const uint4*>(b + g0);
            uint4 vo;
            if (mask == full) vo = va;
            else if (mask == 0) vo = vb;
            else {
                uint32_t wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w}, wo[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t m4 = (mask >> (4 * q)) & 0xF;
                    uint32_t sel = ((m4 & 1) ? 0xFFu : 0) | ((m4 & 2) ? 0xFF00u : 0) | ((m4 & 4) ? 0xFF0000u : 0) |
                                   ((m4 & 8) ? 0xFF000000u : 0);
                    wo[q] = (wa[q] & sel) | (wb[q] & ~sel);
                }
                vo = make_uint4(wo[0], wo[1], wo[2], wo[3]);
            }
            *reinterpret_cast<uint4*>(out + g0) = vo;
        } else {
            for (int k = 0; k < nb; ++k) out[g0 + k] = ((mask >> k) & 1) ? a[g0 + k] : b[g0 + k];
        }
    }
}

// Fast path: rows are whole 16-byte vectors (w*c % 16 == 0) and a block row segment is at least 16 bytes
// (block*c >= 16), so a vector lies in one image row and spans at most two blocks: two map look-ups and a
// bit mask per vector instead of a predicate per byte.
__global__ __launch_bounds__(256) void recompose_rows_u8_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                                const int32_t* __restrict__ map, uint8_t* __restrict__ out,
                                                                int h, int vec_per_row, int seg_bytes, int block, int by,
                                                                int bx, int thr, long long nvec) {
    for (long long vec = (long long)blockIdx.x * blockDim.x + threadIdx.x; vec < nvec; vec += (long long)gridDim.x * blockDim.x) {
        const long long row = vec / vec_per_row;                 // = f*h + y
        const int o = (int)(vec - row * vec_per_row) << 4;       // byte offset inside the row
        const int f = (int)(row / h), y = (int)(row - (long long)f * h);
        const int byi = y / block;
        const int bx0 = o / seg_bytes;
        const int bnd = (bx0 + 1) * seg_bytes - o;               // bytes of this vector that belong to block bx0
        uint32_t mask = 0;                                       // bit k set -> byte k from a
        if (byi < by) {
            const int32_t* mrow = map + ((long long)f * by + byi) * bx;
            const bool p0 = bx0 < bx && mrow[bx0] <= thr;
            const bool p1 = bnd < 16 && bx0 + 1 < bx && mrow[bx0 + 1] <= thr;
            const uint32_t lo = bnd >= 16 ? 0xFFFFu : ((1u << bnd) - 1u);
            mask = (p0 ? lo : 0u) | (p1 ? (0xFFFFu & ~lo) : 0u);
        }
        const long long g0 = vec << 4;
        uint4 va = make_uint4(0, 0, 0, 0), vb = make_uint4(0, 0, 0, 0);
        if (mask != 0) va = *reinterpret_cast<const uint4*>(a + g0);
        if (mask != 0xFFFFu) vb = *reinterpret_cast<const uint4*>(b + g0);
        uint4 vo;
        if (mask == 0xFFFFu) vo = va;
        else if (mask == 0) vo = vb;
        else {
            uint32_t wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w}, wo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t m4 = (mask >> (4 * q)) & 0xF;
                const uint32_t sel = ((m4 & 1) ? 0xFFu : 0) | ((m4 & 2) ? 0xFF00u : 0) | ((m4 & 4) ? 0xFF0000u : 0) |
                                     ((m4 & 8) ? 0xFF000000u : 0);
                wo[q] = (wa[q] & sel) | (wb[q] & ~sel);
            }
            vo = make_uint4(wo[0], wo[1], wo[2], wo[3]);
        }
        *reinterpret_cast<uint4*>(out + g0) = vo;
    }
}

__global__ void clamp_map_kernel(const int32_t* __restrict__ map, int32_t* __restrict__ map_out, int count,
                                 int thr, int clamp_to) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        int v = map[i];
        map_out[i] = (v <= thr) ? v : clamp_to;
    }
}

extern "C" int elvis_recompose_u8(const uint8_t* a, const uint8_t* b, const int32_t* map, uint8_t* out,
                                  int32_t* map_out, int n, int h, int w, int c, int block, int by, int bx,
                                  int thr, int clamp_to, elvis_stream_t stream) {
    ELVIS_REQUIRE(a && b && map && out, "elvis_recompose_u8: null pointer");
    ELVIS_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && block > 0 && by > 0 && bx > 0,
                  "elvis_recompose_u8: bad shape n=%d h=%d w=%d c=%d block=%d by=%d bx=%d", n, h, w, c, block, by, bx);
    ELVIS_REQUIRE(((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) % 16 == 0, "elvis_recompose_u8: pointers must be 16-byte aligned");
    long long total = (long long)n * h * w * c;
    long long nvec = (total + 15) >> 4;
    int grid = (int)((nvec + 255) / 256);
    if (grid > 256 * 16) grid = 256 * 16;
    if (((long long)w * c) % 16 == 0 && block * c >= 16) {
        hipLaunchKernelGGL(recompose_rows_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, map, out, h,
                           (w * c) / 16, block * c, block, by, bx, thr, nvec);
        ELVIS_CHECK_LAUNCH("elvis_recompose_u8");
        if (map_out) {
            int count = n * by * bx;
            hipLaunchKernelGGL(clamp_map_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, map,
                               map_out, count, thr, clamp_to);
            ELVIS_CHECK_LAUNCH("elvis_recompose_u8(map)");
        }
        return ELVIS_OK;
    }
    int bshift = 0;
    while ((1 << bshift) < block) ++bshift;
    if ((1 << bshift) == block)
        hipLaunchKernelGGL(recompose_u8_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, map, out, n, h, w,
                           c, block, bshift, by, bx, thr, total);
    else
        hipLaunchKernelGGL(recompose_u8_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, map, out, n, h, w,
                           c, block, bshift, by, bx, thr, total);
    ELVIS_CHECK_LAUNCH("elvis_recompose_u8");
    if (map_out) {
        int count = n * by * bx;
        hipLaunchKernelGGL(clamp_map_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, map,
                           map_out, count, thr, clamp_to);
        ELVIS_CHECK_LAUNCH("elvis_recompose_u8(map)");
    }
    return ELVIS_OK;
}

// ---------------------------------------------------------------------------------------
// area downscale: one lane per output pixel-channel group; rows of f*c contiguous bytes.
__global__ __launch_bounds__(256) void area_downscale_u8_kernel(const uint8_t* __restrict__ src,
                                                                uint8_t* __restrict__ dst, int n, int h, int w,
                                                                int c, int f, int rounding, long long total_out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int ho = h / f, wo = w / f;
    float inv = 1.0f / (float)(f * f);
    int area = f * f;
    for (; i < total_out; i += (long long)gridDim.x * blockDim.x) {
        long long px = i / c;
        int ch = (int)(i - px * c);
        long long row = px / wo;
        int xo = (int)(px - row * wo);
        int fi = (int)(row / ho);
        int yo = (int)(row - (long long)fi * ho);
        const uint8_t* base = src + (((long long)fi * h + (long long)yo * f) * w + (long long)xo * f) * c + ch;
        uint32_t s = 0;
        for (int dy = 0; dy < f; ++dy) {
            const uint8_t* r = base + (long long)dy * w * c;
            for (int dx = 0; dx < f; ++dx) s += r[dx * c];
        }
        uint32_t v;
        if (rounding == ELVIS_ROUND_HALF_UP || f == 2) {
            v = (s + area / 2) / area;
        } else {
            float prod = __fmul_rn((float)s, inv);
            v = (uint32_t)__float2int_rn(prod);  // round half to even, like cvRound
        }
        dst[i] = (uint8_t)(v > 255 ? 255 : v);
    }
}

// Fast path: c == 3, f == 4 (the 1080p SinSR 4x path): one lane per output pixel, reads four
// 12-byte row segments as 3 dwords each (rows are 4-byte aligned when w*3 % 4 == 0).
__global__ __launch_bounds__(256) void area_downscale4_c3_kernel(const uint8_t* __restrict__ src,
                                                                 uint8_t* __restrict__ dst, int n, int h, int w,
                                                                 int rounding, long long total_px) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int ho = h >> 2, wo = w >> 2;
    for (; i < total_px; i += (long long)gridDim.x * blockDim.x) {
        long long row = i / wo;
        int xo = (int)(i - row * wo);
        int fi = (int)(row / ho);
        int yo = (int)(row - (long long)fi * ho);
        const uint8_t* base = src + (((long long)fi * h + (long long)yo * 4) * w + (long long)xo * 4) * 3;
        uint32_t s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            const uint32_t* r = reinterpret_cast<const uint32_t*>(base + (long long)dy * w * 3);
            uint32_t d0 = r[0], d1 = r[1], d2 = r[2];
            // bytes: c0 c1 c2 c0 | c1 c2 c0 c1 | c2 c0 c1 c2
            s0 += (d0 & 0xFF) + (d0 >> 24) + ((d1 >> 16) & 0xFF) + ((d2 >> 8) & 0xFF);
            s1 += ((d0 >> 8) & 0xFF) + (d1 & 0xFF) + (d1 >> 24) + ((d2 >> 16) & 0xFF);
            s2 += ((d0 >> 16) & 0xFF) + ((d1 >> 8) & 0xFF) + (d2 & 0xFF) + (d2 >> 24);
        }
        uint32_t v0, v1, v2;
        if (rounding == ELVIS_ROUND_HALF_UP) {
            v0 = (s0 + 8) >> 4; v1 = (s1 + 8) >> 4; v2 = (s2 + 8) >> 4;
        } else {
            v0 = (uint32_t)__float2int_rn(__fmul_rn((float)s0, 0.0625f));
            v1 = (uint32_t)__float2int_rn(__fmul_rn((float)s1, 0.0625f));
            v2 = (uint32_t)__float2int_rn(__fmul_rn((float)s2, 0.0625f));
        }
        uint8_t* o = dst + i * 3;
        o[0] = (uint8_t)v0; o[1] = (uint8_t)v1; o[2] = (uint8_t)v2;
    }
}

extern "C" int elvis_area_downscale_u8(const uint8_t* src, uint8_t* dst, int n, int h, int w, int c, int factor,
                                       int rounding, elvis_stream_t stream) {
    ELVIS_REQUIRE(src && dst, "elvis_area_downscale_u8: null pointer");
    ELVIS_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && factor >= 1, "elvis_area_downscale_u8: bad shape");
    ELVIS_REQUIRE(h % factor == 0 && w % factor == 0, "elvis_area_downscale_u8: H,W (%d,%d) not divisible by factor %d", h, w, factor);
    ELVIS_REQUIRE(rounding == ELVIS_ROUND_CV2 || rounding == ELVIS_ROUND_HALF_UP, "elvis_area_downscale_u8: bad rounding");
    long long total_px = (long long)n * (h / factor) * (w / factor);
    if (c == 3 && factor == 4 && (w * 3) % 4 == 0 && ((uintptr_t)src % 4) == 0) {
        int grid = (int)((total_px + 255) / 256);
        if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(area_downscale4_c3_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, n, h,
                           w, rounding, total_px);
    } else {
        long long total = total_px * c;
        int grid = (int)((total + 255) / 256);
        if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(area_downscale_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, n, h, w,
                           c, factor, rounding, total);
    }
    ELVIS_CHECK_LAUNCH("elvis_area_downscale_u8");
    return ELVIS_OK;
}

// ---------------------------------------------------------------------------------------
// blend: bit-exact with numpy's float32 evaluation order of utils.py:1592-1599
//   w_rest = mask*alpha ; w_orig = 1 - w_rest ; out = trunc(clip(orig*w_orig + rest*w_rest))
__global__ __launch_bounds__(256) void blend_u8_kernel(const uint8_t* __restrict__ orig,
                                                       const uint8_t* __restrict__ rest,
                                                       const int32_t* __restrict__ map, uint8_t* __restrict__ out,
                                                       int n, int h, int w, int c, int block, int by, int bx,
                                                       float alpha, long long total) {
    long long vec = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long nvec = (total + 15) >> 4;
    for (; vec < nvec; vec += (long long)gridDim.x * blockDim.x) {
        long long g0 = vec << 4;
        BytePos p = decode_pos(g0, h, w, c);
        int nb = (int)((total - g0) < 16 ? (total - g0) : 16);
        uint8_t bo[16], br[16], bout[16];
        if (nb == 16) {
            *reinterpret_cast<uint4*>(bo) = *reinterpret_cast<const uint4*>(orig + g0);
            *reinterpret_cast<uint4*>(br) = *reinterpret_cast<const uint4*>(rest + g0);
        } else {
            for (int k = 0; k < nb; ++k) { bo[k] = orig[g0 + k]; br[k] = rest[g0 + k]; }
        }
        for (int k = 0; k < nb; ++k) {
            // INTER_NEAREST of the map to (w,h): src index = floor(dst * by / h)
            int byi = (int)(((long long)p.y * by) / h), bxi = (int)(((long long)p.x * bx) / w);
            if (byi > by - 1) byi = by - 1;
            if (bxi > bx - 1) bxi = bx - 1;
            float m = map[(p.f * by + byi) * bx + bxi] > 0 ? 1.0f : 0.0f;
            float w_rest = __fmul_rn(m, alpha);
            float w_orig = __fsub_rn(1.0f, w_rest);
            float v = __fadd_rn(__fmul_rn((float)bo[k], w_orig), __fmul_rn((float)br[k], w_rest));
            v = fminf(fmaxf(v, 0.0f), 255.0f);
            bout[k] = (uint8_t)(int)v;  // truncation, like ndarray.astype(np.uint8)
            advance_pos(p, h, w, c);
        }
        if (nb == 16) *reinterpret_cast<uint4*>(out + g0) = *reinterpret_cast<uint4*>(bout);
        else for (int k = 0; k < nb; ++k) out[g0 + k] = bout[k];
    }
}

extern "C" int elvis_blend_u8(const uint8_t* orig, const uint8_t* rest, const int32_t* map, uint8_t* out, int n,
                              int h, int w, int c, int block, int by, int bx, float alpha, elvis_stream_t stream) {
    ELVIS_REQUIRE(orig && rest && map && out, "elvis_blend_u8: null pointer");
    ELVIS_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && block > 0 && by > 0 && bx > 0, "elvis_blend_u8: bad shape");
    ELVIS_REQUIRE(by == h / block && bx == w / block, "elvis_blend_u8: map grid (%d,%d) != floor(H/b),floor(W/b)", by, bx);
    ELVIS_REQUIRE(((uintptr_t)orig | (uintptr_t)rest | (uintptr_t)out) % 16 == 0, "elvis_blend_u8: pointers must be 16-byte aligned");
    long long total = (long long)n * h * w * c;
    long long nvec = (total + 15) >> 4;
    int grid = (int)((nvec + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(blend_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, orig, rest, map, out, n, h, w,
                       c, block, by, bx, alpha, total);
    ELVIS_CHECK_LAUNCH("elvis_blend_u8");
    return ELVIS_OK;
}

// ---------------------------------------------------------------------------------------
// per-level select (presley.py:1262-1273)
__global__ __launch_bounds__(256) void select_levels_u8_kernel(const uint8_t* const* __restrict__ versions,
                                                               const int32_t* __restrict__ slot_of_level,
                                                               int n_levels, const int32_t* __restrict__ map,
                                                               uint8_t* __restrict__ out, int n, int h, int w,
                                                               int c, int block, int by, int bx, long long total) {
    long long vec = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long nvec = (total + 15) >> 4;
    for (; vec < nvec; vec += (long long)gridDim.x * blockDim.x) {
        long long g0 = vec << 4;
        BytePos p = decode_pos(g0, h, w, c);
        int nb = (int)((total - g0) < 16 ? (total - g0) : 16);
        uint8_t bout[16];
        int last_key = -1;
        const uint8_t* srcp = nullptr;
        for (int k = 0; k < nb; ++k) {
            int byi = p.y / block, bxi = p.x / block;
            uint8_t v = 0;
            if (byi < by && bxi < bx) {
                int key = (p.f * by + byi) * bx + bxi;
                if (key != last_key) {
                    last_key = key;
                    int lvl = map[key];
                    int slot = (lvl >= 0 && lvl < n_levels) ? slot_of_level[lvl] : -1;
                    srcp = slot >= 0 ? versions[slot] : nullptr;
                }
                if (srcp) v = srcp[g0 + k];
            }
            bout[k] = v;
            advance_pos(p, h, w, c);
        }
        if (nb == 16) *reinterpret_cast<uint4*>(out + g0) = *reinterpret_cast<uint4*>(bout);
        else for (int k = 0; k < nb; ++k) out[g0 + k] = bout[k];
    }
}

extern "C" int elvis_select_levels_u8(const uint8_t* const* versions, const int32_t* slot_of_level, int n_levels,
                                      const int32_t* map, uint8_t* out, int n, int h, int w, int c, int block,
                                      int by, int bx, elvis_stream_t stream) {
    ELVIS_REQUIRE(versions && slot_of_level && map && out, "elvis_select_levels_u8: null pointer");
    ELVIS_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && block > 0 && by > 0 && bx > 0 && n_levels > 0,
                  "elvis_select_levels_u8: bad shape");
    ELVIS_REQUIRE((uintptr_t)out % 16 == 0, "elvis_select_levels_u8: out must be 16-byte aligned");
    long long total = (long long)n * h * w * c;
    long long nvec = (total + 15) >> 4;
    int grid = (int)((nvec + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(select_levels_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, versions,
                       slot_of_level, n_levels, map, out, n, h, w, c, block, by, bx, total);
    ELVIS_CHECK_LAUNCH("elvis_select_levels_u8");
    return ELVIS_OK;
}

// ---------------------------------------------------------------------------------------
// feathered tile accumulate (utils.py:275-314), bit-exact with numpy's evaluation order:
//   sw = f32( f64( f32( f64(wy[y]) * wx1[x] ) ) * wx2[x] )  (two in-place f32 *= f64 ramps: left, right)
//   total = sw * f32(tw)
//   acc += f32(tile) * total ; wsum += total
__global__ __launch_bounds__(256) void tile_accumulate_kernel(float* __restrict__ acc, float* __restrict__ wsum,
                                                              const uint8_t* __restrict__ tile,
                                                              const float* __restrict__ wy,
                                                              const double* __restrict__ wx,
                                                              const double* __restrict__ wx2, int h, int w, int y0,
                                                              int x0, int th, int tw, int c, float temporal_weight) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= th * tw) return;
    int ty = i / tw, tx = i - ty * tw;
    float sw = (float)((double)wy[ty] * wx[tx]);
    sw = (float)((double)sw * wx2[tx]);
    float total = __fmul_rn(sw, temporal_weight);
    long long o = ((long long)(y0 + ty) * w + (x0 + tx));
    const uint8_t* t = tile + (long long)i * c;
    for (int ch = 0; ch < c; ++ch) {
        float v = __fmul_rn((float)t[ch], total);
        acc[o * c + ch] = __fadd_rn(acc[o * c + ch], v);
    }
    wsum[o] = __fadd_rn(wsum[o], total);
}

extern "C" int elvis_tile_accumulate_f32(float* acc, float* wsum, const uint8_t* tile, const float* wy,
                                         const double* wx, const double* wx2, int h, int w, int y0, int x0,
                                         int th, int tw, int c, float temporal_weight, elvis_stream_t stream) {
    ELVIS_REQUIRE(acc && wsum && tile && wy && wx && wx2, "elvis_tile_accumulate_f32: null pointer");
    ELVIS_REQUIRE(th > 0 && tw > 0 && y0 >= 0 && x0 >= 0 && y0 + th <= h && x0 + tw <= w && c > 0,
                  "elvis_tile_accumulate_f32: tile (%d,%d,%d,%d) outside frame (%d,%d)", y0, x0, th, tw, h, w);
    int total = th * tw;
    hipLaunchKernelGGL(tile_accumulate_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, acc,
                       wsum, tile, wy, wx, wx2, h, w, y0, x0, th, tw, c, temporal_weight);
    ELVIS_CHECK_LAUNCH("elvis_tile_accumulate_f32");
    return ELVIS_OK;
}

__global__ __launch_bounds__(256) void tile_normalize_kernel(const float* __restrict__ acc,
                                                             const float* __restrict__ wsum,
                                                             uint8_t* __restrict__ out, long long pixels, int c) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < pixels; i += (long long)gridDim.x * blockDim.x) {
        float ws = wsum[i];
        float safe = ws > 0.0f ? ws : 1.0f;
        for (int ch = 0; ch < c; ++ch) {
            float v = __fdiv_rn(acc[i * c + ch], safe);
            v = fminf(fmaxf(v, 0.0f), 255.0f);
            out[i * c + ch] = (uint8_t)(int)v;
        }
    }
}

extern "C" int elvis_tile_normalize_u8(const float* acc, const float* wsum, uint8_t* out, int h, int w, int c,
                                       elvis_stream_t stream) {
    ELVIS_REQUIRE(acc && wsum && out && h > 0 && w > 0 && c > 0, "elvis_tile_normalize_u8: bad argument");
    long long pixels = (long long)h * w;
    int grid = (int)((pixels + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(tile_normalize_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, acc, wsum, out, pixels, c);
    ELVIS_CHECK_LAUNCH("elvis_tile_normalize_u8");
    return ELVIS_OK;
}

// ---------------------------------------------------------------------------------------
// SSE per frame (integer-exact: sum of squared u8 differences fits in u64).
__global__ __launch_bounds__(256) void sse_u8_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                     const uint8_t* __restrict__ mask,
                                                     unsigned long long* __restrict__ sse,
                                                     unsigned long long* __restrict__ cnt, long long per_frame,
                                                     int c) {
    int f = blockIdx.y;
    const uint8_t* pa = a + (long long)f * per_frame;
    const uint8_t* pb = b + (long long)f * per_frame;
    const uint8_t* pm = mask ? mask + (long long)f * (per_frame / c) : nullptr;
    unsigned long long s = 0, k = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per_frame;
         i += (long long)gridDim.x * blockDim.x) {
        bool use = pm ? pm[i / c] != 0 : true;
        if (use) {
            int d = (int)pa[i] - (int)pb[i];
            s += (unsigned long long)(d * d);
            ++k;
        }
    }
    // wave reduce then one atomic per wave
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        k += __shfl_xor(k, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&sse[f], s);
        atomicAdd(&cnt[f], k);
    }
}

extern "C" int elvis_sse_u8(const uint8_t* a, const uint8_t* b, const uint8_t* mask, unsigned long long* sse_out,
                            unsigned long long* cnt_out, int n, int h, int w, int c, elvis_stream_t stream) {
    ELVIS_REQUIRE(a && b && sse_out && cnt_out && n > 0 && h > 0 && w > 0 && c > 0, "elvis_sse_u8: bad argument");
    long long per_frame = (long long)h * w * c;
    int gx = (int)((per_frame + 256 * 16 - 1) / (256 * 16));
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(sse_u8_kernel, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, a, b, mask, sse_out, cnt_out,
                       per_frame, c);
    ELVIS_CHECK_LAUNCH("elvis_sse_u8");
    return ELVIS_OK;
}

// ---------------------------------------------------------------------------------------
// per-block SSIM (utils.py:572-608: pytorch_msssim.ssim on every block_size x block_size patch, data_range 1,
// 11-tap Gaussian window sigma 1.5, K = (0.01, 0.03), size_average=False).  pytorch_msssim smooths a dimension only
// when it is at least as long as the window ("valid" convolution) and skips it otherwise, so for blocks smaller than
// 11 pixels the local means are the pixels themselves and the structure term is identically 1.  One thread per
// (frame, block): float32 throughout, channels averaged last.  win11: the normalised window (device pointer).
__global__ __launch_bounds__(64) void block_ssim_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                        float* __restrict__ out, const float* __restrict__ win11, int n,
                                                        int h, int w, int c, int bs, int by, int bx, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int bxi = (int)(i % bx);
    const long long t = i / bx;
    const int byi = (int)(t % by);
    const int f = (int)(t / by);
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const bool smooth = bs >= 11;
    const int m = smooth ? bs - 10 : bs;   // side of the SSIM map
    float wv[11];
    for (int k = 0; k < 11; ++k) wv[k] = win11[k];
    float ch_sum = 0.f;
    for (int ch = 0; ch < c; ++ch) {
        const uint8_t* pa = a + (((long long)f * h + (long long)byi * bs) * w + (long long)bxi * bs) * c + ch;
        const uint8_t* pb = b + (((long long)f * h + (long long)byi * bs) * w + (long long)bxi * bs) * c + ch;
        const long long rs = (long long)w * c;
        float acc = 0.f;
        for (int y = 0; y < m; ++y)
            for (int x = 0; x < m; ++x) {
                float mu1, mu2, xx, yy, xy;
                if (!smooth) {
                    const float u = __fdiv_rn((float)pa[y * rs + (long long)x * c], 255.0f), v = __fdiv_rn((float)pb[y * rs + (long long)x * c], 255.0f);
                    mu1 = u; mu2 = v; xx = u * u; yy = v * v; xy = u * v;
                } else {
                    mu1 = mu2 = xx = yy = xy = 0.f;
                    for (int dy = 0; dy < 11; ++dy) {
                        float r1 = 0.f, r2 = 0.f, rxx = 0.f, ryy = 0.f, rxy = 0.f;
                        for (int dx = 0; dx < 11; ++dx) {
                            const float u = __fdiv_rn((float)pa[(y + dy) * rs + (long long)(x + dx) * c], 255.0f);
                            const float v = __fdiv_rn((float)pb[(y + dy) * rs + (long long)(x + dx) * c], 255.0f);
                            r1 += wv[dx] * u; r2 += wv[dx] * v; rxx += wv[dx] * (u * u); ryy += wv[dx] * (v * v); rxy += wv[dx] * (u * v);
                        }
                        mu1 += wv[dy] * r1; mu2 += wv[dy] * r2; xx += wv[dy] * rxx; yy += wv[dy] * ryy; xy += wv[dy] * rxy;
                    }
                }
                const float s1 = xx - mu1 * mu1, s2 = yy - mu2 * mu2, s12 = xy - mu1 * mu2;
                const float cs = (2.f * s12 + C2) / (s1 + s2 + C2);
                acc += ((2.f * mu1 * mu2 + C1) / (mu1 * mu1 + mu2 * mu2 + C1)) * cs;
            }
        ch_sum += acc / (float)(m * m);
    }
    out[i] = ch_sum / (float)c;
}

extern "C" int elvis_block_ssim_u8(const uint8_t* a, const uint8_t* b, float* ssim_out, const float* win11, int n, int h,
                                   int w, int c, int block_size, elvis_stream_t stream) {
    ELVIS_REQUIRE(a && b && ssim_out && win11, "elvis_block_ssim_u8: null pointer");
    ELVIS_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && block_size > 0 && block_size <= h && block_size <= w,
                  "elvis_block_ssim_u8: bad shape");
    const int by = h / block_size, bx = w / block_size;   // floored grid, like utils.py:580-584
    const long long total = (long long)n * by * bx;
    hipLaunchKernelGGL(block_ssim_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a, b, ssim_out,
                       win11, n, h, w, c, block_size, by, bx, total);
    ELVIS_CHECK_LAUNCH("elvis_block_ssim_u8");
    return ELVIS_OK;
}
