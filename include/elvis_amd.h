/*
 * elvis_amd.h - C ABI of the MI355X-native ELVIS client-side restoration hot path.
 *
 * The reference (emanuele-artioli/elvis) is pure Python and has NO FFI of its own
 * (SURVEY.md F2, section 8b): its drop-in boundary is three Python callable protocols
 * (P1 upsample_fn, P2 process_fn, P3 restore_fn) plus the sharding helpers.  This header
 * is the C-ABI that the Python shim in `elvis_amd/` calls through ctypes to implement
 * those protocols; each entry point cites the reference code whose arithmetic it
 * replaces.  INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *  - every function returns 0 on success, a negative ELVIS_E_* code on failure; the
 *    message for the calling thread is available from elvis_last_error().
 *  - all pointers are DEVICE pointers (HBM) unless the name ends in `_host`.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *  - no entry point allocates, frees or synchronises: callers own all buffers
 *    (hipGraph-capturable, cdna guide G9).
 *  - images are NHWC, C-contiguous.  u8 frames are (n,h,w,3).  Float tensors carry an
 *    explicit channel pitch (`*_pitch`, in elements, multiple of 8) so producers can write
 *    straight into channel slices of a wider buffer.
 *  - dtype codes: ELVIS_F32 = 0 (exact-parity mode), ELVIS_F16 = 1 (MFMA fast mode,
 *    fp32 accumulate); elvis_conv2d / elvis_conv_pack_weights also take ELVIS_F32X3 = 2
 *    (fp32 tensors, f16 MFMA with the rounding error compensated: fp32-grade results at
 *    2-2.5x the fp32 MFMA's rate, for the shapes elvis_conv_x3_eligible accepts).
 */
#ifndef ELVIS_AMD_H
#define ELVIS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ELVIS_ABI_VERSION 1

#define ELVIS_OK 0
#define ELVIS_E_INVALID (-1)   /* bad shape / argument  -> Python ValueError   */
#define ELVIS_E_RUNTIME (-2)   /* HIP launch failure    -> Python RuntimeError */
#define ELVIS_E_UNSUPPORTED (-3)

#define ELVIS_F32 0
#define ELVIS_F16 1
#define ELVIS_F32X3 2   /* conv only: fp32 tensors; products on the f16 matrix pipe with the rounding error compensated
                         (operands split hi + lo, ~1e-6 relative), 2-4x the fp32 MFMA's rate.  See elvis_conv_x3_eligible.
                         OPERAND RANGE: hi = f16(v), so every activation (after the fused prologue) and weight must
                         satisfy |v| < 65504; beyond that the result is non-finite (inf / NaN), never silently wrong.
                         Run a layer whose operands can exceed the f16 range as ELVIS_F32. */

#define ELVIS_ROUND_CV2 0      /* 2x2: (s+2)>>2 ; else rint(float(s)*(1.f/area)) half-even */
#define ELVIS_ROUND_HALF_UP 1  /* (s + area/2) / area */

typedef void* elvis_stream_t;

int elvis_abi_version(void);
const char* elvis_last_error(void);

/* ------------------------------------------------------------------ block-map glue (u8) */

/* out[blk(i,j)] = (map[n,i,j] <= thr) ? a : b, per b x b block; pixels outside the
 * by*block x bx*block grid take `b`.  Replaces the python By x Bx paste loop of
 * upscale_realesrgan_adaptive (elvis.py:2584-2595) and the boolean-mask block copy of
 * _instantir_chunk_worker (elvis.py:2975-2978).  If map_out != NULL it receives the clamped
 * map of elvis.py:2592: map_out = (map <= thr) ? map : clamp_to. */
int elvis_recompose_u8(const uint8_t* a, const uint8_t* b, const int32_t* map, uint8_t* out,
                       int32_t* map_out, int n, int h, int w, int c, int block, int by, int bx,
                       int thr, int clamp_to, elvis_stream_t stream);

/* Integer-factor box-mean downscale; replaces cv2.resize(..., INTER_AREA) at
 * elvis.py:2565 and elvis.py:2581.  h,w must be divisible by factor. */
int elvis_area_downscale_u8(const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                            int factor, int rounding, elvis_stream_t stream);

/* out = trunc(clip(orig*(1-alpha*m) + rest*(alpha*m), 0, 255)), m = (map>0) upsampled by
 * `block` (nearest).  Replaces blended_restoration's per-frame fp32 blend (utils.py:1581-1599). */
int elvis_blend_u8(const uint8_t* orig, const uint8_t* rest, const int32_t* map, uint8_t* out,
                   int n, int h, int w, int c, int block, int by, int bx, float alpha,
                   elvis_stream_t stream);

/* Per-level select: out block (i,j) of frame f = versions[level_slot[map[f,i,j]]][f] block;
 * pixels outside the floored grid are 0.  `versions` is a DEVICE array of n_versions device
 * pointers, each (n,h,w,c) u8; `slot_of_level` is a DEVICE int32 table of size n_levels.
 * Replaces the F x By x Bx python loop of restore_video_adaptively (presley.py:1262-1273). */
int elvis_select_levels_u8(const uint8_t* const* versions, const int32_t* slot_of_level, int n_levels,
                           const int32_t* map, uint8_t* out, int n, int h, int w, int c, int block,
                           int by, int bx, elvis_stream_t stream);

/* Feathered tile accumulate, bit-exact with numpy's float32 evaluation order:
 *   sw = f32(f64(f32(f64(wy[y]) * wx[x])) * wx2[x]); wgt = sw * temporal_weight;
 *   acc[y0+y, x0+x, :] += f32(tile) * wgt;  wsum[y0+y, x0+x] += wgt.
 * wy (th floats) holds the top/bottom ramps already applied in float32, wx / wx2 (tw doubles)
 * the left / right np.linspace ramps (1.0 where no ramp).  Replaces resource_aware_restore's
 * blend loop (utils.py:275-314).  acc is (h,w,c) f32, wsum (h,w) f32, tile (th,tw,c) u8. */
int elvis_tile_accumulate_f32(float* acc, float* wsum, const uint8_t* tile, const float* wy,
                              const double* wx, const double* wx2, int h, int w, int y0, int x0, int th,
                              int tw, int c, float temporal_weight, elvis_stream_t stream);

/* out = trunc(clip(acc / (wsum>0 ? wsum : 1), 0, 255)) (utils.py:317-324). */
int elvis_tile_normalize_u8(const float* acc, const float* wsum, uint8_t* out, int h, int w, int c,
                            elvis_stream_t stream);

/* Integer-exact sum of squared u8 differences per frame, for PSNR/MSE (elvis.py:627-671,
 * presley.py:226-245).  mask (n,h,w) u8 may be NULL.  sse_out / cnt_out: n u64 each (sum, number
 * of compared elements); both must be zeroed by the caller. */
int elvis_sse_u8(const uint8_t* a, const uint8_t* b, const uint8_t* mask, unsigned long long* sse_out,
                 unsigned long long* cnt_out, int n, int h, int w, int c, elvis_stream_t stream);

/* Per-block SSIM (utils.py:572-608 = pytorch_msssim.ssim per block_size x block_size patch: data_range 1, 11-tap
 * Gaussian window sigma 1.5 passed in as win11 (device f32[11], normalised), "valid" smoothing that is skipped for
 * blocks shorter than the window, K = (0.01, 0.03), mean over the map then over channels).  The block grid is
 * floored (H // block_size, W // block_size); ssim_out: f32 [n, H // b, W // b]. */
int elvis_block_ssim_u8(const uint8_t* a, const uint8_t* b, float* ssim_out, const float* win11, int n, int h, int w, int c,
                        int block_size, elvis_stream_t stream);

/* ------------------------------------------------------------------ u8 <-> float */

/* dst[n,h,w,pitch] = (div255 ? src_u8/255 : src_u8) * scale + bias for the first 3 channels
 * (optionally swapping R and B), zero for the pad channels.  Replaces the cvtColor/PIL//255 conversions of
 * elvis.py:2959-2960 and RealESRGANer.enhance's pre-processing (elvis.py:2515). */
int elvis_u8_to_float(const uint8_t* src, void* dst, int dtype, int n, int h, int w, int pitch,
                      float scale, float bias, int swap_rb, int div255, elvis_stream_t stream);

/* t = clip(src*scale + bias, 0, 1); dst_u8 = cast(t*255); mode 0 = round-half-even, 1 = truncate
 * (utils.py:324).  Optionally also writes the pre-quantisation t as f32 (n,h,w,3) to `f32_out`
 * (may be NULL) for the max-abs parity report. */
int elvis_float_to_u8(const void* src, int dtype, uint8_t* dst, float* f32_out, int n, int h, int w,
                      int pitch, float scale, float bias, int mode, int swap_rb, elvis_stream_t stream);

/* ------------------------------------------------------------------ model kernels */

typedef struct elvis_conv_desc {
    int dtype;          /* ELVIS_F32 / ELVIS_F16 (activations, weights; fp32 accumulate)      */
    int n, h, w;        /* INPUT spatial size (before the optional nearest 2x upsample)        */
    int cin, cin_pitch; /* channels of input 1 and its pitch                                   */
    int cin2, cin2_pitch; /* optional second input (virtual channel concat), 0 if unused       */
    int cout, cout_pitch; /* logical output channels, output pitch                             */
    int ksize;          /* 1 or 3 (2: one parity of a sub-pixel upsample conv, see `subpixel`)  */
    int stride;         /* 1 or 2                                                              */
    int pad_before;     /* zero padding before (top/left); after is implied by ho/wo           */
    int upsample;       /* 1: input is nearest-upsampled 2x before the conv                    */
    int ho, wo;         /* output spatial size                                                 */
    int act;            /* epilogue activation: 0 none, 1 GELU(erf), 2 SiLU, 3 ReLU            */
    int prologue;       /* 0 none, 1: x <- silu(x*pa[n,c]+pb[n,c]) on load (fused GroupNorm)   */
    int subpixel;       /* ksize == 2 only.  1..4: 1 + parity (2a+b) of the sub-pixel decomposition of
                           "nearest-2x upsample + 3x3 conv": this launch writes output pixels
                           (2y+a, 2x+b) of the 2h x 2w output from 2x2 pre-summed taps.
                           ELVIS_CONV_S2D (5): the space-to-depth form of "pad (0,1,0,1) + 3x3 conv,
                           stride 2" (the autoencoder's Downsample): x is the full-res h x w tensor of
                           C = cin/4 channels, the kernel reads its four (row, column) phases as
                           4C channels of an ho x wo image (ho = h/2, wo = w/2) and applies a 2x2
                           conv whose OIHW weights [cout][4C][2][2] hold W[2ry+py][2rx+px] at input
                           channel (2py+px)*C + c, tap (ry, rx) (zero where 2r+p > 2).  f16, C % 32 == 0.
                           With pad_before = 1 the same kernel runs "3x3 conv, stride 2, pad 1" (the Blur / DCT
                           slots' down convs): the 2x2 taps then sit at phase rows y-1, y and hold
                           W[2ry+py-1][2rx+px-1] (zero where 2r+p < 1).  */
} elvis_conv_desc;
#define ELVIS_CONV_S2D 5

/* Number of bytes of the packed weight buffer for a conv (depends on cin/cin2/cout/ksize/dtype). */
size_t elvis_conv_packed_weight_bytes(const elvis_conv_desc* d);

/* Pack PyTorch OIHW fp32 weights (host or device pointer given by `w_oihw_device`) into the
 * kernel's [tap][kchunk][cout_pad][kvec] layout, converting to `dtype`. */
int elvis_conv_pack_weights(const elvis_conv_desc* d, const float* w_oihw, void* packed,
                            elvis_stream_t stream);

/* y = act(conv(prologue(x [,x2])) + bias) + residual.  Implicit-GEMM on MFMA.  `bias` f32[cout] or
 * NULL, `residual` same dtype as out or NULL, `pa`,`pb` f32[n, cin+cin2] for the prologue.
 * `stats` (may be NULL): per-tile GroupNorm partial sums of the STORED output,
 * f32[elvis_conv_stats_tiles(d)][cout][2]; only for convs where elvis_conv_stats_tiles(d) > 0
 * (3x3 / stride 1 / pad 1 / cout >= 64, the LDS halo-tile kernel).
 * This is the slot where the reference calls RealESRGANer.enhance (elvis.py:2515) /
 * restore_images_batch (elvis.py:2963-2970): the conv/linear layers of the restorer. */
int elvis_conv2d(const elvis_conv_desc* d, const void* x, const void* x2, const void* w_packed,
                 const float* bias, const void* residual, int residual_pitch, const float* pa,
                 const float* pb, void* out, float* stats, elvis_stream_t stream);

/* Number of per-tile statistics rows `elvis_conv2d` writes for this conv (n * tiles per image),
 * 0 when the conv cannot produce fused statistics. */
int elvis_conv_stats_tiles(const elvis_conv_desc* d);

/* Writes the name of the kernel instantiation `elvis_conv2d` dispatches this conv to (the template
 * name rocprofv3's kernel trace shows, e.g. "conv3x3_halo_kernel<half,128,256,6,true,3,false>")
 * into buf (NUL-terminated, truncated to n).  For per-kernel profiling (bench.py roofline). */
int elvis_conv_kernel_name(const elvis_conv_desc* d, char* buf, size_t n);

/* Test / experiment switch: "no_halo" = 1 routes every conv to the generic implicit-GEMM kernel (what the environment
 * variable ELVIS_NO_HALO does for a whole process), 0 forces the halo kernels, -1 returns to the environment's choice. */
int elvis_conv_debug_set(const char* key, int value);

/* 1 when a descriptor with dtype ELVIS_F32X3 has a compensated-f16 kernel (3x3 stride 1 / sub-pixel 2x2 / 1x1 on the
 * halo-tile kernels with a 64- or 128-channel output tile), else 0.  ELVIS_F32X3 weights are packed by
 * elvis_conv_pack_weights as f16 (hi, lo) parts - planes of 32 channels for 3x3 layers (three MFMAs per 32 channels),
 * interleaved pairs otherwise (four) - which only those kernels read: run every other conv as ELVIS_F32 with weights
 * packed as ELVIS_F32 (elvis_conv2d refuses an ELVIS_F32X3 descriptor that is not eligible). */
int elvis_conv_x3_eligible(const elvis_conv_desc* d);

/* sums[n, sums_coff + c, 0..1] = sum over the image's tiles of partials[tile][c][0..1] (f64). */
int elvis_gn_partials_to_sums(const float* partials, int tiles_per_image, int n, int c, double* sums,
                              int sums_ctot, int sums_coff, elvis_stream_t stream);

/* GroupNorm statistics of a tensor that has no fused statistics: per-(n,channel) sum and sum of
 * squares into sums[n, sums_ctot, 2] f64 at channels [sums_coff, sums_coff+c) (a virtual concat of
 * two tensors shares one buffer).  One partial row per workgroup is written to `workspace`
 * (elvis_groupnorm_workspace_floats(...) floats) and reduced in a fixed order: bit-reproducible,
 * no atomics. */
size_t elvis_groupnorm_workspace_floats(int dtype, int n, int hw, int c);
int elvis_groupnorm_sums(const void* x, int dtype, int n, int hw, int c, int pitch, double* sums,
                         int sums_ctot, int sums_coff, float* workspace, elvis_stream_t stream);

/* Turn sums into per-(n,channel) affine pa,pb so that GN(x)*(1+scale)+shift == x*pa+pb.
 * gamma,beta f32[c]; scale,shift f32[c] may be NULL (then 0). */
int elvis_groupnorm_affine(const double* sums, const float* gamma, const float* beta, const float* scale,
                           const float* shift, float* pa, float* pb, int n, int hw, int c, int groups,
                           float eps, elvis_stream_t stream);

/* y = act(x*pa[n,c] + pb[n,c]); act 0 none / 2 SiLU.  In-place allowed. */
int elvis_affine_act(const void* x, void* y, int dtype, int n, int hw, int c, int pitch_in,
                     int pitch_out, const float* pa, const float* pb, int act, elvis_stream_t stream);

/* LayerNorm over the channel dim of each token. */
int elvis_layernorm(const void* x, void* y, int dtype, long long tokens, int c, int pitch_in,
                    int pitch_out, const float* gamma, const float* beta, float eps, elvis_stream_t stream);

/* Fused per-token blocks of a Swin layer, f16 tensors, fp32 accumulate (csrc/swin.hip; model slots a5 / a7 of SURVEY.md 8a:
 * the reference delegates these layers to absent pip packages, there is no reference file:line for them).
 *   elvis_swin_mlp       : out = x + fc2(GELU(fc1(LayerNorm(x))))   - one launch, the hidden tensor never reaches HBM
 *   elvis_swin_ln_linear : out = W . LayerNorm(x) + bias            - LayerNorm fused into a projection (qkv)
 * c (channels) in {64, 128, 192, 256}; hidden / n_out multiples of 64; x[tokens, x_pitch], out[tokens, out_pitch] f16;
 * biases, gamma, beta f32; all pointers 16-byte aligned.  Weights are packed once by elvis_swin_pack_weights from
 * row-major f32 matrices w1[n1, c] (fc1 or the projection) and, for the MLP, w2[c, n1] (fc2) into
 * elvis_swin_packed_bytes(c, n1, mlp) bytes. */
size_t elvis_swin_packed_bytes(int c, int n1, int mode);   /* mode 0: LN + linear, 1: MLP, 2: projection + MLP */
int elvis_swin_pack_weights(const float* w1, const float* w2, void* packed, int c, int n1, int mlp, elvis_stream_t stream);
int elvis_swin_mlp(const void* x, void* out, const void* packed, const float* b1, const float* b2, const float* gamma,
                   const float* beta, long long tokens, int c, int hidden, int x_pitch, int out_pitch, float eps,
                   elvis_stream_t stream);
int elvis_swin_ln_linear(const void* x, void* out, const void* packed, const float* bias, const float* gamma, const float* beta,
                         long long tokens, int c, int n_out, int x_pitch, int out_pitch, float eps, elvis_stream_t stream);
/*   elvis_swin_proj_mlp  : y' = y + Wp . attn + bp ; out = y' + fc2(GELU(fc1(LayerNorm(y'))))  - the attention output projection
 *                          folded in front of the MLP (y' never reaches HBM).  Weights packed by elvis_swin_pack_proj_mlp from
 *                          wp[c, c], w1[hidden, c], w2[c, hidden] into elvis_swin_packed_bytes(c, hidden, 2) bytes. */
int elvis_swin_pack_proj_mlp(const float* wp, const float* w1, const float* w2, void* packed, int c, int hidden, elvis_stream_t stream);
int elvis_swin_proj_mlp(const void* attn, const void* y, void* out, const void* packed, const float* bp, const float* b1,
                        const float* b2, const float* gamma, const float* beta, long long tokens, int c, int hidden,
                        int attn_pitch, int y_pitch, int out_pitch, float eps, elvis_stream_t stream);

/* Swin (shifted-)window attention on a token image qkv[n,h,w,3*E] (q|k|v, head-major inside
 * each), window ws, `shift` cyclic shift (0 or ws/2) with the standard region mask, relative
 * position bias table [(2ws-1)^2, heads] f32.  out[n,h,w,E] in image order (un-shifted). */
int elvis_window_attention(const void* qkv, void* out, int dtype, int n, int h, int w, int heads,
                           int head_dim, int ws, int shift, int qkv_pitch, int out_pitch,
                           const float* bias_table, float scale, elvis_stream_t stream);

/* PyTorch bicubic (A=-0.75, align_corners=False) x`sf` upsample of a float NHWC image. */
int elvis_bicubic_upsample(const void* x, void* y, int dtype, int n, int h, int w, int c, int pitch_in,
                           int pitch_out, int sf, elvis_stream_t stream);

/* Nearest-codebook lookup: zq = codebook[argmin_k sum_c (z_c - e_kc)^2] (first index wins). */
int elvis_vq_nearest(const void* z, void* zq, int32_t* idx_out, int dtype, long long pixels, int c,
                     int pitch_in, int pitch_out, const float* codebook, int n_embed, elvis_stream_t stream);

/* Reflect-pad (right/bottom) copy of a float NHWC image into a larger one, optionally into a
 * channel slice, with y = x*mul + add_mul*add[...] (used for x_T = z_y + kappa*sqrt(eta)*eps
 * and the 1/sqrt(eta*kappa^2+1) input scaling; `add` is f32 NCHW noise or NULL). */
int elvis_pad_reflect_axpy(const void* x, void* y, int dtype, int n, int h, int w, int c, int pitch_in,
                           int hp, int wp, int pitch_out, int ch_offset_out, float mul, const float* add,
                           float add_mul, elvis_stream_t stream);

/* y[n,h,w,:c] = x[n,:h,:w,:c] crop-copy between pitched tensors (dtype-preserving). */
int elvis_crop_copy(const void* x, void* y, int dtype, int n, int h_in, int w_in, int pitch_in, int h,
                    int w, int c, int pitch_out, elvis_stream_t stream);

/* y = (dst dtype) x for a pitched NHWC tensor of `pixels` x `pitch` elements (pitch % 8 == 0; f16 <-> f32):
 * the section boundaries of the mixed-precision mode.  No reference counterpart. */
int elvis_convert_act(const void* x, int src_dtype, void* y, int dst_dtype, long long pixels, int pitch,
                      elvis_stream_t stream);

/* ------------------------------------------------------------------ server-side degrade filters
 * Every block_size x block_size block of a uint8 NHWC frame is filtered as its own image (nothing leaks between
 * blocks); map[n, by, bx] int32 with by = H / block_size, bx = W / block_size (H, W divisible by block_size). */

/* filter_frame_downsample (elvis.py:2141-2169): per block, INTER_AREA downscale by 2**level then INTER_LINEAR
 * back to block_size (OpenCV's u8 fixed-point rules restated; block_size a power of two <= 16). */
int elvis_degrade_downsample_u8(const uint8_t* src, const int32_t* levels, uint8_t* dst, int n, int h, int w, int c,
                                int block_size, int by, int bx, elvis_stream_t stream);

/* filter_frame_gaussian (elvis.py:2171-2196): per block, `rounds` passes of a separable 5-tap Gaussian with the
 * symmetric taps (tap0, tap1, tap2, tap1, tap0), BORDER_REFLECT_101 at the block's own edges, float32
 * arithmetic, round-half-even to uint8 after every pass pair (block_size <= 16). */
int elvis_degrade_gaussian_u8(const uint8_t* src, const int32_t* rounds, uint8_t* dst, int n, int h, int w, int c,
                              int block_size, int by, int bx, float tap0, float tap1, float tap2, elvis_stream_t stream);

/* DCT-coefficient dampening (the build's definition of ELVIS v2 DCT's degrade; README.md:44 names it, the reference
 * holds no code): 8x8 blocks; basis64 = f32[8][8] DCT-II basis, gain = f32[n_levels][8][8] per-level coefficient
 * gains (both device pointers), level clamped to [0, n_levels). */
int elvis_degrade_dct_u8(const uint8_t* src, const int32_t* levels, uint8_t* dst, const float* basis64, const float* gain,
                         int n_levels, int n, int h, int w, int c, int by, int bx, elvis_stream_t stream);

/* ------------------------------------------------------------------ DCT slot (LaplacianVCAR-style) */

/* DCNv2 modulated deformable 3x3 convolution (stride 1, pad 1, dilation 1), NHWC.
 * offset_mask[n,h,w,om_pitch]: channels [0, 18*G) offsets ((g*9+k)*2 + {dy,dx}), [18*G, 27*G) masks
 * (g*9+k); `mask_sigmoid` applies the sigmoid to the mask channels on load.  weight: [cout][cin][9]
 * in the tensor dtype, bias f32[cout] or NULL, act 0 / 3 (ReLU).  Out-of-image bilinear corners
 * contribute zero.  The reference only names this op (README.md:14-16, an absent CUDA build). */
int elvis_dcnv2(const void* x, const void* offset_mask, const void* weight, const float* bias, void* out,
                int dtype, int n, int h, int w, int cin, int x_pitch, int deformable_groups, int om_pitch,
                int mask_sigmoid, int cout, int out_pitch, int act, elvis_stream_t stream);

/* frames u8 [nf,h,w,3] -> planes [(f*3+c), h, w, pitch] for f in [f0, f0+nsel): channel t holds
 * colour c of frame clamp(f + t - radius) / 255 (the temporal window of the DCN restorer). */
int elvis_temporal_stack(const uint8_t* frames, void* out, int dtype, int nf, int f0, int nsel, int h, int w,
                         int radius, int pitch, elvis_stream_t stream);

/* out_u8[f,h,w,c] = round(clip(frames[f0+f][c]/255 + residual[(f*3+c),h,w,0], 0, 1) * 255). */
int elvis_plane_merge(const uint8_t* frames, const void* residual, uint8_t* out, int dtype, int f0, int nsel,
                      int h, int w, int pitch, elvis_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ELVIS_AMD_H */
