/*
 * jxl_oracle.h -- CPU restatement (the ORACLE) of jxlatte's per-frame transform stage.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product path (jxlatte_amd/, the C-ABI
 * library) links, imports or calls this; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (pure Java, no JVM available here or on the GPU box)
 * ships no tests, golden vectors or fixtures for this path, and
 * cannot be built or run here. This restatement follows the reference line by line
 * (loop order, association, Java int/float semantics; file:line cited per function)
 * and is anchored by analytic known-answer tests and an independent scipy cross-check
 * (tests/test_oracle_*.py), not by outputs of the reference itself.
 *
 * Build: plain C11, gcc -O2 -ffp-contract=off (no FMA contraction, SSE2 f32 arithmetic,
 * no fast-math); integer arithmetic that may overflow is done in uint32_t (Java wraps).
 */
#ifndef JXL_ORACLE_H
#define JXL_ORACLE_H
#include <stdint.h>
#include "../include/jxlatte_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* cosineLut[l][n][k] of MathHelper.java:17-30, flattened: returns pointer to the
 * (s-1) x s table for s = 1<<l, l = 0..8 */
const float* orc_cosine_lut(int l);

/* MathHelper.inverseDCTHorizontal / forwardDCTHorizontal (MathHelper.java:68-94) */
void orc_idct1d(const float* src, float* dst, int log_len, int len);
void orc_fdct1d(const float* src, float* dst, int log_len, int len);
/* MathHelper.inverseDCT2D (:96-122) / forwardDCT2D (:124-136) on strided planes */
void orc_idct2d(const float* src, int64_t sstride, float* dst, int64_t dstride, int h, int w, int transposed);
void orc_fdct2d(const float* src, int64_t sstride, float* dst, int64_t dstride, int h, int w);

/* Whole VarDCT frame: the sequence Frame.decodePassGroups tail -> Gab -> EPF -> invertXYB
 * -> transfer/quantise, honouring p->stages. Inputs are the boundary's own structures. */
typedef struct orc_vardct_frame {
    jxl_vardct_params p;
    const float* weights;      /* as jxl_vardct_set_weights */
    const int32_t* woffs;      /* [51] */
    int32_t n_lfg;             /* LF groups, raster order */
    const jxl_lfgroup_desc* lfg;
    const int32_t* coeff[3];   /* frame-level planes [height][width]: pass-summed quantizedCoeffs */
    int32_t threads;           /* >1: groups / rows spread over that many threads (same results) */
} orc_vardct_frame;
/* out[c]: float planes (JXL_OUT_F32) or int32 planes holding the quantised ints */
jxl_status orc_vardct_frame_run(const orc_vardct_frame* f, void* const out[3]);

/* stage functions, same argument meaning as the jxl_stage_* entries */
void orc_gab(const float* const in[3], float* const out[3], int h, int w, const float w1[3], const float w2[3]);
jxl_status orc_epf_sigma(const int32_t* hf_mul, const int32_t* sharpness, int bh, int bw, float global_scale_f,
                         const float sharp_lut[8], float* inv_sigma);
void orc_epf(const float* const in[3], float* const out[3], int h, int w, int iterations,
             const float* inv_sigma, float inv_sigma_modular, const float channel_scale[3],
             float pass0, float pass2, float border_sad_mul);
void orc_xyb(float* const planes[3], int64_t n, const float matrix[9], const float opsin_bias[3],
             const float cbrt_opsin_bias[3], float intensity_target);
void orc_ycbcr(float* const planes[3], int64_t n);
/* LFCoefficients.java:65-180 (dequant, LF chroma-from-luma, adaptiveSmooth) for one LF group */
void orc_lf_dequant(const jxl_lfquant_desc* d, float base_corr_x, float base_corr_b, int32_t color_factor, float* const out[3]);
void orc_transfer(const float* in, int64_t n, int transfer, int max_value, float* out_f, int32_t* out_i);
void orc_inv_hsqueeze(const int32_t* avg, int aw, const int32_t* res, int rw, int h, int32_t* out);
void orc_inv_vsqueeze(const int32_t* avg, int ah, const int32_t* res, int rh, int w, int32_t* out);
jxl_status orc_rct(int32_t* const v[3], int64_t n, int rct_type);
void orc_modular_to_float(const int32_t* a, const int32_t* b, int64_t n, float scale, float* out);

int32_t orc_default_squeeze_params(const int32_t* widths, const int32_t* heights, int32_t n_channels,
                                   int32_t nb_meta, jxl_squeeze_param* out, int32_t cap);
int32_t orc_squeezed_shapes(const int32_t* widths, const int32_t* heights, int32_t n_channels,
                            const jxl_squeeze_param* sp, int32_t n_sp, int32_t* out_w, int32_t* out_h, int32_t cap);
/* ModularStream.applyTransforms squeeze + RCT branches (ModularStream.java:224-326) */
jxl_status orc_modular_apply(const jxl_channel* chans, int32_t n_chans, const jxl_squeeze_param* sp, int32_t n_sp,
                             int32_t rct_type, int32_t rct_begin, jxl_channel* out, int32_t n_out);

/* rows f4 / f3 (jxl_oracle_post.c): same argument meaning as the jxl_stage_* entries of the same name */
void orc_chroma_upsample(const float* in, int h, int w, int x_shift, int y_shift, float* out);
jxl_status orc_upsampling_weights(int k, const float* packed, float* out);
void orc_upsample(const float* in, int h, int w, int k, const float* weights, float* out);
void orc_noise_init(int h, int w, int group_dim, uint64_t seed0, int colors, float* const out[3]);
void orc_noise_add(float* const planes[3], const float* const noise[3], int64_t n, const float lut[8], float base_corr_x,
                   float base_corr_b);
jxl_status orc_blend(int mode, uint32_t flags, int is_int, void* canvas, int ch, int cw, const void* frame, int fh, int fw,
                     const void* ref, int rh, int rw, const float* frame_alpha, const float* ref_alpha, const jxl_blend_rect* r);
jxl_status orc_orient(const void* in, int h, int w, int orientation, void* out);
jxl_status orc_pack(const void* const planes[4], const jxl_pack_params* p, void* out);

/* TEST-ONLY forward squeeze steps (the reference has no encoder; derived from the inverse,
 * SURVEY.md Appendix A.11): split in (h x w) into avg and res. */
void orc_fwd_hsqueeze(const int32_t* in, int h, int w, int32_t* avg, int32_t* res);
void orc_fwd_vsqueeze(const int32_t* in, int h, int w, int32_t* avg, int32_t* res);

#ifdef __cplusplus
}
#endif
#endif
