/*
 * jxlatte_frontend.h -- C API of the host-side JPEG XL front-end.
 *
 * The reference keeps bitstream parsing on the Java host (container demux, headers, ANS / prefix / LZ77 entropy
 * decoding, MA trees and predictors, TOC, LfGlobal / LfGroup / HfGlobal / pass groups). No JVM exists in this image,
 * so the same job is done by this C++ library: it turns a .jxl file into exactly the tensors that
 * include/jxlatte_amd.h takes (quantised coefficients per pass and group, LF images, varblock maps, modular channel
 * lists + transform descriptors). It contains NO transform-stage arithmetic for frame-level data: inverse Squeeze / RCT
 * of the frame's modular stream are delegated to the caller through jxf_hooks (the device library), and the VarDCT
 * pipeline is not present here at all. Plain CPU code (g++), no GPU needed.
 */
#ifndef JXLATTE_FRONTEND_H
#define JXLATTE_FRONTEND_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct jxf_dec jxf_dec;

#define JXF_OK 0
#define JXF_END 1              /* no more frames */
#define JXF_ERR_BITSTREAM -2   /* InvalidBitstreamException */
#define JXF_ERR_UNSUPPORTED -3 /* UnsupportedOperationException */
#define JXF_ERR_ARGUMENT -1
#define JXF_ERR_STATE -6

#define JXF_MAX_EXTRA 16

typedef struct jxf_image_info { /* J/bundle/ImageHeader.java */
    int32_t width, height, level, orientation;
    int32_t bits_per_sample, exp_bits, modular_16bit;
    int32_t num_extra, xyb_encoded;
    int32_t colour_space, white_point, primaries, transfer, rendering_intent, use_icc;
    float white_xy[2], prim_xy[6];
    float intensity_target, min_nits, linear_below;
    int32_t relative_to_max_display;
    float opsin_matrix[9], opsin_bias[3], quant_bias[3], quant_bias_numerator;
    int32_t have_animation, have_preview;
    int32_t custom_up[3];
    int32_t ec_type[JXF_MAX_EXTRA], ec_bits[JXF_MAX_EXTRA], ec_exp_bits[JXF_MAX_EXTRA], ec_dim_shift[JXF_MAX_EXTRA],
        ec_alpha_associated[JXF_MAX_EXTRA];
} jxf_image_info;

typedef struct jxf_frame_info { /* J/frame/FrameHeader.java + LFGlobal.java + HFGlobal.numHfPresets */
    int32_t type, encoding, do_ycbcr, upsampling, group_dim, xqm, bqm, lf_level;
    uint64_t flags;
    int32_t jpeg_up_y[3], jpeg_up_x[3];
    int32_t ec_upsampling[JXF_MAX_EXTRA];
    int32_t num_passes, pass_shift[11];
    int32_t x0, y0, width, height, padded_width, padded_height;
    int32_t blend_mode, blend_alpha, blend_clamp, blend_source;
    int32_t ec_blend_mode[JXF_MAX_EXTRA], ec_blend_alpha[JXF_MAX_EXTRA], ec_blend_clamp[JXF_MAX_EXTRA],
        ec_blend_source[JXF_MAX_EXTRA];
    uint32_t duration;
    int32_t is_last, save_as_reference, save_before_ct;
    /* RestorationFilter */
    int32_t gab, epf_iters;
    float gab1[3], gab2[3], epf_sharp_lut[8], epf_channel_scale[3];
    float epf_pass0_sigma, epf_pass2_sigma, epf_border_sad_mul, epf_sigma_modular;
    /* geometry */
    int32_t num_groups, num_lf_groups, group_cols, lf_group_cols;
    /* LfGlobal */
    int32_t num_patches, has_splines, has_noise;
    float noise[8];
    float lf_dequant[3], scaled_dequant[3];
    int32_t global_scale, quant_lf;
    int32_t colour_factor, x_factor_lf, b_factor_lf;
    float base_corr_x, base_corr_b;
    /* HfGlobal */
    int32_t quant_all_default, num_hf_presets;
    /* modular */
    int32_t num_modular_channels;
} jxf_frame_info;

typedef struct jxf_chan {
    int32_t w, h, hshift, vshift;
    int32_t* data; /* h * w, row-major */
} jxf_chan;

typedef struct jxf_squeeze_step { /* same layout as jxl_squeeze_param */
    int32_t horizontal, in_place, begin_c, num_c;
} jxf_squeeze_step;

/* Hooks through which the frame-level modular transforms reach the device library. Return 0 on success.
 * squeeze: in[n_in] = the stream's channel list before the inverse, steps in bitstream order (undone last to first),
 *          out[n_out] = pre-allocated result channels (the list after the inverse). Bind to jxl_modular_apply.
 * rct:     three planes of n samples, updated in place and left in OUTPUT channel order (permutation applied), as
 *          jxl_stage_rct does. */
typedef struct jxf_hooks {
    void* user;
    int32_t (*squeeze)(void* user, const jxf_chan* in, int32_t n_in, const jxf_squeeze_step* steps, int32_t n_steps,
                       jxf_chan* out, int32_t n_out);
    int32_t (*rct)(void* user, int32_t* v0, int32_t* v1, int32_t* v2, int64_t n, int32_t rct_type);
} jxf_hooks;

typedef struct jxf_lfgroup_view { /* one LF group: J/frame/group/LFGroup.java, LFCoefficients (integers), HFMetadata */
    int32_t cells_h, cells_w;
    int32_t extra_precision, has_lf_quant;
    const int32_t* lf_quant[3]; /* X, Y, B buffer order (lfQuant[cMap[i]]) */
    int32_t lf_h[3], lf_w[3];
    int32_t n_blocks;
    const uint8_t* dct_select; /* [cells_h][cells_w] */
    const int32_t* hf_mul;
    const int32_t* sharpness;
    const int32_t* x_from_y; /* [ceil(cells_h/8)][ceil(cells_w/8)] */
    const int32_t* b_from_y;
    const int32_t* block_yx; /* n_blocks x (y, x) */
} jxf_lfgroup_view;

typedef struct jxf_coeff_view { /* HFCoefficients.quantizedCoeffs of one (pass, group), X, Y, B order */
    const int32_t* q[3];
    int32_t h[3], w[3];
} jxf_coeff_view;

typedef struct jxf_quant_view { /* one DCTParams set (HFGlobal.java); arrays are [3][n] flattened */
    int32_t mode;
    float denominator;
    int32_t n_dct, n_par, n_p44;
    const float* dct;
    const float* par;
    const float* p44;
} jxf_quant_view;

typedef struct jxf_patch_view { /* J/frame/features/Patch.java */
    int32_t ref, x0, y0, w, h, n_positions, n_blend; /* n_blend = 1 + extra channels */
    const int32_t* positions;                        /* n_positions x (y, x) */
    const int32_t* blend;                            /* n_positions x n_blend x (mode, alpha, clamp) */
} jxf_patch_view;

typedef struct jxf_spline_view { /* one spline of J/frame/features/spline/SplinesBundle.java */
    int32_t quant_adjust;      /* SplinesBundle.quantAdjust (frame-wide) */
    int32_t n_control;
    const int32_t* control;    /* n_control x (y, x) */
    const int32_t* coeff;      /* [4][32]: X, Y, B, sigma */
} jxf_spline_view;

/* Parses the container (if any) and the image header. data is copied. Returns NULL and fills err on failure. */
jxf_dec* jxf_open(const uint8_t* data, size_t size, char* err, size_t err_len);
void jxf_close(jxf_dec* d);
const char* jxf_last_error(const jxf_dec* d);
int32_t jxf_get_image_info(const jxf_dec* d, jxf_image_info* out);
/* custom upsampling weights of the image header (15 / 55 / 210 floats) when custom_up[k] is set; count returned */
int32_t jxf_get_up_weights(const jxf_dec* d, int32_t k_index, float* out, int32_t cap);

/* Decodes the next frame's bitstream (all sections). JXF_OK, JXF_END, or a negative status. hooks may be NULL: the
 * frame-level modular transforms then FAIL (JXF_ERR_STATE) if the frame needs an inverse Squeeze or RCT -- there is no
 * CPU fallback for the frame-level stream. */
int32_t jxf_next_frame(jxf_dec* d, const jxf_hooks* hooks);
int32_t jxf_get_frame_info(const jxf_dec* d, jxf_frame_info* out);
int32_t jxf_get_lfgroup(const jxf_dec* d, int32_t index, jxf_lfgroup_view* out);
int32_t jxf_get_coeffs(const jxf_dec* d, int32_t pass, int32_t group, jxf_coeff_view* out);
int32_t jxf_get_quant_params(const jxf_dec* d, int32_t index, jxf_quant_view* out);
int32_t jxf_get_patch(const jxf_dec* d, int32_t index, jxf_patch_view* out);
/* number of splines of the current frame, and one of them */
int32_t jxf_num_splines(const jxf_dec* d);
int32_t jxf_get_spline(const jxf_dec* d, int32_t index, jxf_spline_view* out);
/* channel i of the frame-level modular stream after its inverse transforms */
int32_t jxf_get_modular_channel(const jxf_dec* d, int32_t index, jxf_chan* out);

#ifdef __cplusplus
}
#endif
#endif
