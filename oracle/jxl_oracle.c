/*
 * jxl_oracle.c -- CPU restatement (ORACLE) of jxlatte's per-frame transform stage.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED by reference fixtures (see jxl_oracle.h).
 *
 * J/ = java/com/traneptora/jxlatte/ in the reference tree. Every function cites the
 * reference lines it follows. All float sums are f32, left to right, multiply and add
 * rounded separately (-ffp-contract=off); ints wrap like Java.
 */
#include "jxl_oracle.h"
#include "../include/jxl_transform_types.h"
#include "../include/jxl_tables.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- Java semantics helpers ------------------------------------------------------ */
static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
/* Java (int)float: NaN -> 0, saturating */
static inline int32_t java_f2i(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
/* MathHelper.ceilLog2 (MathHelper.java:154-160) for x >= 1 */
static inline int ceil_log2(int x) {
    int l = 0;
    while ((1 << l) < x) l++;
    return l;
}
/* MathHelper.mirrorCoordinate (MathHelper.java:323-329) */
static inline int mirror(int c, int size) {
    while (c < 0 || c >= size) {
        int tc = ~c;
        c = tc >= 0 ? tc : (size << 1) + tc;
    }
    return c;
}

/* ---- cosine LUT (MathHelper.java:17-30) -------------------------------------------- */
static float* g_lut[9];
static volatile int g_lut_ready = 0;

static void lut_init(void) {
    if (g_lut_ready) return;
#pragma omp critical(orc_lut)
    {
        if (!g_lut_ready) {
            const double root2 = sqrt(2.0);
            for (int l = 0; l < 9; l++) {
                int s = 1 << l;
                float* t = (float*)malloc(sizeof(float) * (size_t)(s > 1 ? (s - 1) * s : 1));
                for (int n = 0; n < s - 1; n++)
                    for (int k = 0; k < s; k++)
                        t[n * s + k] = (float)(root2 * cos(M_PI * (n + 1) * (k + 0.5) / s));
                g_lut[l] = t;
            }
            g_lut_ready = 1;
        }
    }
}

const float* orc_cosine_lut(int l) {
    lut_init();
    return (l >= 0 && l < 9) ? g_lut[l] : NULL;
}

/* MathHelper.inverseDCTHorizontal (MathHelper.java:68-78) */
void orc_idct1d(const float* src, float* dst, int log_len, int len) {
    lut_init();
    const float s0 = src[0];
    for (int k = 0; k < len; k++) dst[k] = s0;
    const float* lutX = g_lut[log_len];
    for (int n = 1; n < len; n++) {
        const float* lut = lutX + (size_t)(n - 1) * len;
        const float s2 = src[n];
        for (int k = 0; k < len; k++) dst[k] += s2 * lut[k];
    }
}

/* MathHelper.forwardDCTHorizontal (MathHelper.java:80-94) */
void orc_fdct1d(const float* src, float* dst, int log_len, int len) {
    lut_init();
    const float inv = 1.0f / (float)len;
    float d2 = src[0];
    for (int x = 1; x < len; ++x) d2 += src[x];
    dst[0] = d2 * inv;
    for (int k = 1; k < len; ++k) {
        const float* lut = g_lut[log_len] + (size_t)(k - 1) * len;
        d2 = src[0] * lut[0];
        for (int n = 1; n < len; ++n) d2 += src[n] * lut[n];
        dst[k] = d2 * inv;
    }
}

#define SCR 256 /* scratch plane row stride: float[256][256] as PassGroup.java:208 */

/* MathHelper.transposeMatrixInto (MathHelper.java:138-145) */
static void transpose_into(const float* src, int64_t ss, float* dst, int64_t ds, int sh, int sw) {
    for (int y = 0; y < sh; y++)
        for (int x = 0; x < sw; x++) dst[x * ds + y] = src[y * ss + x];
}

/* MathHelper.inverseDCT2D (MathHelper.java:96-122); s0,s1 = scratchSpace0/1 (stride SCR) */
static void idct2d_s(const float* src, int64_t ss, float* dst, int64_t ds, int h, int w, int transposed,
                     float* s0, float* s1) {
    int lh = ceil_log2(h), lw = ceil_log2(w);
    if (transposed) {
        for (int y = 0; y < h; y++) orc_idct1d(src + y * ss, s1 + y * SCR, lw, w);
        transpose_into(s1, SCR, s0, SCR, h, w);
        for (int y = 0; y < w; y++) orc_idct1d(s0 + y * SCR, dst + y * ds, lh, h);
    } else {
        transpose_into(src, ss, s0, SCR, h, w);
        for (int y = 0; y < w; y++) orc_idct1d(s0 + y * SCR, s1 + y * SCR, lh, h);
        transpose_into(s1, SCR, s0, SCR, w, h);
        for (int y = 0; y < h; y++) orc_idct1d(s0 + y * SCR, dst + y * ds, lw, w);
    }
}

/* MathHelper.forwardDCT2D (MathHelper.java:124-136) */
static void fdct2d_s(const float* src, int64_t ss, float* dst, int64_t ds, int h, int w, float* s0, float* s1) {
    int lh = ceil_log2(h), lw = ceil_log2(w);
    for (int y = 0; y < h; y++) orc_fdct1d(src + y * ss, s0 + y * SCR, lw, w);
    transpose_into(s0, SCR, s1, SCR, h, w);
    for (int x = 0; x < w; x++) orc_fdct1d(s1 + x * SCR, s0 + x * SCR, lh, h);
    transpose_into(s0, SCR, dst, ds, w, h);
}

void orc_idct2d(const float* src, int64_t sstride, float* dst, int64_t dstride, int h, int w, int transposed) {
    float* s = (float*)malloc(sizeof(float) * 2 * SCR * SCR);
    idct2d_s(src, sstride, dst, dstride, h, w, transposed, s, s + SCR * SCR);
    free(s);
}

void orc_fdct2d(const float* src, int64_t sstride, float* dst, int64_t dstride, int h, int w) {
    float* s = (float*)malloc(sizeof(float) * 2 * SCR * SCR);
    fdct2d_s(src, sstride, dst, dstride, h, w, s, s + SCR * SCR);
    free(s);
}

/* ---- PassGroup special transforms --------------------------------------------------- */
static const float AFV_BASIS[16][16] = JXL_AFV_BASIS_INIT; /* PassGroup.java:19-58 */
static const float LLF_SCALE[32] = JXL_LLF_SCALE_INIT;     /* LLFScale.java:7-24 */

/* TransformType ctor llfScale (TransformType.java:158-165) + LLFScale.scaleF (:21-23) */
static inline float llf_scale(int y, int x, int dsh, int dsw) {
    int yll = ceil_log2(dsh), xll = ceil_log2(dsw);
    return LLF_SCALE[y << (5 - yll)] * LLF_SCALE[x << (5 - xll)];
}

/* PassGroup.auxDCT2 (PassGroup.java:149-168): in @ (stride is), out @ (stride os) */
static void aux_dct2(const float* in, int64_t is, float* out, int64_t os, int s) {
    for (int y = 0; y < 8; y++) /* layBlock 8x8 (:150) */
        for (int x = 0; x < 8; x++) out[y * os + x] = in[y * is + x];
    int num = s / 2;
    for (int iy = 0; iy < num; iy++) {
        for (int ix = 0; ix < num; ix++) {
            float c00 = in[iy * is + ix];
            float c01 = in[iy * is + ix + num];
            float c10 = in[(iy + num) * is + ix];
            float c11 = in[(iy + num) * is + ix + num];
            float r00 = c00 + c01 + c10 + c11;
            float r01 = c00 + c01 - c10 - c11;
            float r10 = c00 - c01 + c10 - c11;
            float r11 = c00 - c01 - c10 + c11;
            out[(iy * 2) * os + ix * 2] = r00;
            out[(iy * 2) * os + ix * 2 + 1] = r01;
            out[(iy * 2 + 1) * os + ix * 2] = r10;
            out[(iy * 2 + 1) * os + ix * 2 + 1] = r11;
        }
    }
}

/* PassGroup.invertAFV (PassGroup.java:88-147). co = coefficients @ppg (stride cs),
 * fb = frame buffer @ppf (stride fs); sb = 4 scratch planes of stride SCR. */
static void invert_afv(const float* co, int64_t cs, float* fb, int64_t fs, int type, float* sb[5]) {
    float* s0 = sb[0];
    float* s1 = sb[1];
    s0[0] = (co[0] + co[cs] + co[1]) * 4.0f;
    for (int iy = 0; iy < 4; iy++)
        for (int ix = (iy == 0 ? 1 : 0); ix < 4; ix++) s0[iy * SCR + ix] = co[(iy * 2) * cs + ix * 2];
    int flipY = (type == 16 || type == 17) ? 1 : 0; /* AFV2, AFV3 */
    int flipX = (type == 15 || type == 17) ? 1 : 0; /* AFV1, AFV3 */
    for (int iy = 0; iy < 4; iy++) {
        for (int ix = 0; ix < 4; ix++) {
            float sample = 0.0f;
            for (int j = 0; j < 16; j++) {
                int jy = j >> 2, jx = j & 3;
                sample += s0[jy * SCR + jx] * AFV_BASIS[j][iy * 4 + ix];
            }
            s1[iy * SCR + ix] = sample;
        }
    }
    for (int iy = 0; iy < 4; iy++)
        for (int ix = 0; ix < 4; ix++)
            fb[(flipY * 4 + iy) * fs + flipX * 4 + ix] = s1[(flipY == 1 ? 3 - iy : iy) * SCR + (flipX == 1 ? 3 - ix : ix)];
    s0[0] = co[0] + co[cs] - co[1];
    for (int iy = 0; iy < 4; iy++)
        for (int ix = (iy == 0 ? 1 : 0); ix < 4; ix++) s0[iy * SCR + ix] = co[(iy * 2) * cs + ix * 2 + 1];
    idct2d_s(s0, SCR, s1, SCR, 4, 4, 0, sb[2], sb[3]);
    for (int iy = 0; iy < 4; iy++)
        for (int ix = 0; ix < 4; ix++) /* transposed intentionally (:129-131) */
            fb[(flipY * 4 + iy) * fs + (flipX == 1 ? 0 : 4) + ix] = s1[ix * SCR + iy];
    s0[0] = co[0] - co[cs];
    for (int iy = 0; iy < 4; iy++)
        for (int ix = (iy == 0 ? 1 : 0); ix < 8; ix++) s0[iy * SCR + ix] = co[(1 + iy * 2) * cs + ix];
    idct2d_s(s0, SCR, s1, SCR, 4, 8, 0, sb[2], sb[3]);
    for (int iy = 0; iy < 4; iy++)
        for (int ix = 0; ix < 8; ix++) fb[((flipY == 1 ? 0 : 4) + iy) * fs + ix] = s1[iy * SCR + ix];
}

/* the per-varblock, per-channel switch of PassGroup.invertVarDCT (PassGroup.java:229-328) */
static jxl_status invert_block(const float* co, int64_t cs, float* fb, int64_t fs, const jxl_tt_info* tt, float* sb[5]) {
    float coeff0, coeff1, lfs[2];
    switch (tt->method) {
    case JXL_METHOD_DCT:
        idct2d_s(co, cs, fb, fs, tt->ph, tt->pw, 0, sb[0], sb[1]);
        break;
    case JXL_METHOD_DCT8_4: /* :234-251 */
        coeff0 = co[0];
        coeff1 = co[cs];
        lfs[0] = coeff0 + coeff1;
        lfs[1] = coeff0 - coeff1;
        for (int x = 0; x < 2; x++) {
            sb[0][0] = lfs[x];
            for (int iy = 0; iy < 4; iy++)
                for (int ix = (iy == 0 ? 1 : 0); ix < 8; ix++) sb[0][iy * SCR + ix] = co[(x + iy * 2) * cs + ix];
            idct2d_s(sb[0], SCR, fb + (x << 2), fs, 4, 8, 1, sb[1], sb[2]);
        }
        break;
    case JXL_METHOD_DCT4_8: /* :252-269 */
        coeff0 = co[0];
        coeff1 = co[cs];
        lfs[0] = coeff0 + coeff1;
        lfs[1] = coeff0 - coeff1;
        for (int y = 0; y < 2; y++) {
            sb[0][0] = lfs[y];
            for (int iy = 0; iy < 4; iy++)
                for (int ix = (iy == 0 ? 1 : 0); ix < 8; ix++) sb[0][iy * SCR + ix] = co[(y + iy * 2) * cs + ix];
            idct2d_s(sb[0], SCR, fb + (int64_t)(y << 2) * fs, fs, 4, 8, 0, sb[1], sb[2]);
        }
        break;
    case JXL_METHOD_AFV:
        invert_afv(co, cs, fb, fs, tt->type, sb);
        break;
    case JXL_METHOD_DCT2: /* :273-277 */
        aux_dct2(co, cs, sb[0], SCR, 2);
        aux_dct2(sb[0], SCR, sb[1], SCR, 4);
        aux_dct2(sb[1], SCR, fb, fs, 8);
        break;
    case JXL_METHOD_HORNUSS: /* :278-305 */
        aux_dct2(co, cs, sb[1], SCR, 2);
        for (int y = 0; y < 2; y++) {
            for (int x = 0; x < 2; x++) {
                float blockLF = sb[1][y * SCR + x];
                float residual = 0.0f;
                for (int iy = 0; iy < 4; iy++)
                    for (int ix = (iy == 0 ? 1 : 0); ix < 4; ix++) residual += co[(y + iy * 2) * cs + x + ix * 2];
                sb[0][(4 * y + 1) * SCR + 4 * x + 1] = blockLF - residual * 0.0625f;
                for (int iy = 0; iy < 4; iy++) {
                    for (int ix = 0; ix < 4; ix++) {
                        if (ix == 1 && iy == 1) continue;
                        sb[0][(y * 4 + iy) * SCR + x * 4 + ix] =
                            co[(y + iy * 2) * cs + x + ix * 2] + sb[0][(4 * y + 1) * SCR + 4 * x + 1];
                    }
                }
                sb[0][(4 * y) * SCR + 4 * x] = co[(y + 2) * cs + x + 2] + sb[0][(4 * y + 1) * SCR + 4 * x + 1];
            }
        }
        for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) fb[y * fs + x] = sb[0][y * SCR + x];
        break;
    case JXL_METHOD_DCT4: /* :306-325 */
        aux_dct2(co, cs, sb[4], SCR, 2);
        for (int y = 0; y < 2; y++) {
            for (int x = 0; x < 2; x++) {
                sb[0][0] = sb[4][y * SCR + x];
                for (int iy = 0; iy < 4; iy++)
                    for (int ix = (iy == 0 ? 1 : 0); ix < 4; ix++) sb[0][iy * SCR + ix] = co[(y + iy * 2) * cs + x + ix * 2];
                idct2d_s(sb[0], SCR, sb[1], SCR, 4, 4, 1, sb[2], sb[3]);
                for (int iy = 0; iy < 4; iy++)
                    for (int ix = 0; ix < 4; ix++) fb[(4 * y + iy) * fs + 4 * x + ix] = sb[1][iy * SCR + ix];
            }
        }
        break;
    default:
        return JXL_ERR_UNSUPPORTED; /* PassGroup.java:326-327 */
    }
    return JXL_OK;
}

/* ---- one group: HFCoefficients.bakeDequantizedCoeffs + PassGroup.invertVarDCT -------- */
static jxl_status vardct_group(const orc_vardct_frame* f, int group, float* const out[3]) {
    const jxl_vardct_params* p = &f->p;
    const int W = p->width;
    const int grs = ceil_div(W, 256);  /* groupRowStride */
    const int lrs = ceil_div(W, 2048); /* lfGroupRowStride */
    const int gy = group / grs, gx = group % grs; /* Frame.getGroupLocation (:883) */
    const jxl_lfgroup_desc* lfg = &f->lfg[(gy >> 3) * lrs + (gx >> 3)]; /* getLFGroupForGroup (:849) */
    /* groupPosInLFGroup (:897) << 5: position in cells inside the LF group */
    const int gposy = (gy - (lfg->lfg_y << 3)) << 5, gposx = (gx - (lfg->lfg_x << 3)) << 5;
    const int cw = lfg->cells_w;
    /* per-channel geometry of chroma-subsampled frames (FrameHeader.jpegUpsamplingY/X; PassGroup.java:213-226) */
    int sy[3], sx[3], Wc[3], lfw[3], subsampled = 0;
    int64_t goffc[3];
    for (int c = 0; c < 3; c++) {
        sy[c] = p->jpeg_upsampling_y[c];
        sx[c] = p->jpeg_upsampling_x[c];
        subsampled |= sy[c] | sx[c];
        Wc[c] = W >> sx[c];
        lfw[c] = cw >> sx[c];
        goffc[c] = (int64_t)((gy << 8) >> sy[c]) * Wc[c] + ((gx << 8) >> sx[c]); /* groupLocation << 8 >> upsampling */
    }
    jxl_status st = JXL_OK;

    float* dq[3];
    float* mem = (float*)calloc((size_t)(3 + 5) * SCR * SCR + 2 * 32 * SCR, sizeof(float));
    if (!mem) return JXL_ERR_OOM;
    for (int c = 0; c < 3; c++) dq[c] = mem + (size_t)c * SCR * SCR; /* dequantHFCoeff[c], zero-initialised */
    float* sb[5];
    for (int i = 0; i < 5; i++) sb[i] = mem + (size_t)(3 + i) * SCR * SCR; /* scratchBlock (PassGroup.java:208) */
    float* ls0 = mem + (size_t)8 * SCR * SCR; /* finalizeLLF scratchBlock[2][32][32] (:195) */
    float* ls1 = ls0 + 32 * SCR;

    /* blocks[] filter of the HFCoefficients ctor (HFCoefficients.java:76-85) */
    int nb = lfg->n_blocks;
    uint8_t* inb = (uint8_t*)malloc((size_t)(nb > 0 ? nb : 1));
    for (int i = 0; i < nb; i++) {
        int groupY = lfg->block_yx[2 * i] - gposy, groupX = lfg->block_yx[2 * i + 1] - gposx;
        inb[i] = !(groupY < 0 || groupX < 0 || groupY >= 32 || groupX >= 32);
    }

    /* dequantizeHFCoefficients (HFCoefficients.java:267-319) */
    for (int i = 0; i < nb; i++) {
        if (!inb[i]) continue;
        int posy = lfg->block_yx[2 * i], posx = lfg->block_yx[2 * i + 1];
        int type = lfg->dct_select[posy * cw + posx];
        if (type > 26) { st = JXL_ERR_INVALID_BITSTREAM; goto done; } /* HFMetadata.java:46-47 */
        const jxl_tt_info* tt = &JXL_TT[type];
        int groupY = posy - gposy, groupX = posx - gposx;
        int flip = jxl_tt_flip(tt);
        int mw = jxl_tt_mw(tt);
        int dsh = tt->ph >> 3, dsw = tt->pw >> 3;
        for (int c = 0; c < 3; c++) {
            int sGroupY = groupY >> sy[c], sGroupX = groupX >> sx[c];
            if (groupY != sGroupY << sy[c] || groupX != sGroupX << sx[c]) continue; /* subsampled block (:292-297) */
            const float* w3 = f->weights + f->woffs[tt->param_index * 3 + c];
            float sfc = p->scale_factor[c] / (float)lfg->hf_mul[posy * cw + posx];
            int pgy = sGroupY << 3, pgx = sGroupX << 3;
            float qbc[3] = {-p->quant_bias[c], 0.0f, p->quant_bias[c]};
            for (int y = 0; y < tt->ph; y++) {
                for (int x = 0; x < tt->pw; x++) {
                    if (y < dsh && x < dsw) continue;
                    int pY = pgy + y, pX = pgx + x;
                    int32_t coeff = f->coeff[c][goffc[c] + (int64_t)pY * Wc[c] + pX];
                    float quant = (coeff > -2 && coeff < 2) ? qbc[coeff + 1]
                                                            : (float)coeff - p->quant_bias_numerator / (float)coeff;
                    int wy = flip ? x : y;
                    int wx = x ^ y ^ wy;
                    dq[c][pY * SCR + pX] = quant * sfc * w3[wy * mw + wx];
                }
            }
        }
    }

    /* chromaFromLuma (HFCoefficients.java:146-192): skipped for chroma-subsampled frames (:149-151) */
    if (!subsampled) {
        int th = ceil_div(lfg->cells_h, 8), tw = ceil_div(lfg->cells_w, 8);
        float* xF = (float*)calloc((size_t)th * tw * 2, sizeof(float)); /* xFactors / bFactors, per call */
        float* bF = xF + (size_t)th * tw;
        for (int i = 0; i < nb; i++) {
            if (!inb[i]) continue;
            int posy = lfg->block_yx[2 * i], posx = lfg->block_yx[2 * i + 1];
            const jxl_tt_info* tt = &JXL_TT[lfg->dct_select[posy * cw + posx]];
            int pPosY = posy << 3, pPosX = posx << 3;
            for (int iy = 0; iy < tt->ph; iy++) {
                int y = pPosY + iy;
                int fy = y >> 6;
                int by = (fy << 6) == y;
                for (int ix = 0; ix < tt->pw; ix++) {
                    int x = pPosX + ix;
                    int fx = x >> 6;
                    float kX, kB;
                    if (by && (fx << 6) == x) {
                        kX = p->base_corr_x + (float)lfg->x_from_y[fy * tw + fx] / (float)p->color_factor;
                        kB = p->base_corr_b + (float)lfg->b_from_y[fy * tw + fx] / (float)p->color_factor;
                        xF[fy * tw + fx] = kX;
                        bF[fy * tw + fx] = kB;
                    } else {
                        kX = xF[fy * tw + fx];
                        kB = bF[fy * tw + fx];
                    }
                    float dequantY = dq[1][(y & 0xFF) * SCR + (x & 0xFF)];
                    dq[0][(y & 0xFF) * SCR + (x & 0xFF)] += kX * dequantY;
                    dq[2][(y & 0xFF) * SCR + (x & 0xFF)] += kB * dequantY;
                }
            }
        }
        free(xF);
    }

    /* finalizeLLF (HFCoefficients.java:194-229) */
    for (int i = 0; i < nb; i++) {
        if (!inb[i]) continue;
        int posy = lfg->block_yx[2 * i], posx = lfg->block_yx[2 * i + 1];
        const jxl_tt_info* tt = &JXL_TT[lfg->dct_select[posy * cw + posx]];
        int groupY = posy - gposy, groupX = posx - gposx;
        int dsh = tt->ph >> 3, dsw = tt->pw >> 3;
        for (int c = 0; c < 3; c++) {
            int sGroupY = groupY >> sy[c], sGroupX = groupX >> sx[c];
            if (groupY != sGroupY << sy[c] || groupX != sGroupX << sx[c]) continue;
            int pgy = sGroupY << 3, pgx = sGroupX << 3;
            const float* dqlf = lfg->lf[c];
            int sLfgY = posy >> sy[c], sLfgX = posx >> sx[c]; /* :213-214 */
            fdct2d_s(dqlf + (int64_t)sLfgY * lfw[c] + sLfgX, lfw[c], dq[c] + pgy * SCR + pgx, SCR, dsh, dsw, ls0, ls1);
            for (int y = 0; y < dsh; y++)
                for (int x = 0; x < dsw; x++) dq[c][(y + pgy) * SCR + x + pgx] *= llf_scale(y, x, dsh, dsw);
        }
    }

    /* PassGroup.invertVarDCT block loop (PassGroup.java:209-329) */
    for (int i = 0; i < nb; i++) {
        if (!inb[i]) continue;
        int posy = lfg->block_yx[2 * i], posx = lfg->block_yx[2 * i + 1];
        const jxl_tt_info* tt = &JXL_TT[lfg->dct_select[posy * cw + posx]];
        int groupY = posy - gposy, groupX = posx - gposx;
        for (int c = 0; c < 3; c++) {
            int sGroupY = groupY >> sy[c], sGroupX = groupX >> sx[c];
            if (groupY != sGroupY << sy[c] || groupX != sGroupX << sx[c]) continue;
            int ppgy = sGroupY << 3, ppgx = sGroupX << 3;
            st = invert_block(dq[c] + ppgy * SCR + ppgx, SCR, out[c] + goffc[c] + (int64_t)ppgy * Wc[c] + ppgx, Wc[c], tt, sb);
            if (st != JXL_OK) goto done;
        }
    }
done:
    free(inb);
    free(mem);
    return st;
}

/* ---- Frame.performGabConvolution (Frame.java:505-542) -------------------------------- */
void orc_gab(const float* const in[3], float* const out[3], int h, int w, const float w1[3], const float w2[3]) {
    for (int c = 0; c < 3; c++) {
        float mult = 1.0f / (1.0f + 4.0f * (w1[c] + w2[c]));
        float base = mult, adjw = w1[c] * mult, diagw = w2[c] * mult;
        const float* b = in[c];
        float* o = out[c];
#pragma omp parallel for schedule(static)
        for (int y = 0; y < h; y++) {
            int north = (y == 0 ? 0 : y - 1);
            int south = (y + 1 == h) ? h - 1 : y + 1;
            const float* R = b + (int64_t)y * w;
            const float* N = b + (int64_t)north * w;
            const float* S = b + (int64_t)south * w;
            for (int x = 0; x < w; x++) {
                int west = (x == 0 ? 0 : x - 1);
                int east = (x + 1 == w ? w - 1 : x + 1);
                float adj = R[west] + R[east] + N[x] + S[x];
                float diag = N[west] + N[east] + S[west] + S[east];
                o[(int64_t)y * w + x] = base * R[x] + adjw * adj + diagw * diag;
            }
        }
    }
}

/* inverse sigma map of Frame.performEdgePreservingFilter (Frame.java:552-571) */
jxl_status orc_epf_sigma(const int32_t* hf_mul, const int32_t* sharpness, int bh, int bw, float global_scale_f,
                         const float sharp_lut[8], float* inv_sigma) {
    for (int y = 0; y < bh; y++) {
        for (int x = 0; x < bw; x++) {
            int hf = hf_mul[y * bw + x];
            int sharp = sharpness[y * bw + x];
            if (sharp < 0 || sharp > 7) return JXL_ERR_INVALID_BITSTREAM; /* :565-566 */
            float sigma = global_scale_f * sharp_lut[sharp] / (float)hf;
            inv_sigma[y * bw + x] = 1.0f / sigma;
        }
    }
    return JXL_OK;
}

static const int8_t EPF_CROSS[5][2] = {{0, 0}, {0, -1}, {0, 1}, {-1, 0}, {1, 0}}; /* Frame.java:44-48, (y,x) */
static const int8_t EPF_DCROSS[13][2] = {{0, 0}, {0, -1}, {0, 1}, {-1, 0}, {1, 0}, {-1, 1}, {1, 1},
                                         {1, -1}, {-1, -1}, {0, -2}, {0, 2}, {2, 0}, {-2, 0}}; /* :50-55 */

/* Frame.epfDistance1 (Frame.java:638-655) */
static inline float epf_dist1(const float* const b[3], int y, int x, int dy, int dx, int h, int w, const float scale[3]) {
    float dist = 0.0f;
    for (int c = 0; c < 3; c++) {
        const float* bc = b[c];
        for (int q = 0; q < 5; q++) {
            int pY = mirror(y + EPF_CROSS[q][0], h);
            int pX = mirror(x + EPF_CROSS[q][1], w);
            int dY = mirror(y + dy + EPF_CROSS[q][0], h);
            int dX = mirror(x + dx + EPF_CROSS[q][1], w);
            dist += fabsf(bc[(int64_t)pY * w + pX] - bc[(int64_t)dY * w + dX]) * scale[c];
        }
    }
    return dist;
}

/* Frame.epfDistance2 (Frame.java:657-669) */
static inline float epf_dist2(const float* const b[3], int y, int x, int dy, int dx, int h, int w, const float scale[3]) {
    float dist = 0.0f;
    for (int c = 0; c < 3; c++) {
        int dY = mirror(y + dy, h);
        int dX = mirror(x + dx, w);
        dist += fabsf(b[c][(int64_t)y * w + x] - b[c][(int64_t)dY * w + dX]) * scale[c];
    }
    return dist;
}

/* Frame.performEdgePreservingFilter iteration loop (Frame.java:583-635), 3 colour channels */
void orc_epf(const float* const in[3], float* const out[3], int h, int w, int iterations,
             const float* inv_sigma, float inv_sigma_modular, const float channel_scale[3],
             float pass0, float pass2, float border_sad_mul) {
    const float stepMultiplier = 1.65f * 4.0f * (1.0f - (float)sqrt(0.5)); /* :545, MathHelper.SQRT_H */
    const int bw = (w + 7) >> 3;
    size_t n = (size_t)h * w;
    float* A[3];
    float* B[3];
    for (int c = 0; c < 3; c++) {
        A[c] = (float*)malloc(n * sizeof(float));
        B[c] = (float*)calloc(n, sizeof(float));
        memcpy(A[c], in[c], n * sizeof(float));
    }
    for (int i = 0; i < 3; i++) {
        if (i == 0 && iterations < 3) continue;
        if (i == 2 && iterations < 2) break;
        if (iterations <= 0) break;
        float sigmaScale;
        if (i == 0) sigmaScale = stepMultiplier * pass0;
        else if (i == 2) sigmaScale = stepMultiplier * pass2;
        else sigmaScale = stepMultiplier;
        const int8_t(*cross)[2] = i == 0 ? EPF_DCROSS : EPF_CROSS;
        const int ncross = i == 0 ? 13 : 5;
        const float* const ib[3] = {A[0], A[1], A[2]};
#pragma omp parallel for schedule(static)
        for (int y = 0; y < h; y++) {
            for (int x = 0; x < w; x++) {
                float s = inv_sigma ? inv_sigma[(y >> 3) * bw + (x >> 3)] : inv_sigma_modular;
                if (s != s || s > (1.0f / 0.3f)) {
                    for (int c = 0; c < 3; c++) B[c][(int64_t)y * w + x] = ib[c][(int64_t)y * w + x];
                    continue;
                }
                float sumWeights = 0.0f;
                float sumChannels[3] = {0.0f, 0.0f, 0.0f};
                for (int t = 0; t < ncross; t++) {
                    int dy = cross[t][0], dx = cross[t][1];
                    float dist = i == 2 ? epf_dist2(ib, y, x, dy, dx, h, w, channel_scale)
                                        : epf_dist1(ib, y, x, dy, dx, h, w, channel_scale);
                    /* epfWeight (:671-679) */
                    int modY = y & 7, modX = x & 7;
                    if (modY == 0 || modY == 7 || modX == 0 || modX == 7) dist *= border_sad_mul;
                    float v = 1.0f - dist * sigmaScale * s;
                    float weight = v < 0.0f ? 0.0f : v;
                    sumWeights += weight;
                    int mY = mirror(y + dy, h);
                    int mX = mirror(x + dx, w);
                    for (int c = 0; c < 3; c++) sumChannels[c] += ib[c][(int64_t)mY * w + mX] * weight;
                }
                for (int c = 0; c < 3; c++) B[c][(int64_t)y * w + x] = sumChannels[c] / sumWeights;
            }
        }
        for (int c = 0; c < 3; c++) { /* buffer swap (:629-634) */
            float* t = A[c];
            A[c] = B[c];
            B[c] = t;
        }
    }
    for (int c = 0; c < 3; c++) {
        memcpy(out[c], A[c], n * sizeof(float));
        free(A[c]);
        free(B[c]);
    }
}

/* LFCoefficients ctor tail (LFCoefficients.java:65-103) + adaptiveSmooth (:113-180) */
void orc_lf_dequant(const jxl_lfquant_desc* d, float base_corr_x, float base_corr_b, int32_t color_factor, float* const out[3]) {
    const int H = d->cells_h, W = d->cells_w;
    const size_t n = (size_t)H * W;
    float* co[3];
    float* wgt[3];
    for (int i = 0; i < 3; i++) {
        co[i] = (float*)malloc(sizeof(float) * (n ? n : 1));
        wgt[i] = (float*)calloc(n ? n : 1, sizeof(float));
        const float sd = d->scaled_dequant[i] / (float)(1 << d->extra_precision); /* :69 */
        for (size_t k = 0; k < n; k++) co[i][k] = (float)d->lf_quant[i][k] * sd;
    }
    { /* chroma from luma (:78-95), SPEC: -128 */
        const float kX = base_corr_x + ((float)d->x_factor_lf - 128.0f) / (float)color_factor;
        const float kB = base_corr_b + ((float)d->b_factor_lf - 128.0f) / (float)color_factor;
        for (size_t k = 0; k < n; k++) {
            co[0][k] += kX * co[1][k];
            co[2][k] += kB * co[1][k];
        }
    }
    if (!d->adaptive_smoothing) {
        for (int i = 0; i < 3; i++) memcpy(out[i], co[i], sizeof(float) * n);
    } else {
        float* gap = (float*)malloc(sizeof(float) * (n ? n : 1));
        for (size_t k = 0; k < n; k++) gap[k] = 0.5f; /* rows 1..H-2 are the only ones ever read */
        for (int i = 0; i < 3; i++) {
            const float sd = d->scaled_dequant[i];
            for (int y = 1; y < H - 1; y++) {
                const float* coy = co[i] + (size_t)y * W;
                const float* coym = coy - W;
                const float* coyp = coy + W;
                for (int x = 1; x < W - 1; x++) {
                    const float sample = coy[x];
                    const float adjacent = coy[x - 1] + coy[x + 1] + coym[x] + coyp[x];
                    const float diag = coym[x - 1] + coym[x + 1] + coyp[x - 1] + coyp[x + 1];
                    const float wv = 0.05226273532324128f * sample + 0.20345139757231578f * adjacent + 0.0334829185968739f * diag;
                    wgt[i][(size_t)y * W + x] = wv;
                    const float g = fabsf(sample - wv) * sd;
                    if (g > gap[(size_t)y * W + x]) gap[(size_t)y * W + x] = g;
                }
            }
        }
        for (size_t k = 0; k < n; k++) {
            const float v = 3.0f - 4.0f * gap[k];
            gap[k] = v > 0.0f ? v : 0.0f; /* Math.max(0f, 3f - 4f * g) */
        }
        for (int i = 0; i < 3; i++)
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) {
                    const size_t k = (size_t)y * W + x;
                    if (y == 0 || y + 1 == H || x == 0 || x + 1 == W) out[i][k] = co[i][k];
                    else out[i][k] = (co[i][k] - wgt[i][k]) * gap[k] + wgt[i][k];
                }
        free(gap);
    }
    for (int i = 0; i < 3; i++) {
        free(co[i]);
        free(wgt[i]);
    }
}

/* OpsinInverseMatrix.invertXYB (OpsinInverseMatrix.java:105-142) */
void orc_xyb(float* const planes[3], int64_t n, const float matrix[9], const float opsin_bias[3],
             const float cbrt_opsin_bias[3], float intensity_target) {
    const float itScale = 255.0f / intensity_target;
    float sm[9];
    for (int i = 0; i < 9; i++) sm[i] = matrix[i] * itScale;
    const float ob0 = opsin_bias[0], ob1 = opsin_bias[1], ob2 = opsin_bias[2];
    const float cob0 = -cbrt_opsin_bias[0], cob1 = -cbrt_opsin_bias[1], cob2 = -cbrt_opsin_bias[2];
    float* X = planes[0];
    float* Y = planes[1];
    float* B = planes[2];
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        const float xybX = X[i], xybY = Y[i], xybB = B[i];
        const float gammaL = xybY + xybX + cob0;
        const float gammaM = xybY - xybX + cob1;
        const float gammaS = xybB + cob2;
        const float mixL = (gammaL * gammaL) * gammaL + ob0;
        const float mixM = (gammaM * gammaM) * gammaM + ob1;
        const float mixS = (gammaS * gammaS) * gammaS + ob2;
        X[i] = sm[0] * mixL + sm[1] * mixM + sm[2] * mixS;
        Y[i] = sm[3] * mixL + sm[4] * mixM + sm[5] * mixS;
        B[i] = sm[6] * mixL + sm[7] * mixM + sm[8] * mixS;
    }
}

/* YCbCr branch of performColorTransforms (JXLCodestreamDecoder.java:270-281) */
void orc_ycbcr(float* const planes[3], int64_t n) {
    for (int64_t i = 0; i < n; i++) {
        float cb = planes[0][i];
        float yh = planes[1][i] + 0.50196078431372549019f;
        float cr = planes[2][i];
        planes[0][i] = yh + 1.402f * cr;
        planes[1][i] = yh - 0.34413628620102214650f * cb - 0.71413628620102214650f * cr;
        planes[2][i] = yh + 1.772f * cb;
    }
}

/* TransferFunction.TF_PQ.fromLinear via the default fromLinearF (TransferFunction.java:83-87,104-106) */
static inline float tf_pq(float f) {
    double d = pow((double)f, 0.159423828125);
    return (float)pow((0.8359375 + 18.8515625 * d) / (1.0 + 18.6875 * d), 78.84375);
}
/* TransferFunction.TF_SRGB.fromLinearF (:39-44) */
static inline float tf_srgb(float f) {
    if (f < 0.00313066844250063f) return f * 12.92f;
    return 1.055f * (float)pow((double)f, 0.4166666666666667) + -0.055f;
}

/* JXLImage.transferInPlace (JXLImage.java:244-258) + ImageBuffer.castToInt0 (ImageBuffer.java:129-147) */
void orc_transfer(const float* in, int64_t n, int transfer, int max_value, float* out_f, int32_t* out_i) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        float v = in[i];
        if (transfer == JXL_TRANSFER_PQ) v = tf_pq(v);
        else if (transfer == JXL_TRANSFER_SRGB) v = tf_srgb(v);
        if (max_value > 0) {
            float scaleFactor = (float)max_value;
            int32_t q = java_f2i(v * scaleFactor + 0.5f);
            out_i[i] = q < 0 ? 0 : q > max_value ? max_value : q;
        } else {
            out_f[i] = v;
        }
    }
}

/* ---- whole VarDCT frame ------------------------------------------------------------------ */
jxl_status orc_vardct_frame_run(const orc_vardct_frame* f, void* const out[3]) {
    const jxl_vardct_params* p = &f->p;
    const int W = p->width, H = p->height;
    if (W <= 0 || H <= 0 || (W & 7) || (H & 7)) return JXL_ERR_INVALID_ARGUMENT;
    lut_init();
    const size_t n = (size_t)W * H;
    float* buf[3];
    float* tmp[3];
    for (int c = 0; c < 3; c++) {
        buf[c] = (float*)calloc(n, sizeof(float));
        tmp[c] = (float*)calloc(n, sizeof(float));
        if (!buf[c] || !tmp[c]) return JXL_ERR_OOM;
    }
    jxl_status st = JXL_OK;
    const int ngroups = ceil_div(W, 256) * ceil_div(H, 256);
    const int nthreads = f->threads > 1 ? f->threads : 1;
#ifdef _OPENMP
    omp_set_num_threads(nthreads);
#endif

    if (p->stages & JXL_STAGE_IDCT) {
        /* Frame.decodePassGroups tail (Frame.java:367-373): groups in order */
#pragma omp parallel for schedule(dynamic) num_threads(nthreads)
        for (int g = 0; g < ngroups; g++) {
            jxl_status s2 = vardct_group(f, g, buf);
            if (s2 != JXL_OK) {
#pragma omp critical(orc_st)
                st = s2;
            }
        }
    } else {
        /* stage tests may feed pixel planes through coeff[] reinterpretation: not supported */
    }
    if (st != JXL_OK) goto done;

    /* Frame.invertSubsampling (Frame.java:457, 681-723): channels decoded at (H >> sy) x (W >> sx) grow to H x W */
    if (p->stages & JXL_STAGE_IDCT)
        for (int c = 0; c < 3; c++) {
            const int sy = p->jpeg_upsampling_y[c], sx = p->jpeg_upsampling_x[c];
            if (sy < 0 || sx < 0 || sy > 2 || sx > 2) { st = JXL_ERR_INVALID_ARGUMENT; goto done; }
            if (!sy && !sx) continue;
            orc_chroma_upsample(buf[c], H >> sy, W >> sx, sx, sy, tmp[c]);
            float* t = buf[c];
            buf[c] = tmp[c];
            tmp[c] = t;
        }

    if ((p->stages & JXL_STAGE_GAB) && p->gab) {
        const float* const ib[3] = {buf[0], buf[1], buf[2]};
        orc_gab(ib, tmp, H, W, p->gab_w1, p->gab_w2);
        for (int c = 0; c < 3; c++) { /* buffer[c] = newBuffer (Frame.java:540) */
            float* t = buf[c];
            buf[c] = tmp[c];
            tmp[c] = t;
        }
    }
    if ((p->stages & JXL_STAGE_EPF) && p->epf_iters > 0) {
        const int bh = (H + 7) >> 3, bw = (W + 7) >> 3;
        const int lrs = ceil_div(W, 2048);
        float* inv_sigma = (float*)malloc(sizeof(float) * (size_t)bh * bw);
        for (int y = 0; y < bh && st == JXL_OK; y++) { /* Frame.java:555-571 */
            int lfY = y >> 8, bY = y - (lfY << 8);
            for (int x = 0; x < bw; x++) {
                int lfX = x >> 8, bX = x - (lfX << 8);
                const jxl_lfgroup_desc* lfg = &f->lfg[lfY * lrs + lfX];
                int hf = lfg->hf_mul[bY * lfg->cells_w + bX];
                int sharp = lfg->sharpness[bY * lfg->cells_w + bX];
                if (sharp < 0 || sharp > 7) { st = JXL_ERR_INVALID_BITSTREAM; break; }
                float sigma = p->global_scale_f * p->epf_sharp_lut[sharp] / (float)hf;
                inv_sigma[y * bw + x] = 1.0f / sigma;
            }
        }
        if (st == JXL_OK) {
            const float* const ib[3] = {buf[0], buf[1], buf[2]};
            orc_epf(ib, tmp, H, W, p->epf_iters, inv_sigma, 0.0f, p->epf_channel_scale, p->epf_pass0_sigma_scale,
                    p->epf_pass2_sigma_scale, p->epf_border_sad_mul);
            for (int c = 0; c < 3; c++) {
                float* t = buf[c];
                buf[c] = tmp[c];
                tmp[c] = t;
            }
        }
        free(inv_sigma);
        if (st != JXL_OK) goto done;
    }
    if ((p->stages & JXL_STAGE_XYB) && p->xyb)
        orc_xyb(buf, (int64_t)n, p->opsin_matrix, p->opsin_bias, p->cbrt_opsin_bias, p->intensity_target);

    if ((p->stages & JXL_STAGE_OUT) && (p->transfer != JXL_TRANSFER_NONE || p->out_format != JXL_OUT_F32)) {
        /* the interleaved formats (row f3) hold the same samples; the oracle always returns int32 planes */
        int maxv = (p->out_format == JXL_OUT_U16 || p->out_format == JXL_OUT_RGB16) ? 65535
                 : (p->out_format == JXL_OUT_U8 || p->out_format == JXL_OUT_RGB8) ? 255 : 0;
        for (int c = 0; c < 3; c++)
            orc_transfer(buf[c], (int64_t)n, p->transfer, maxv, (float*)out[c], (int32_t*)out[c]);
    } else {
        for (int c = 0; c < 3; c++) memcpy(out[c], buf[c], n * sizeof(float));
    }
done:
    for (int c = 0; c < 3; c++) {
        free(buf[c]);
        free(tmp[c]);
    }
    return st;
}

/* ---- Modular: squeeze -------------------------------------------------------------------- */
/* ModularChannel.tendency (ModularChannel.java:23-47), int32 wrap-around */
static inline int32_t tendency(int32_t a, int32_t b, int32_t c) {
    if (a >= b && b >= c) {
        int32_t x = wadd(wsub(wsub(wmul(4, a), wmul(3, c)), b), 6) / 12;
        int32_t d = wmul(2, wsub(a, b));
        int32_t e = wmul(2, wsub(b, c));
        if (wsub(x, (x & 1)) > d) x = wadd(d, 1);
        if (wadd(x, (x & 1)) > e) x = e;
        return x;
    }
    if (a <= b && b <= c) {
        int32_t x = wsub(wsub(wsub(wmul(4, a), wmul(3, c)), b), 6) / 12;
        int32_t d = wmul(2, wsub(a, b));
        int32_t e = wmul(2, wsub(b, c));
        if (wadd(x, (x & 1)) < d) x = wsub(d, 1);
        if (wsub(x, (x & 1)) < e) x = e;
        return x;
    }
    return 0;
}

/* ModularChannel.inverseHorizontalSqueeze (ModularChannel.java:361-387) */
void orc_inv_hsqueeze(const int32_t* avg, int aw, const int32_t* res, int rw, int h, int32_t* out) {
    const int ow = aw + rw;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        const int32_t* A = avg + (int64_t)y * aw;
        const int32_t* R = res + (int64_t)y * rw;
        int32_t* O = out + (int64_t)y * ow;
        for (int x = 0; x < rw; x++) {
            int32_t a = A[x];
            int32_t residu = R[x];
            int32_t nextAvg = x + 1 < aw ? A[x + 1] : a;
            int32_t left = x > 0 ? O[2 * x - 1] : a;
            int32_t diff = wadd(residu, tendency(left, a, nextAvg));
            int32_t first = wadd(a, diff / 2);
            O[2 * x] = first;
            O[2 * x + 1] = wsub(first, diff);
        }
        if (aw > rw) O[2 * rw] = A[rw];
    }
}

/* ModularChannel.inverseVerticalSqueeze (ModularChannel.java:389-413) */
void orc_inv_vsqueeze(const int32_t* avg, int ah, const int32_t* res, int rh, int w, int32_t* out) {
    for (int y = 0; y < rh; y++) {
        for (int x = 0; x < w; x++) {
            int32_t a = avg[(int64_t)y * w + x];
            int32_t residu = res[(int64_t)y * w + x];
            int32_t nextAvg = y + 1 < ah ? avg[(int64_t)(y + 1) * w + x] : a;
            int32_t top = y > 0 ? out[(int64_t)(2 * y - 1) * w + x] : a;
            int32_t diff = wadd(residu, tendency(top, a, nextAvg));
            int32_t first = wadd(a, diff / 2);
            out[(int64_t)(2 * y) * w + x] = first;
            out[(int64_t)(2 * y + 1) * w + x] = wsub(first, diff);
        }
    }
    if (ah > rh) memcpy(out + (int64_t)(2 * rh) * w, avg + (int64_t)rh * w, sizeof(int32_t) * (size_t)w);
}

/* TEST-ONLY forward steps (no reference counterpart): avg = a - trunc((a-b)/2) so that the
 * inverse's first = avg + diff/2 returns a; res = diff - tendency(prev_b, avg, next_avg). */
void orc_fwd_hsqueeze(const int32_t* in, int h, int w, int32_t* avg, int32_t* res) {
    const int aw = (w + 1) / 2, rw = w / 2;
    for (int y = 0; y < h; y++) {
        const int32_t* I = in + (int64_t)y * w;
        int32_t* A = avg + (int64_t)y * aw;
        int32_t* R = res + (int64_t)y * rw;
        for (int x = 0; x < rw; x++) {
            int32_t a = I[2 * x], b = I[2 * x + 1];
            int32_t diff = wsub(a, b);
            A[x] = wsub(a, diff / 2);
        }
        if (aw > rw) A[rw] = I[2 * rw];
        for (int x = 0; x < rw; x++) {
            int32_t a = I[2 * x], b = I[2 * x + 1];
            int32_t diff = wsub(a, b);
            int32_t nextAvg = x + 1 < aw ? A[x + 1] : A[x];
            int32_t left = x > 0 ? I[2 * x - 1] : A[x];
            R[x] = wsub(diff, tendency(left, A[x], nextAvg));
        }
    }
}

void orc_fwd_vsqueeze(const int32_t* in, int h, int w, int32_t* avg, int32_t* res) {
    const int ah = (h + 1) / 2, rh = h / 2;
    for (int y = 0; y < rh; y++)
        for (int x = 0; x < w; x++) {
            int32_t a = in[(int64_t)(2 * y) * w + x], b = in[(int64_t)(2 * y + 1) * w + x];
            avg[(int64_t)y * w + x] = wsub(a, wsub(a, b) / 2);
        }
    if (ah > rh) memcpy(avg + (int64_t)rh * w, in + (int64_t)(2 * rh) * w, sizeof(int32_t) * (size_t)w);
    for (int y = 0; y < rh; y++)
        for (int x = 0; x < w; x++) {
            int32_t a = in[(int64_t)(2 * y) * w + x], b = in[(int64_t)(2 * y + 1) * w + x];
            int32_t diff = wsub(a, b);
            int32_t A0 = avg[(int64_t)y * w + x];
            int32_t nextAvg = y + 1 < ah ? avg[(int64_t)(y + 1) * w + x] : A0;
            int32_t top = y > 0 ? in[(int64_t)(2 * y - 1) * w + x] : A0;
            res[(int64_t)y * w + x] = wsub(diff, tendency(top, A0, nextAvg));
        }
}

/* RCT branch of ModularStream.applyTransforms (ModularStream.java:255-326) */
jxl_status orc_rct(int32_t* const v[3], int64_t n, int rct_type) {
    static const int permutationLut[6][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}, {0, 2, 1}, {1, 0, 2}, {2, 1, 0}}; /* :35-38 */
    if (rct_type < 0 || rct_type >= 42) return JXL_ERR_INVALID_ARGUMENT;
    int permutation = rct_type / 7, type = rct_type % 7;
    int32_t *v0 = v[0], *v1 = v[1], *v2 = v[2];
    for (int64_t i = 0; i < n; i++) {
        switch (type) {
        case 0: break;
        case 1: v2[i] = wadd(v2[i], v0[i]); break;
        case 2: v1[i] = wadd(v1[i], v0[i]); break;
        case 3: { int32_t a = v0[i]; v2[i] = wadd(v2[i], a); v1[i] = wadd(v1[i], a); break; }
        case 4: v1[i] = wadd(v1[i], wadd(v0[i], v2[i]) >> 1); break;
        case 5: { int32_t a = v0[i]; int32_t ac = wadd(a, v2[i]); v1[i] = wadd(v1[i], wadd(a, ac) >> 1); v2[i] = ac; break; }
        case 6: {
            int32_t b = v1[i], c = v2[i];
            int32_t tmp = wsub(v0[i], c >> 1);
            int32_t f = wsub(tmp, b >> 1);
            v0[i] = wadd(f, b);
            v1[i] = wadd(c, tmp);
            v2[i] = f;
            break;
        }
        }
    }
    /* channels.set(start + permutationLut[permutation][j], v[j]) (:325-326): apply to contents */
    if (permutation != 0) {
        int32_t* t = (int32_t*)malloc(sizeof(int32_t) * (size_t)n * 3);
        for (int j = 0; j < 3; j++) memcpy(t + (size_t)permutationLut[permutation][j] * n, v[j], sizeof(int32_t) * (size_t)n);
        for (int j = 0; j < 3; j++) memcpy(v[j], t + (size_t)j * n, sizeof(int32_t) * (size_t)n);
        free(t);
    }
    return JXL_OK;
}

/* Frame.decodeFrame modular -> float buffer (Frame.java:437-448) */
void orc_modular_to_float(const int32_t* a, const int32_t* b, int64_t n, float scale, float* out) {
    for (int64_t i = 0; i < n; i++) out[i] = b ? scale * (float)wadd(a[i], b[i]) : scale * (float)a[i];
}

/* default squeeze parameters (ModularStream.java:110-131) */
int32_t orc_default_squeeze_params(const int32_t* widths, const int32_t* heights, int32_t n_channels,
                                   int32_t nb_meta, jxl_squeeze_param* out, int32_t cap) {
    int n = 0;
    int first = nb_meta;
    int count = n_channels - first;
    if (count <= 0) return 0;
#define PUSH(H_, IP_, B_, N_) do { if (n >= cap) return JXL_ERR_INVALID_ARGUMENT; \
        out[n].horizontal = (H_); out[n].in_place = (IP_); out[n].begin_c = (B_); out[n].num_c = (N_); n++; } while (0)
    int sw = widths[0], sh = heights[0]; /* channels.get(0).size (:114) */
    if (count > 2 && sw == widths[first + 1] && sh == heights[first + 1]) {
        PUSH(1, 0, first + 1, 2);
        PUSH(0, 0, first + 1, 2);
    }
    if (sh >= sw && sh > 8) {
        PUSH(0, 1, first, count);
        sh = (sh + 1) / 2;
    }
    while (sw > 8 || sh > 8) {
        if (sw > 8) {
            PUSH(1, 1, first, count);
            sw = (sw + 1) / 2;
        }
        if (sh > 8) {
            PUSH(0, 1, first, count);
            sh = (sh + 1) / 2;
        }
    }
#undef PUSH
    return n;
}

/* forward channel-list surgery of the ModularStream ctor (ModularStream.java:134-167) */
int32_t orc_squeezed_shapes(const int32_t* widths, const int32_t* heights, int32_t n_channels,
                            const jxl_squeeze_param* sp, int32_t n_sp, int32_t* out_w, int32_t* out_h, int32_t cap) {
    int n = n_channels;
    if (n > cap) return JXL_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < n; i++) { out_w[i] = widths[i]; out_h[i] = heights[i]; }
    for (int j = 0; j < n_sp; j++) {
        int begin = sp[j].begin_c;
        int end = begin + sp[j].num_c - 1;
        if (begin < 0 || end >= n) return JXL_ERR_INVALID_BITSTREAM;
        int offset = sp[j].in_place ? end + 1 : n;
        for (int k = begin; k <= end; k++) {
            int r = offset + k - begin;
            int rw, rh;
            if (sp[j].horizontal) {
                int w = out_w[k];
                out_w[k] = (w + 1) / 2;
                rw = w / 2;
                rh = out_h[k];
            } else {
                int h = out_h[k];
                out_h[k] = (h + 1) / 2;
                rh = h / 2;
                rw = out_w[k];
            }
            if (n + 1 > cap) return JXL_ERR_INVALID_ARGUMENT;
            for (int m = n; m > r; m--) { out_w[m] = out_w[m - 1]; out_h[m] = out_h[m - 1]; } /* channels.add(r, residu) */
            out_w[r] = rw;
            out_h[r] = rh;
            n++;
        }
    }
    return n;
}

/* ModularStream.applyTransforms: SQUEEZE (ModularStream.java:229-254) then optional RCT */
jxl_status orc_modular_apply(const jxl_channel* chans, int32_t n_chans, const jxl_squeeze_param* sp, int32_t n_sp,
                             int32_t rct_type, int32_t rct_begin, jxl_channel* out, int32_t n_out) {
    int n = n_chans;
    jxl_channel* ch = (jxl_channel*)malloc(sizeof(jxl_channel) * (size_t)(n > 0 ? n : 1));
    uint8_t* owned = (uint8_t*)calloc((size_t)(n > 0 ? n : 1), 1);
    jxl_status st = JXL_OK;
    for (int i = 0; i < n; i++) ch[i] = chans[i];
    for (int j = n_sp - 1; j >= 0 && st == JXL_OK; j--) {
        int begin = sp[j].begin_c;
        int end = begin + sp[j].num_c - 1;
        int offset = sp[j].in_place ? end + 1 : n + begin - end - 1;
        if (begin < 0 || end >= n || offset < 0 || offset + (end - begin) >= n) { st = JXL_ERR_INVALID_BITSTREAM; break; }
        for (int c = begin; c <= end; c++) {
            int r = offset + c - begin;
            jxl_channel chan = ch[c], residu = ch[r], o;
            if (sp[j].horizontal) {
                /* shape checks of inverseHorizontalSqueeze (ModularChannel.java:363-366) */
                if ((chan.width != residu.width && chan.width != 1 + residu.width) || residu.height != chan.height) { st = JXL_ERR_INVALID_ARGUMENT; break; }
                o.width = chan.width + residu.width;
                o.height = chan.height;
                o.data = (int32_t*)malloc(sizeof(int32_t) * (size_t)(o.width * (int64_t)o.height > 0 ? o.width * (int64_t)o.height : 1));
                orc_inv_hsqueeze(chan.data, chan.width, residu.data, residu.width, chan.height, o.data);
            } else {
                if ((chan.height != residu.height && chan.height != 1 + residu.height) || residu.width != chan.width) { st = JXL_ERR_STATE; break; }
                o.width = chan.width;
                o.height = chan.height + residu.height;
                o.data = (int32_t*)malloc(sizeof(int32_t) * (size_t)(o.width * (int64_t)o.height > 0 ? o.width * (int64_t)o.height : 1));
                orc_inv_vsqueeze(chan.data, chan.height, residu.data, residu.height, chan.width, o.data);
            }
            if (owned[c]) free(ch[c].data);
            ch[c] = o;
            owned[c] = 1;
        }
        if (st != JXL_OK) break;
        int cnt = end - begin + 1;
        for (int c = 0; c < cnt; c++) { /* channels.remove(offset) x cnt */
            if (owned[offset]) free(ch[offset].data);
            for (int m = offset; m + 1 < n; m++) { ch[m] = ch[m + 1]; owned[m] = owned[m + 1]; }
            n--;
        }
    }
    if (st == JXL_OK && rct_type >= 0) {
        if (rct_begin < 0 || rct_begin + 2 >= n) st = JXL_ERR_INVALID_ARGUMENT;
        else {
            jxl_channel* v = &ch[rct_begin];
            if (v[1].width != v[0].width || v[1].height != v[0].height || v[2].width != v[1].width || v[2].height != v[1].height)
                st = JXL_ERR_INVALID_BITSTREAM; /* :266-267 */
            else {
                /* RCT mutates in place: make private copies of channels still aliasing the input */
                for (int j2 = 0; j2 < 3; j2++) {
                    if (!owned[rct_begin + j2]) {
                        size_t bytes = sizeof(int32_t) * (size_t)v[j2].width * v[j2].height;
                        int32_t* d = (int32_t*)malloc(bytes ? bytes : 4);
                        memcpy(d, v[j2].data, bytes);
                        v[j2].data = d;
                        owned[rct_begin + j2] = 1;
                    }
                }
                int32_t* vv[3] = {v[0].data, v[1].data, v[2].data};
                st = orc_rct(vv, (int64_t)v[0].width * v[0].height, rct_type);
            }
        }
    }
    if (st == JXL_OK) {
        if (n != n_out) st = JXL_ERR_INVALID_ARGUMENT;
        else
            for (int i = 0; i < n; i++) {
                if (out[i].width != ch[i].width || out[i].height != ch[i].height) { st = JXL_ERR_INVALID_ARGUMENT; break; }
                memcpy(out[i].data, ch[i].data, sizeof(int32_t) * (size_t)ch[i].width * ch[i].height);
            }
    }
    for (int i = 0; i < n; i++)
        if (owned[i]) free(ch[i].data);
    free(ch);
    free(owned);
    return st;
}
