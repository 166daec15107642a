/*
 * jxl_oracle_post.c -- CPU restatement (ORACLE, test infrastructure only; see jxl_oracle.h) of the
 * pixel-domain stages either side of the colour transform: SURVEY.md section 8 rows f4 (chroma upsampling,
 * k-times upsampling, noise synthesis) and f3 (blending, orientation, sample packing).
 *
 * PARITY UNPINNED, like the rest of the oracle: the reference holds no fixtures for these functions and
 * cannot run here (no JVM). Each function follows the cited Java lines statement by statement.
 */
#include "jxl_oracle.h"

#include <stdlib.h>
#include <string.h>

/* Java (int)float: NaN -> 0, saturating (JLS 5.1.3) */
static inline int32_t f2i(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}

/* MathHelper.mirrorCoordinate (MathHelper.java:323-329) */
static inline int mirror(int c, int size) {
    while (c < 0 || c >= size) {
        int tc = ~c;
        c = tc >= 0 ? tc : (size << 1) + tc;
    }
    return c;
}

/* MathHelper.clampAsc (MathHelper.java:215-217) */
static inline float clamp_asc(float v, float lo, float hi) { return v < lo ? lo : v > hi ? hi : v; }

/* ---- f4 ------------------------------------------------------------------------------------------------ */

/* Frame.invertSubsampling (Frame.java:681-723), one channel */
void orc_chroma_upsample(const float* in, int h, int w, int x_shift, int y_shift, float* out) {
    size_t cap = (size_t)(h << y_shift) * (size_t)(w << x_shift);
    float* cur = (float*)malloc(sizeof(float) * (cap ? cap : 1));
    float* nxt = (float*)malloc(sizeof(float) * (cap ? cap : 1));
    memcpy(cur, in, sizeof(float) * (size_t)h * w);
    int ch = h, cw = w;
    while (x_shift-- > 0) { /* :684-699 */
        for (int y = 0; y < ch; y++) {
            const float* oldRow = cur + (size_t)y * cw;
            float* newRow = nxt + (size_t)y * cw * 2;
            for (int x = 0; x < cw; x++) {
                float b75 = 0.75f * oldRow[x];
                newRow[2 * x] = b75 + 0.25f * oldRow[x == 0 ? 0 : x - 1];
                newRow[2 * x + 1] = b75 + 0.25f * oldRow[x + 1 == cw ? cw - 1 : x + 1];
            }
        }
        cw *= 2;
        float* t = cur; cur = nxt; nxt = t;
    }
    while (y_shift-- > 0) { /* :701-720 */
        for (int y = 0; y < ch; y++) {
            const float* oldRow = cur + (size_t)y * cw;
            const float* oldRowPrev = cur + (size_t)(y == 0 ? 0 : y - 1) * cw;
            const float* oldRowNext = cur + (size_t)(y + 1 == ch ? ch - 1 : y + 1) * cw;
            float* firstNewRow = nxt + (size_t)(2 * y) * cw;
            float* secondNewRow = nxt + (size_t)(2 * y + 1) * cw;
            for (int x = 0; x < cw; x++) {
                float b75 = 0.75f * oldRow[x];
                firstNewRow[x] = b75 + 0.25f * oldRowPrev[x];
                secondNewRow[x] = b75 + 0.25f * oldRowNext[x];
            }
        }
        ch *= 2;
        float* t = cur; cur = nxt; nxt = t;
    }
    memcpy(out, cur, sizeof(float) * (size_t)ch * cw);
    free(cur);
    free(nxt);
}

/* ImageHeader.getUpWeights (ImageHeader.java:441-470) for one k */
jxl_status orc_upsampling_weights(int k, const float* packed, float* out) {
    if (k != 2 && k != 4 && k != 8) return JXL_ERR_INVALID_ARGUMENT;
    for (int ky = 0; ky < k; ky++)
        for (int kx = 0; kx < k; kx++)
            for (int iy = 0; iy < 5; iy++)
                for (int ix = 0; ix < 5; ix++) {
                    int j = (ky < k / 2) ? (iy + 5 * ky) : ((4 - iy) + 5 * (k - 1 - ky));
                    int i = (kx < k / 2) ? (ix + 5 * kx) : ((4 - ix) + 5 * (k - 1 - kx));
                    int x = i < j ? j : i;
                    int y = x ^ j ^ i;
                    int index = 5 * k * y / 2 - y * (y - 1) / 2 + x - y;
                    out[((ky * k + kx) * 5 + iy) * 5 + ix] = packed[index];
                }
    return JXL_OK;
}

/* Frame.performUpsampling (Frame.java:217-260) */
void orc_upsample(const float* in, int h, int w, int k, const float* weights, float* out) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int ky = 0; ky < k; ky++)
            for (int x = 0; x < w; x++)
                for (int kx = 0; kx < k; kx++) {
                    const float* wt = weights + (size_t)(ky * k + kx) * 25;
                    float total = 0.0f;
                    float min = 3.4028234663852886e38f; /* Float.MAX_VALUE */
                    float max = 1.4e-45f;               /* Float.MIN_VALUE (:237): smallest positive, as in the reference */
                    for (int iy = 0; iy < 5; iy++)
                        for (int ix = 0; ix < 5; ix++) {
                            int newY = mirror(y + iy - 2, h);
                            int newX = mirror(x + ix - 2, w);
                            float sample = in[(size_t)newY * w + newX];
                            if (sample < min) min = sample;
                            if (sample > max) max = sample;
                            total += wt[iy * 5 + ix] * sample;
                        }
                    out[(size_t)(y * k + ky) * ((size_t)w * k) + (size_t)x * k + kx] = total < min ? min : total > max ? max : total;
                }
}

/* features/XorShiro.java */
static inline uint64_t split_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
typedef struct { uint64_t s0[8], s1[8]; } xorshiro;
static void xs_init(xorshiro* r, uint64_t seed0, uint64_t seed1) {
    r->s0[0] = split_mix64(seed0 + 0x9e3779b97f4a7c15ull);
    r->s1[0] = split_mix64(seed1 + 0x9e3779b97f4a7c15ull);
    for (int i = 1; i < 8; i++) {
        r->s0[i] = split_mix64(r->s0[i - 1]);
        r->s1[i] = split_mix64(r->s1[i - 1]);
    }
}
static void xs_fill(xorshiro* r, uint32_t batch[16]) { /* fillBatch; fill(bits) with bits.length == 16 */
    for (int i = 0; i < 8; i++) {
        const uint64_t a = r->s1[i];
        uint64_t b = r->s0[i];
        const uint64_t c = a + b;
        r->s0[i] = a;
        b ^= b << 23;
        r->s1[i] = b ^ a ^ (b >> 18) ^ (a >> 5);
        batch[2 * i] = (uint32_t)(c & 0xffffffffull);
        batch[2 * i + 1] = (uint32_t)(c >> 32);
    }
}

/* Frame.initializeNoise (Frame.java:748-788) */
void orc_noise_init(int h, int w, int group_dim, uint64_t seed0, int colors, float* const out[3]) {
    static const float laplacian[5][5] = { /* Frame.java:57-63 */
        {0.16f, 0.16f, 0.16f, 0.16f, 0.16f},  {0.16f, 0.16f, 0.16f, 0.16f, 0.16f}, {0.16f, 0.16f, -3.84f, 0.16f, 0.16f},
        {0.16f, 0.16f, 0.16f, 0.16f, 0.16f},  {0.16f, 0.16f, 0.16f, 0.16f, 0.16f}};
    const size_t n = (size_t)h * w;
    float* local[3] = {0, 0, 0};
    for (int c = 0; c < colors; c++) local[c] = (float*)calloc(n ? n : 1, sizeof(float));
    const int row_stride = (w + group_dim - 1) / group_dim;
    const int num_groups = row_stride * ((h + group_dim - 1) / group_dim);
    for (int group = 0; group < num_groups; group++) {
        const int y0 = (group / row_stride) * group_dim;
        const int x0 = (group % row_stride) * group_dim;
        const uint64_t seed1 = (((uint64_t)(uint32_t)x0) << 32) | (uint64_t)(uint32_t)y0;
        const int ySize = group_dim < h - y0 ? group_dim : h - y0;
        const int xSize = group_dim < w - x0 ? group_dim : w - x0;
        xorshiro rng;
        xs_init(&rng, seed0, seed1);
        uint32_t bits[16];
        for (int c = 0; c < colors; c++)
            for (int y = 0; y < ySize; y++)
                for (int x = 0; x < xSize; x += 16) {
                    xs_fill(&rng, bits);
                    for (int i = 0; i < 16 && x + i < xSize; i++) {
                        uint32_t f = (bits[i] >> 9) | 0x3f800000u;
                        memcpy(&local[c][(size_t)(y0 + y) * w + x0 + x + i], &f, 4);
                    }
                }
    }
    for (int c = 0; c < colors; c++) {
#pragma omp parallel for schedule(static)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                float acc = 0.0f;
                for (int iy = 0; iy < 5; iy++)
                    for (int ix = 0; ix < 5; ix++) {
                        int cy = mirror(y + iy - 2, h);
                        int cx = mirror(x + ix - 2, w);
                        acc += local[c][(size_t)cy * w + cx] * laplacian[iy][ix];
                    }
                out[c][(size_t)y * w + x] = acc;
            }
        free(local[c]);
    }
}

/* Frame.synthesizeNoise (Frame.java:790-831) */
void orc_noise_add(float* const planes[3], const float* const noise[3], int64_t n, const float lut[8], float base_corr_x,
                   float base_corr_b) {
    for (int64_t i = 0; i < n; i++) {
        float inScaledR = planes[1][i] + planes[0][i];
        inScaledR = inScaledR < 0.0f ? 0.0f : 3.0f * inScaledR;
        float inScaledG = planes[1][i] - planes[0][i];
        inScaledG = inScaledG < 0.0f ? 0.0f : 3.0f * inScaledG;
        int intInR, intInG;
        float fracInR, fracInG;
        if (inScaledR >= 7.0f) {
            intInR = 6;
            fracInR = 1.0f;
        } else {
            intInR = f2i(inScaledR);
            fracInR = inScaledR - (float)intInR;
        }
        if (inScaledG >= 7.0f) {
            intInG = 6;
            fracInG = 1.0f;
        } else {
            intInG = f2i(inScaledG);
            fracInG = inScaledG - (float)intInG;
        }
        float sr = (lut[intInR + 1] - lut[intInR]) * fracInR + lut[intInR];
        float sg = (lut[intInG + 1] - lut[intInG]) * fracInG + lut[intInG];
        sr = clamp_asc(sr, 0.0f, 1.0f);
        sg = clamp_asc(sg, 0.0f, 1.0f);
        float nr = sr * (0.00171875f * noise[0][i] + 0.21828125f * noise[2][i]);
        float ng = sg * (0.00171875f * noise[1][i] + 0.21828125f * noise[2][i]);
        float nrg = nr + ng;
        planes[1][i] += nrg;
        planes[0][i] += base_corr_x * nrg + nr - ng;
        planes[2][i] += base_corr_b * nrg;
    }
}

/* ---- f3 ------------------------------------------------------------------------------------------------ */

/* JXLCodestreamDecoder.java:26-40, 285-422 */
jxl_status orc_blend(int mode, uint32_t flags, int is_int, void* canvas, int ch, int cw, const void* frame, int fh, int fw,
                     const void* ref, int rh, int rw, const float* frame_alpha, const float* ref_alpha, const jxl_blend_rect* r) {
    (void)ch; (void)fh; (void)rh;
    const int isAlpha = (flags & JXL_BLEND_FLAG_IS_ALPHA) != 0, hasExtra = (flags & JXL_BLEND_FLAG_HAS_EXTRA) != 0;
    const int clamp = (flags & JXL_BLEND_FLAG_CLAMP) != 0, premult = (flags & JXL_BLEND_FLAG_PREMULT) != 0;
    /* which inner function runs */
    enum { COPY_FRAME, COPY_REF, ADD, MULT, BLEND, MULADD } op;
    switch (mode) {
        case JXL_BLEND_REPLACE: op = COPY_FRAME; break;
        case JXL_BLEND_ADD: op = ADD; break;
        case JXL_BLEND_MULT: op = MULT; break;
        case JXL_BLEND_BLEND: op = hasExtra ? BLEND : ADD; break;                               /* :346-349 */
        case JXL_BLEND_MULADD: op = !hasExtra ? ADD : isAlpha ? COPY_REF : MULADD; break;       /* :391-399 */
        default: return JXL_ERR_INVALID_BITSTREAM;                                              /* "Illegal blend mode" */
    }
    if (is_int && op != COPY_FRAME && op != COPY_REF && op != ADD) return JXL_ERR_INVALID_ARGUMENT;
    for (int y = 0; y < r->h; y++) {
        const int cy = y + r->canvas_y, fy = y + r->frame_y, ry = y + r->ref_y;
        for (int x = 0; x < r->w; x++) {
            const size_t ci = (size_t)cy * cw + (x + r->canvas_x);
            const size_t fi = (size_t)fy * fw + (x + r->frame_x);
            const size_t ri = (size_t)ry * rw + (x + r->ref_x);
            if (op == COPY_FRAME) { /* copyToCanvas(canvas, patchStart, frameOffset, size, frameBuffer) */
                ((uint32_t*)canvas)[ci] = ((const uint32_t*)frame)[fi];
            } else if (op == COPY_REF) { /* :396-398: copyToCanvas(canvas, patchStart, frameOffset, size, ref) -- frameOffset */
                ((uint32_t*)canvas)[ci] = ((const uint32_t*)ref)[(size_t)fy * rw + (x + r->frame_x)];
            } else if (op == ADD) {
                if (is_int)
                    ((int32_t*)canvas)[ci] = (int32_t)((uint32_t)((const int32_t*)ref)[ri] + (uint32_t)((const int32_t*)frame)[fi]);
                else
                    ((float*)canvas)[ci] = ((const float*)ref)[ri] + ((const float*)frame)[fi];
            } else if (op == MULT) {
                float newSample = ((const float*)frame)[fi];
                if (clamp) newSample = clamp_asc(newSample, 0.0f, 1.0f);
                ((float*)canvas)[ci] = newSample * ((const float*)ref)[ri];
            } else if (op == BLEND) {
                float oldSample = ((const float*)ref)[ri];
                float newSample = ((const float*)frame)[fi];
                float oldAlpha = isAlpha ? oldSample : ref_alpha[ri];
                float newAlpha = isAlpha ? newSample : frame_alpha[fi];
                if (clamp) newAlpha = clamp_asc(newAlpha, 0.0f, 1.0f);
                float v;
                if (isAlpha) v = oldAlpha + newAlpha * (1.0f - oldAlpha);
                else if (premult) v = newSample + oldSample * (1.0f - newAlpha);
                else v = (newSample * newAlpha + oldSample * oldAlpha * (1.0f - newAlpha)) / (oldAlpha + newAlpha * (1.0f - oldAlpha));
                ((float*)canvas)[ci] = v;
            } else {
                float oldSample = ((const float*)ref)[ri];
                float newSample = ((const float*)frame)[fi];
                float newAlpha = frame_alpha[fi];
                if (clamp) newAlpha = clamp_asc(newAlpha, 0.0f, 1.0f);
                ((float*)canvas)[ci] = oldSample + newAlpha * newSample;
            }
        }
    }
    return JXL_OK;
}

/* JXLCodestreamDecoder.transposeBufferFloat / transposeBufferInt (:43-177) */
jxl_status orc_orient(const void* in, int srcHeight, int srcWidth, int orientation, void* out) {
    const uint32_t* src = (const uint32_t*)in;
    uint32_t* dest = (uint32_t*)out;
    const int srcH1 = srcHeight - 1, srcW1 = srcWidth - 1;
    if (orientation < 1 || orientation > 8) return JXL_ERR_STATE;
    for (int y = 0; y < srcHeight; y++)
        for (int x = 0; x < srcWidth; x++) {
            const uint32_t v = src[(size_t)y * srcWidth + x];
            switch (orientation) {
                case 1: dest[(size_t)y * srcWidth + x] = v; break;
                case 2: dest[(size_t)y * srcWidth + (srcW1 - x)] = v; break;
                case 3: dest[(size_t)(srcH1 - y) * srcWidth + (srcW1 - x)] = v; break;
                case 4: dest[(size_t)(srcH1 - y) * srcWidth + x] = v; break;
                case 5: dest[(size_t)x * srcHeight + y] = v; break;
                case 6: dest[(size_t)x * srcHeight + (srcH1 - y)] = v; break;
                case 7: dest[(size_t)(srcW1 - x) * srcHeight + (srcH1 - y)] = v; break;
                default: dest[(size_t)(srcW1 - x) * srcHeight + y] = v; break;
            }
        }
    return JXL_OK;
}

/* PNGWriter ctor tail (PNGWriter.java:79-111) + writeIDAT sample order (:191-203) */
jxl_status orc_pack(const void* const planes[4], const jxl_pack_params* p, void* out) {
    const int nch = p->n_color + (p->has_alpha ? 1 : 0);
    if ((p->n_color != 1 && p->n_color != 3) || (p->bit_depth != 8 && p->bit_depth != 16)) return JXL_ERR_INVALID_ARGUMENT;
    if (p->premultiplied && !p->has_alpha) return JXL_ERR_INVALID_ARGUMENT;
    const int maxValue = ~(~0 << p->bit_depth);
    const size_t n = (size_t)p->height * p->width;
    int coerce = p->premultiplied; /* :79 */
    if (!coerce)
        for (int c = 0; c < nch; c++)
            if (p->is_int[c] && p->tagged_depth[c] != p->bit_depth) { coerce = 1; break; }
    for (int c = 0; c < nch; c++)
        if (p->is_int[c] && coerce && (p->tagged_depth[c] < 1 || p->tagged_depth[c] > 31)) return JXL_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < n; i++) {
        float fv[4];
        int32_t iv[4];
        int isf[4];
        for (int c = 0; c < nch; c++) {
            isf[c] = !p->is_int[c];
            if (p->is_int[c]) iv[c] = ((const int32_t*)planes[c])[i];
            else fv[c] = ((const float*)planes[c])[i];
            if (coerce && p->is_int[c]) { /* castToFloat(taggedBitDepth), ImageBuffer.java:112-127 */
                float scaleFactor = 1.0f / (float)(~(~0 << p->tagged_depth[c]));
                fv[c] = (float)iv[c] * scaleFactor;
                isf[c] = 1;
            }
        }
        if (p->premultiplied) /* :91-101 */
            for (int c = 0; c < p->n_color; c++) fv[c] /= fv[p->n_color];
        for (int c = 0; c < nch; c++) {
            int v;
            if (!isf[c]) { /* isInt && tagged == bitDepth: clamp(maxValue) */
                v = iv[c];
                v = v < 0 ? 0 : v > maxValue ? maxValue : v;
            } else { /* castToIntWithMax(maxValue), ImageBuffer.java:129-147 */
                v = f2i(fv[c] * (float)maxValue + 0.5f);
                v = v < 0 ? 0 : v > maxValue ? maxValue : v;
            }
            if (p->bit_depth == 8) ((uint8_t*)out)[i * nch + c] = (uint8_t)v;
            else if (p->big_endian) {
                ((uint8_t*)out)[(i * nch + c) * 2] = (uint8_t)(v >> 8);
                ((uint8_t*)out)[(i * nch + c) * 2 + 1] = (uint8_t)v;
            } else ((uint16_t*)out)[i * nch + c] = (uint16_t)v;
        }
    }
    return JXL_OK;
}
