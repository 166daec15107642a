// Pixel-domain stages either side of the colour transform:
//   f4: Frame.invertSubsampling, Frame.performUpsampling, Frame.initializeNoise / synthesizeNoise
//   f3: JXLCodestreamDecoder blend functions, transposeBuffer (orientation), PNGWriter sample packing
// Element-wise / small-stencil kernels, HBM-bound; every expression keeps the reference's association
// (file compiled with -ffp-contract=off).
#include <hip/hip_runtime.h>

#include "jxl_internal.h"

namespace jxl {
namespace {

// MathHelper.mirrorCoordinate (MathHelper.java:323-329)
__device__ __forceinline__ int mirror_c(int c, int size) {
    while (c < 0 || c >= size) {
        const int tc = ~c;
        c = tc >= 0 ? tc : (size << 1) + tc;
    }
    return c;
}
// Java (int)float
__device__ __forceinline__ int32_t f2i_sat(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}
__device__ __forceinline__ float clamp_asc(float v, float lo, float hi) { return v < lo ? lo : v > hi ? hi : v; }

// ---- Frame.invertSubsampling (Frame.java:681-723) -------------------------------------------------------
// horizontal doubling: one thread per input sample, two outputs (an 8-byte store)
__global__ __launch_bounds__(256) void k_chroma_up_h(const float* __restrict__ in, int h, int w, float* __restrict__ out) {
    const int64_t n = (int64_t)h * w;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w);
        const float* row = in + (i - x);
        const float b75 = 0.75f * row[x];
        float2 o;
        o.x = b75 + 0.25f * row[x == 0 ? 0 : x - 1];
        o.y = b75 + 0.25f * row[x + 1 == w ? w - 1 : x + 1];
        *reinterpret_cast<float2*>(out + 2 * i) = o;
    }
}
// vertical doubling: one thread per input sample, rows 2y and 2y+1
__global__ __launch_bounds__(256) void k_chroma_up_v(const float* __restrict__ in, int h, int w, float* __restrict__ out) {
    const int64_t n = (int64_t)h * w;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / w), x = (int)(i - (int64_t)y * w);
        const float b75 = 0.75f * in[i];
        const float prev = in[(int64_t)(y == 0 ? 0 : y - 1) * w + x];
        const float next = in[(int64_t)(y + 1 == h ? h - 1 : y + 1) * w + x];
        out[(int64_t)(2 * y) * w + x] = b75 + 0.25f * prev;
        out[(int64_t)(2 * y + 1) * w + x] = b75 + 0.25f * next;
    }
}

// ---- Frame.performUpsampling (Frame.java:217-260) --------------------------------------------------------
// one thread per (input pixel, ky): K contiguous outputs. The 25 taps, their min / max window and the weights
// (LDS) are shared by the K outputs; sum order iy, ix from 0f as in the reference.
template <int K>
__global__ __launch_bounds__(256) void k_upsample(const float* __restrict__ in, int h, int w, const float* __restrict__ weights,
                                                  float* __restrict__ out) {
    __shared__ float wl[K * K * 25];
    for (int i = threadIdx.x; i < K * K * 25; i += 256) wl[i] = weights[i];
    __syncthreads();
    const int64_t n = (int64_t)h * w * K;
    const int64_t ow = (int64_t)w * K;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        // i = (y*K + ky) * w + x: consecutive threads walk along an output row
        const int x = (int)(i % w);
        const int64_t oy = i / w;
        const int y = (int)(oy / K), ky = (int)(oy % K);
        float s[25];
        float mn = 3.4028234663852886e38f;
        float mx = 1.4e-45f;  // Float.MIN_VALUE (:237), not -MAX_VALUE: kept as the reference has it
#pragma unroll
        for (int iy = 0; iy < 5; iy++) {
            const int ny = mirror_c(y + iy - 2, h);
#pragma unroll
            for (int ix = 0; ix < 5; ix++) {
                const int nx = mirror_c(x + ix - 2, w);
                const float v = in[(int64_t)ny * w + nx];
                s[iy * 5 + ix] = v;
                if (v < mn) mn = v;
                if (v > mx) mx = v;
            }
        }
        float* o = out + oy * ow + (int64_t)x * K;
#pragma unroll
        for (int kx = 0; kx < K; kx++) {
            const float* wt = wl + (ky * K + kx) * 25;
            float total = 0.0f;
#pragma unroll
            for (int t = 0; t < 25; t++) total += wt[t] * s[t];
            o[kx] = total < mn ? mn : total > mx ? mx : total;
        }
    }
}

// ---- Frame.initializeNoise (Frame.java:748-788) + features/XorShiro.java --------------------------------
__device__ __forceinline__ uint64_t split_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
// One group per 8 lanes: lane i owns XorShiro's state0[i] / state1[i] and emits batch[2i], batch[2i+1], i.e.
// pixels x+2i and x+2i+1 of each run of 16. The stream is serial in (c, y, x/16) inside a group, groups are independent.
__global__ __launch_bounds__(64) void k_noise_rng(int h, int w, int group_dim, uint64_t seed0, int colors, int row_stride,
                                                  int num_groups, float* o0, float* o1, float* o2) {
    const int lane = threadIdx.x & 7;
    const int group = blockIdx.x * 8 + (threadIdx.x >> 3);
    if (group >= num_groups) return;
    const int y0 = (group / row_stride) * group_dim;
    const int x0 = (group % row_stride) * group_dim;
    const uint64_t seed1 = ((uint64_t)(uint32_t)x0 << 32) | (uint64_t)(uint32_t)y0;
    uint64_t s0 = split_mix64(seed0 + 0x9e3779b97f4a7c15ull);
    uint64_t s1 = split_mix64(seed1 + 0x9e3779b97f4a7c15ull);
    for (int i = 0; i < lane; i++) {
        s0 = split_mix64(s0);
        s1 = split_mix64(s1);
    }
    const int ySize = min(group_dim, h - y0);
    const int xSize = min(group_dim, w - x0);
    float* outs[3] = {o0, o1, o2};
    for (int c = 0; c < colors; c++) {
        float* o = outs[c];
        for (int y = 0; y < ySize; y++) {
            float* row = o + (int64_t)(y0 + y) * w + x0;
            for (int x = 0; x < xSize; x += 16) {
                const uint64_t a = s1;
                uint64_t b = s0;
                const uint64_t cc = a + b;
                s0 = a;
                b ^= b << 23;
                s1 = b ^ a ^ (b >> 18) ^ (a >> 5);
                const uint32_t lo = (uint32_t)cc, hi = (uint32_t)(cc >> 32);
                const int px = x + 2 * lane;
                if (px < xSize) row[px] = __uint_as_float((lo >> 9) | 0x3f800000u);
                if (px + 1 < xSize) row[px + 1] = __uint_as_float((hi >> 9) | 0x3f800000u);
            }
        }
    }
}
// the 5x5 high-pass: 0.16 everywhere, -3.84 at the centre (Frame.java:57-63), accumulated iy, ix from 0f
__global__ __launch_bounds__(256) void k_noise_conv(const float* __restrict__ in, int h, int w, float* __restrict__ out) {
    const int64_t n = (int64_t)h * w;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int y = (int)(i / w), x = (int)(i - (int64_t)y * w);
        float acc = 0.0f;
#pragma unroll
        for (int iy = 0; iy < 5; iy++) {
            const int cy = mirror_c(y + iy - 2, h);
#pragma unroll
            for (int ix = 0; ix < 5; ix++) {
                const int cx = mirror_c(x + ix - 2, w);
                acc += in[(int64_t)cy * w + cx] * ((iy == 2 && ix == 2) ? -3.84f : 0.16f);
            }
        }
        out[i] = acc;
    }
}

// ---- Frame.synthesizeNoise (Frame.java:790-831) ----------------------------------------------------------
struct NoiseLut {
    float v[8];
};
__global__ __launch_bounds__(256) void k_noise_add(float* p0, float* p1, float* p2, const float* __restrict__ n0,
                                                   const float* __restrict__ n1, const float* __restrict__ n2, int64_t n,
                                                   NoiseLut lut, float bcx, float bcb) {
    __shared__ float l[8];
    if (threadIdx.x < 8) l[threadIdx.x] = lut.v[threadIdx.x];
    __syncthreads();
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float bx = p0[i], by = p1[i], bb = p2[i];
        float inR = by + bx;
        inR = inR < 0.0f ? 0.0f : 3.0f * inR;
        float inG = by - bx;
        inG = inG < 0.0f ? 0.0f : 3.0f * inG;
        int iR, iG;
        float fR, fG;
        if (inR >= 7.0f) { iR = 6; fR = 1.0f; } else { iR = f2i_sat(inR); fR = inR - (float)iR; }
        if (inG >= 7.0f) { iG = 6; fG = 1.0f; } else { iG = f2i_sat(inG); fG = inG - (float)iG; }
        float sr = (l[iR + 1] - l[iR]) * fR + l[iR];
        float sg = (l[iG + 1] - l[iG]) * fG + l[iG];
        sr = clamp_asc(sr, 0.0f, 1.0f);
        sg = clamp_asc(sg, 0.0f, 1.0f);
        const float nr = sr * (0.00171875f * n0[i] + 0.21828125f * n2[i]);
        const float ng = sg * (0.00171875f * n1[i] + 0.21828125f * n2[i]);
        const float nrg = nr + ng;
        p1[i] = by + nrg;
        p0[i] = bx + (bcx * nrg + nr - ng);
        p2[i] = bb + bcb * nrg;
    }
}

// ---- blending (JXLCodestreamDecoder.java:26-40, 285-422) -------------------------------------------------
enum BlendOp { OP_COPY_FRAME, OP_COPY_REF, OP_ADD_I, OP_ADD_F, OP_MULT, OP_BLEND, OP_MULADD };
struct BlendArgs {
    void* canvas;
    const void* frame;
    const void* ref;
    const float* frame_alpha;
    const float* ref_alpha;
    int cw, fw, rw;
    jxl_blend_rect r;
    int op, is_alpha, clamp, premult;
};
__global__ __launch_bounds__(256) void k_blend(BlendArgs a) {
    const int64_t n = (int64_t)a.r.h * a.r.w;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int y = (int)(i / a.r.w), x = (int)(i - (int64_t)y * a.r.w);
        const int64_t ci = (int64_t)(y + a.r.canvas_y) * a.cw + (x + a.r.canvas_x);
        const int64_t fi = (int64_t)(y + a.r.frame_y) * a.fw + (x + a.r.frame_x);
        const int64_t ri = (int64_t)(y + a.r.ref_y) * a.rw + (x + a.r.ref_x);
        switch (a.op) {
            case OP_COPY_FRAME: ((uint32_t*)a.canvas)[ci] = ((const uint32_t*)a.frame)[fi]; break;
            case OP_COPY_REF:  // blendMulAdd's alpha case copies ref at frameOffset (:396-398)
                ((uint32_t*)a.canvas)[ci] = ((const uint32_t*)a.ref)[(int64_t)(y + a.r.frame_y) * a.rw + (x + a.r.frame_x)];
                break;
            case OP_ADD_I: ((uint32_t*)a.canvas)[ci] = ((const uint32_t*)a.ref)[ri] + ((const uint32_t*)a.frame)[fi]; break;
            case OP_ADD_F: ((float*)a.canvas)[ci] = ((const float*)a.ref)[ri] + ((const float*)a.frame)[fi]; break;
            case OP_MULT: {
                float nw = ((const float*)a.frame)[fi];
                if (a.clamp) nw = clamp_asc(nw, 0.0f, 1.0f);
                ((float*)a.canvas)[ci] = nw * ((const float*)a.ref)[ri];
                break;
            }
            case OP_BLEND: {
                const float oldS = ((const float*)a.ref)[ri];
                const float newS = ((const float*)a.frame)[fi];
                const float oldA = a.is_alpha ? oldS : a.ref_alpha[ri];
                float newA = a.is_alpha ? newS : a.frame_alpha[fi];
                if (a.clamp) newA = clamp_asc(newA, 0.0f, 1.0f);
                float v;
                if (a.is_alpha) v = oldA + newA * (1.0f - oldA);
                else if (a.premult) v = newS + oldS * (1.0f - newA);
                else v = (newS * newA + oldS * oldA * (1.0f - newA)) / (oldA + newA * (1.0f - oldA));
                ((float*)a.canvas)[ci] = v;
                break;
            }
            default: {
                const float oldS = ((const float*)a.ref)[ri];
                const float newS = ((const float*)a.frame)[fi];
                float newA = a.frame_alpha[fi];
                if (a.clamp) newA = clamp_asc(newA, 0.0f, 1.0f);
                ((float*)a.canvas)[ci] = oldS + newA * newS;
            }
        }
    }
}

// ---- orientation (JXLCodestreamDecoder.java:43-177) ------------------------------------------------------
// 1..4 keep the row direction: one thread per element, rows stay contiguous on both sides
__global__ __launch_bounds__(256) void k_orient_flip(const uint32_t* __restrict__ in, int h, int w, int orientation,
                                                     uint32_t* __restrict__ out) {
    const int64_t n = (int64_t)h * w;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int y = (int)(i / w), x = (int)(i - (int64_t)y * w);
        const int dy = (orientation == 3 || orientation == 4) ? h - 1 - y : y;
        const int dx = (orientation == 2 || orientation == 3) ? w - 1 - x : x;
        out[(int64_t)dy * w + dx] = in[i];
    }
}
// 5..8 swap the axes: 32x32 tiles through LDS so that loads run along source rows and stores along destination rows
__global__ __launch_bounds__(256) void k_orient_transpose(const uint32_t* __restrict__ in, int h, int w, int orientation,
                                                          uint32_t* __restrict__ out) {
    __shared__ uint32_t tile[32][33];
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int y = y0 + r, x = x0 + tx;
        if (y < h && x < w) tile[r][tx] = in[(int64_t)y * w + x];
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int x = x0 + r, y = y0 + tx;  // destination row follows the source x, column follows the source y
        if (y >= h || x >= w) continue;
        const int dr = (orientation == 5 || orientation == 6) ? x : w - 1 - x;
        const int dc = (orientation == 5 || orientation == 8) ? y : h - 1 - y;
        out[(int64_t)dr * h + dc] = tile[tx][r];
    }
}

// ---- PNGWriter ctor tail + writeIDAT sample order (PNGWriter.java:79-111, 191-203) -----------------------
struct PackArgs {
    const void* planes[4];
    int64_t n;
    int nch, n_color, premultiplied, coerce, bit_depth, big_endian;
    int is_int[4];
    float cast_scale[4];  // 1f / maxValue(tagged depth) of castToFloat0
};
__global__ __launch_bounds__(256) void k_pack(PackArgs a, void* out) {
    const int maxv = a.bit_depth == 8 ? 255 : 65535;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
        float fv[4];
        int32_t iv[4];
        bool isf[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (c >= a.nch) break;
            isf[c] = !a.is_int[c];
            if (a.is_int[c]) {
                iv[c] = ((const int32_t*)a.planes[c])[i];
                if (a.coerce) {
                    fv[c] = (float)iv[c] * a.cast_scale[c];
                    isf[c] = true;
                }
            } else {
                fv[c] = ((const float*)a.planes[c])[i];
            }
        }
        if (a.premultiplied) {
            const float al = fv[a.n_color];
#pragma unroll
            for (int c = 0; c < 3; c++)
                if (c < a.n_color) fv[c] = fv[c] / al;
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (c >= a.nch) break;
            int v = isf[c] ? f2i_sat(fv[c] * (float)maxv + 0.5f) : iv[c];
            v = v < 0 ? 0 : v > maxv ? maxv : v;
            const int64_t o = i * a.nch + c;
            if (a.bit_depth == 8) ((uint8_t*)out)[o] = (uint8_t)v;
            else ((uint16_t*)out)[o] = a.big_endian ? (uint16_t)(((v & 0xff) << 8) | (v >> 8)) : (uint16_t)v;
        }
    }
}

inline int grid_for(int64_t n, int per_block = 256) {
    return (int)std::min<int64_t>(std::max<int64_t>(1, (n + per_block - 1) / per_block), 256 * 64);
}

}  // namespace

void launch_chroma_upsample_h(const float* in, int h, int w, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_chroma_up_h, dim3(grid_for((int64_t)h * w)), dim3(256), 0, s, in, h, w, out);
}
void launch_chroma_upsample_v(const float* in, int h, int w, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_chroma_up_v, dim3(grid_for((int64_t)h * w)), dim3(256), 0, s, in, h, w, out);
}
void launch_upsample(const float* in, int h, int w, int k, const float* weights, float* out, hipStream_t s) {
    const int g = grid_for((int64_t)h * w * k);
    if (k == 2) hipLaunchKernelGGL(k_upsample<2>, dim3(g), dim3(256), 0, s, in, h, w, weights, out);
    else if (k == 4) hipLaunchKernelGGL(k_upsample<4>, dim3(g), dim3(256), 0, s, in, h, w, weights, out);
    else hipLaunchKernelGGL(k_upsample<8>, dim3(g), dim3(256), 0, s, in, h, w, weights, out);
}
void launch_noise_init(int h, int w, int group_dim, uint64_t seed0, int colors, float* const tmp[3], float* const out[3],
                       hipStream_t s) {
    const int row_stride = (w + group_dim - 1) / group_dim;
    const int num_groups = row_stride * ((h + group_dim - 1) / group_dim);
    hipLaunchKernelGGL(k_noise_rng, dim3((num_groups + 7) / 8), dim3(64), 0, s, h, w, group_dim, seed0, colors, row_stride,
                       num_groups, tmp[0], colors > 1 ? tmp[1] : nullptr, colors > 2 ? tmp[2] : nullptr);
    for (int c = 0; c < colors; c++)
        hipLaunchKernelGGL(k_noise_conv, dim3(grid_for((int64_t)h * w)), dim3(256), 0, s, tmp[c], h, w, out[c]);
}
void launch_noise_add(float* const planes[3], const float* const noise[3], int64_t n, const float lut[8], float bcx, float bcb,
                      hipStream_t s) {
    NoiseLut l;
    for (int i = 0; i < 8; i++) l.v[i] = lut[i];
    hipLaunchKernelGGL(k_noise_add, dim3(grid_for(n)), dim3(256), 0, s, planes[0], planes[1], planes[2], noise[0], noise[1],
                       noise[2], n, l, bcx, bcb);
}

int blend_op(int mode, unsigned flags, int is_int) {
    const bool is_alpha = flags & JXL_BLEND_FLAG_IS_ALPHA, has_extra = flags & JXL_BLEND_FLAG_HAS_EXTRA;
    int op;
    switch (mode) {
        case JXL_BLEND_REPLACE: op = OP_COPY_FRAME; break;
        case JXL_BLEND_ADD: op = is_int ? OP_ADD_I : OP_ADD_F; break;
        case JXL_BLEND_MULT: op = OP_MULT; break;
        case JXL_BLEND_BLEND: op = has_extra ? OP_BLEND : (is_int ? OP_ADD_I : OP_ADD_F); break;  // :346-349
        case JXL_BLEND_MULADD: op = !has_extra ? (is_int ? OP_ADD_I : OP_ADD_F) : is_alpha ? OP_COPY_REF : OP_MULADD; break;
        default: return -1;  // "Illegal blend mode"
    }
    if (is_int && op != OP_COPY_FRAME && op != OP_COPY_REF && op != OP_ADD_I) return -2;
    return op;
}
bool blend_needs(int op, bool* frame, bool* ref, bool* frame_alpha, bool* ref_alpha, bool is_alpha) {
    *frame = op != OP_COPY_REF;
    *ref = op != OP_COPY_FRAME;
    *frame_alpha = (op == OP_BLEND && !is_alpha) || op == OP_MULADD;
    *ref_alpha = op == OP_BLEND && !is_alpha;
    return true;
}
void launch_blend(int op, unsigned flags, void* canvas, int cw, const void* frame, int fw, const void* ref, int rw,
                  const float* frame_alpha, const float* ref_alpha, const jxl_blend_rect& r, hipStream_t s) {
    BlendArgs a{canvas, frame, ref, frame_alpha, ref_alpha, cw, fw, rw, r, op, (flags & JXL_BLEND_FLAG_IS_ALPHA) != 0,
                (flags & JXL_BLEND_FLAG_CLAMP) != 0, (flags & JXL_BLEND_FLAG_PREMULT) != 0};
    if (r.h <= 0 || r.w <= 0) return;
    hipLaunchKernelGGL(k_blend, dim3(grid_for((int64_t)r.h * r.w)), dim3(256), 0, s, a);
}
void launch_orient(const void* in, int h, int w, int orientation, void* out, hipStream_t s) {
    if (orientation <= 4)
        hipLaunchKernelGGL(k_orient_flip, dim3(grid_for((int64_t)h * w)), dim3(256), 0, s, (const uint32_t*)in, h, w, orientation,
                           (uint32_t*)out);
    else
        hipLaunchKernelGGL(k_orient_transpose, dim3((w + 31) / 32, (h + 31) / 32), dim3(256), 0, s, (const uint32_t*)in, h, w,
                           orientation, (uint32_t*)out);
}
void launch_pack(const void* const planes[4], const jxl_pack_params& p, bool coerce, void* out, hipStream_t s) {
    PackArgs a{};
    a.n = (int64_t)p.height * p.width;
    a.n_color = p.n_color;
    a.nch = p.n_color + (p.has_alpha ? 1 : 0);
    a.premultiplied = p.premultiplied;
    a.coerce = coerce;
    a.bit_depth = p.bit_depth;
    a.big_endian = p.big_endian;
    for (int c = 0; c < a.nch; c++) {
        a.planes[c] = planes[c];
        a.is_int[c] = p.is_int[c];
        const int d = p.tagged_depth[c];
        a.cast_scale[c] = (d >= 1 && d <= 31) ? 1.0f / (float)(~(~0 << d)) : 0.0f;  // ImageBuffer.java:119
    }
    if (a.n > 0) hipLaunchKernelGGL(k_pack, dim3(grid_for(a.n)), dim3(256), 0, s, a, out);
}

}  // namespace jxl
