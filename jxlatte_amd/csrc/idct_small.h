// Register-array forms of the 8x8-footprint transforms of PassGroup.invertVarDCT (J/frame/group/PassGroup.java:88-168, 229-328) and of
// MathHelper.inverseDCT2D for 2..8 points (J/util/MathHelper.java:68-122): a lane holds a whole block of one channel. Shared by
// k_idct.hip (k_idct_special_wg, the per-channel kernels) and, since r6, by the special-type items of the persistent launch in
// k_idct_wg3.hip. Included INSIDE namespace jxl of a translation unit that has included jxl_tables.h (JXL_AFV_BASIS_INIT).
#pragma once
#include "lut_small.inc"

static constexpr float kAfv[16][16] = JXL_AFV_BASIS_INIT;

template <int N>
__device__ __forceinline__ constexpr float lutc(int n, int k) {
    return N == 2 ? kLut2[n][k & 1] : N == 4 ? kLut4[n % 3][k & 3] : kLut8[n % 7][k & 7];
}

// MathHelper.inverseDCTHorizontal (MathHelper.java:68-78) on register arrays with compile-time strides
// The float32 table is exactly mirror-(anti)symmetric, lut[n-1][N-1-k] == (-1)^n * lut[n-1][k] bit for bit (checked for
// every N = 2..256 in tests/test_oracle_kats.py), and x * (-c) == -(x * c), a + (-p) == a - p exactly in IEEE arithmetic:
// each product is formed once and added to output k and added / subtracted to output N-1-k. Same bits, half the multiplies.
template <int N, int SS, int DS>
__device__ __forceinline__ void idct1d_reg(const float* s, float* d) {
    const float s0 = s[0];
#pragma unroll
    for (int k = 0; k < N; k++) d[k * DS] = s0;
#pragma unroll
    for (int n = 1; n < N; n++) {
        const float s2 = s[n * SS];
#pragma unroll
        for (int k = 0; k < N / 2; k++) {
            const float p = s2 * lutc<N>(n - 1, k);
            d[k * DS] = d[k * DS] + p;
            d[(N - 1 - k) * DS] = (n & 1) ? d[(N - 1 - k) * DS] - p : d[(N - 1 - k) * DS] + p;
        }
    }
}

// MathHelper.inverseDCT2D (MathHelper.java:96-122). src is H x W (row stride SS).
// TRANSPOSED=false: dst is H x W; true: dst is W rows x H columns. Row stride of dst = DS.
template <int H, int W, bool TRANSPOSED, int SS, int DS>
__device__ __forceinline__ void idct2d_reg(const float* src, float* dst) {
    float t[H * W];
    if (TRANSPOSED) {
#pragma unroll
        for (int y = 0; y < H; y++) idct1d_reg<W, 1, 1>(src + y * SS, t + y * W);  // rows, length W
#pragma unroll
        for (int x = 0; x < W; x++) idct1d_reg<H, W, 1>(t + x, dst + x * DS);      // columns of t -> row x of dst
    } else {
#pragma unroll
        for (int x = 0; x < W; x++) idct1d_reg<H, SS, W>(src + x, t + x);          // columns, length H
#pragma unroll
        for (int y = 0; y < H; y++) idct1d_reg<W, 1, 1>(t + y * W, dst + y * DS);  // rows, length W
    }
}

// PassGroup.auxDCT2 (PassGroup.java:149-168) on 8x8 register blocks
template <int S>
__device__ __forceinline__ void aux_dct2_reg(const float* in, float* out) {
#pragma unroll
    for (int i = 0; i < 64; i++) out[i] = in[i];
    constexpr int num = S / 2;
#pragma unroll
    for (int iy = 0; iy < num; iy++) {
#pragma unroll
        for (int ix = 0; ix < num; ix++) {
            const float c00 = in[iy * 8 + ix];
            const float c01 = in[iy * 8 + ix + num];
            const float c10 = in[(iy + num) * 8 + ix];
            const float c11 = in[(iy + num) * 8 + ix + num];
            out[(iy * 2) * 8 + ix * 2] = c00 + c01 + c10 + c11;
            out[(iy * 2) * 8 + ix * 2 + 1] = c00 + c01 - c10 - c11;
            out[(iy * 2 + 1) * 8 + ix * 2] = c00 - c01 + c10 - c11;
            out[(iy * 2 + 1) * 8 + ix * 2 + 1] = c00 - c01 - c10 + c11;
        }
    }
}

// the per-type switch of PassGroup.invertVarDCT (PassGroup.java:229-328) for the 8x8-footprint
// types; co = dequantised coefficients (8x8, stride 8), px = pixels (8x8, stride 8)
template <int TYPE>
__device__ __forceinline__ void invert_small(const float* co, float* px) {
    if (TYPE == 0) {  // DCT8
        idct2d_reg<8, 8, false, 8, 8>(co, px);
    } else if (TYPE == 13) {  // DCT8_4 (:234-251)
        const float coeff0 = co[0], coeff1 = co[8];
        const float lfs[2] = {coeff0 + coeff1, coeff0 - coeff1};
#pragma unroll
        for (int x = 0; x < 2; x++) {
            float s[32];
#pragma unroll
            for (int iy = 0; iy < 4; iy++)
#pragma unroll
                for (int ix = 0; ix < 8; ix++) s[iy * 8 + ix] = co[(x + iy * 2) * 8 + ix];
            s[0] = lfs[x];
            idct2d_reg<4, 8, true, 8, 8>(s, px + (x << 2));
        }
    } else if (TYPE == 12) {  // DCT4_8 (:252-269)
        const float coeff0 = co[0], coeff1 = co[8];
        const float lfs[2] = {coeff0 + coeff1, coeff0 - coeff1};
#pragma unroll
        for (int y = 0; y < 2; y++) {
            float s[32];
#pragma unroll
            for (int iy = 0; iy < 4; iy++)
#pragma unroll
                for (int ix = 0; ix < 8; ix++) s[iy * 8 + ix] = co[(y + iy * 2) * 8 + ix];
            s[0] = lfs[y];
            idct2d_reg<4, 8, false, 8, 8>(s, px + (y << 2) * 8);
        }
    } else if (TYPE >= 14 && TYPE <= 17) {  // AFV0..3: PassGroup.invertAFV (:88-147)
        constexpr int flipY = (TYPE == 16 || TYPE == 17) ? 1 : 0;
        constexpr int flipX = (TYPE == 15 || TYPE == 17) ? 1 : 0;
        float s0[16], s1[16];
#pragma unroll
        for (int iy = 0; iy < 4; iy++)
#pragma unroll
            for (int ix = 0; ix < 4; ix++) s0[iy * 4 + ix] = co[(iy * 2) * 8 + ix * 2];
        s0[0] = (co[0] + co[8] + co[1]) * 4.0f;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            float sample = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; j++) sample = sample + s0[j] * kAfv[j][i];
            s1[i] = sample;
        }
#pragma unroll
        for (int iy = 0; iy < 4; iy++)
#pragma unroll
            for (int ix = 0; ix < 4; ix++)
                px[(flipY * 4 + iy) * 8 + flipX * 4 + ix] = s1[(flipY ? 3 - iy : iy) * 4 + (flipX ? 3 - ix : ix)];
#pragma unroll
        for (int iy = 0; iy < 4; iy++)
#pragma unroll
            for (int ix = 0; ix < 4; ix++) s0[iy * 4 + ix] = co[(iy * 2) * 8 + ix * 2 + 1];
        s0[0] = co[0] + co[8] - co[1];
        idct2d_reg<4, 4, false, 4, 4>(s0, s1);
#pragma unroll
        for (int iy = 0; iy < 4; iy++)
#pragma unroll
            for (int ix = 0; ix < 4; ix++)  // transposed intentionally (:129-131)
                px[(flipY * 4 + iy) * 8 + (flipX ? 0 : 4) + ix] = s1[ix * 4 + iy];
        float r0[32], r1[32];
#pragma unroll
        for (int iy = 0; iy < 4; iy++)
#pragma unroll
            for (int ix = 0; ix < 8; ix++) r0[iy * 8 + ix] = co[(1 + iy * 2) * 8 + ix];
        r0[0] = co[0] - co[8];
        idct2d_reg<4, 8, false, 8, 8>(r0, r1);
#pragma unroll
        for (int iy = 0; iy < 4; iy++)
#pragma unroll
            for (int ix = 0; ix < 8; ix++) px[((flipY ? 0 : 4) + iy) * 8 + ix] = r1[iy * 8 + ix];
    } else if (TYPE == 2) {  // DCT2 (:273-277)
        float a[64], b[64];
        aux_dct2_reg<2>(co, a);
        aux_dct2_reg<4>(a, b);
        aux_dct2_reg<8>(b, px);
    } else if (TYPE == 1) {  // HORNUSS (:278-305)
        // auxDCT2(coeffs, s1, 2): only s1[y][x], y,x < 2 are consumed
        const float c00 = co[0], c01 = co[1], c10 = co[8], c11 = co[9];
        const float lf4[4] = {c00 + c01 + c10 + c11, c00 + c01 - c10 - c11, c00 - c01 + c10 - c11, c00 - c01 - c10 + c11};
#pragma unroll
        for (int y = 0; y < 2; y++) {
#pragma unroll
            for (int x = 0; x < 2; x++) {
                const float blockLF = lf4[y * 2 + x];
                float residual = 0.0f;
#pragma unroll
                for (int iy = 0; iy < 4; iy++)
#pragma unroll
                    for (int ix = (iy == 0 ? 1 : 0); ix < 4; ix++) residual = residual + co[(y + iy * 2) * 8 + x + ix * 2];
                const float a = blockLF - residual * 0.0625f;
                px[(4 * y + 1) * 8 + 4 * x + 1] = a;
#pragma unroll
                for (int iy = 0; iy < 4; iy++)
#pragma unroll
                    for (int ix = 0; ix < 4; ix++) {
                        if (ix == 1 && iy == 1) continue;
                        px[(y * 4 + iy) * 8 + x * 4 + ix] = co[(y + iy * 2) * 8 + x + ix * 2] + a;
                    }
                px[(4 * y) * 8 + 4 * x] = co[(y + 2) * 8 + x + 2] + a;
            }
        }
    } else if (TYPE == 3) {  // DCT4 (:306-325)
        const float c00 = co[0], c01 = co[1], c10 = co[8], c11 = co[9];
        const float lf4[4] = {c00 + c01 + c10 + c11, c00 + c01 - c10 - c11, c00 - c01 + c10 - c11, c00 - c01 - c10 + c11};
#pragma unroll
        for (int y = 0; y < 2; y++) {
#pragma unroll
            for (int x = 0; x < 2; x++) {
                float s[16], t[16];
#pragma unroll
                for (int iy = 0; iy < 4; iy++)
#pragma unroll
                    for (int ix = 0; ix < 4; ix++) s[iy * 4 + ix] = co[(y + iy * 2) * 8 + x + ix * 2];
                s[0] = lf4[y * 2 + x];
                idct2d_reg<4, 4, true, 4, 4>(s, t);
#pragma unroll
                for (int iy = 0; iy < 4; iy++)
#pragma unroll
                    for (int ix = 0; ix < 4; ix++) px[(4 * y + iy) * 8 + 4 * x + ix] = t[iy * 4 + ix];
            }
        }
    }
}

