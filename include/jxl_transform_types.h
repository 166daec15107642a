/*
 * Varblock transform-type geometry table: a data restatement of the 27 entries of
 * J/frame/vardct/TransformType.java:10-36 (type id, quant-weight parameter index,
 * coefficient-order id, transform method, pixel height/width). Shared by the HIP
 * library, the oracle and (through jxl_transform_type_info) the Python host.
 */
#ifndef JXL_TRANSFORM_TYPES_H
#define JXL_TRANSFORM_TYPES_H
#include <stdint.h>

/* TransformType.METHOD_* (TransformType.java:47-53) */
#define JXL_METHOD_DCT     0
#define JXL_METHOD_DCT2    1
#define JXL_METHOD_DCT4    2
#define JXL_METHOD_HORNUSS 3
#define JXL_METHOD_DCT8_4  4
#define JXL_METHOD_DCT4_8  5
#define JXL_METHOD_AFV     6

typedef struct jxl_tt_info {
    uint8_t type, param_index, order_id, method;
    uint16_t ph, pw; /* pixelHeight, pixelWidth */
} jxl_tt_info;

#ifdef __cplusplus
static constexpr jxl_tt_info JXL_TT[27] = {
#else
static const jxl_tt_info JXL_TT[27] = {
#endif
    {0, 0, 0, JXL_METHOD_DCT, 8, 8},          /* DCT8 */
    {1, 1, 1, JXL_METHOD_HORNUSS, 8, 8},      /* HORNUSS */
    {2, 2, 1, JXL_METHOD_DCT2, 8, 8},         /* DCT2 */
    {3, 3, 1, JXL_METHOD_DCT4, 8, 8},         /* DCT4 */
    {4, 4, 2, JXL_METHOD_DCT, 16, 16},        /* DCT16 */
    {5, 5, 3, JXL_METHOD_DCT, 32, 32},        /* DCT32 */
    {6, 6, 4, JXL_METHOD_DCT, 16, 8},         /* DCT16_8 */
    {7, 6, 4, JXL_METHOD_DCT, 8, 16},         /* DCT8_16 */
    {8, 7, 5, JXL_METHOD_DCT, 32, 8},         /* DCT32_8 */
    {9, 7, 5, JXL_METHOD_DCT, 8, 32},         /* DCT8_32 */
    {10, 8, 6, JXL_METHOD_DCT, 32, 16},       /* DCT32_16 */
    {11, 8, 6, JXL_METHOD_DCT, 16, 32},       /* DCT16_32 */
    {12, 9, 1, JXL_METHOD_DCT4_8, 8, 8},      /* DCT4_8 */
    {13, 9, 1, JXL_METHOD_DCT8_4, 8, 8},      /* DCT8_4 */
    {14, 10, 1, JXL_METHOD_AFV, 8, 8},        /* AFV0 */
    {15, 10, 1, JXL_METHOD_AFV, 8, 8},        /* AFV1 */
    {16, 10, 1, JXL_METHOD_AFV, 8, 8},        /* AFV2 */
    {17, 10, 1, JXL_METHOD_AFV, 8, 8},        /* AFV3 */
    {18, 11, 7, JXL_METHOD_DCT, 64, 64},      /* DCT64 */
    {19, 12, 8, JXL_METHOD_DCT, 64, 32},      /* DCT64_32 */
    {20, 12, 8, JXL_METHOD_DCT, 32, 64},      /* DCT32_64 */
    {21, 13, 9, JXL_METHOD_DCT, 128, 128},    /* DCT128 */
    {22, 14, 10, JXL_METHOD_DCT, 128, 64},    /* DCT128_64 */
    {23, 14, 10, JXL_METHOD_DCT, 64, 128},    /* DCT64_128 */
    {24, 15, 11, JXL_METHOD_DCT, 256, 256},   /* DCT256 */
    {25, 16, 12, JXL_METHOD_DCT, 256, 128},   /* DCT256_128 */
    {26, 16, 12, JXL_METHOD_DCT, 128, 256},   /* DCT128_256 */
};

/* TransformType.flip() (TransformType.java:129-131) */
static inline int jxl_tt_flip(const jxl_tt_info* t) {
    return t->ph > t->pw || (t->method == JXL_METHOD_DCT && t->ph == t->pw);
}
/* matrixHeight/matrixWidth = min/max of the pixel size (TransformType.java:151-152) */
static inline int jxl_tt_mh(const jxl_tt_info* t) { return t->ph < t->pw ? t->ph : t->pw; }
static inline int jxl_tt_mw(const jxl_tt_info* t) { return t->ph < t->pw ? t->pw : t->ph; }

#endif
