/*
 * JNI glue between com.traneptora.jxlatte.gpu.NativeBackend and include/jxlatte_amd.h (row f4).
 *
 * NOT COMPILED OR TESTED HERE: this image has no JDK / jni.h. Build on a machine that has one:
 *   cc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include jxlatte_amd_jni.c \
 *      -L../../jxlatte_amd -ljxlatte_amd -o libjxlatte_amd_jni.so
 * Status codes are rethrown as the exceptions the reference throws at the same places.
 */
#include <jni.h>
#include <stdint.h>
#include <string.h>

#include "jxlatte_amd.h"

static jxl_ctx* ctx_of(JNIEnv* e, jobject self) {
    jclass cls = (*e)->GetObjectClass(e, self);
    jfieldID f = (*e)->GetFieldID(e, cls, "ctx", "J");
    return (jxl_ctx*)(intptr_t)(*e)->GetLongField(e, self, f);
}

static void rethrow(JNIEnv* e, jxl_ctx* c, jxl_status st) {
    const char* cls = st == JXL_ERR_INVALID_BITSTREAM ? "com/traneptora/jxlatte/io/InvalidBitstreamException"
                    : st == JXL_ERR_UNSUPPORTED       ? "java/lang/UnsupportedOperationException"
                    : st == JXL_ERR_INVALID_ARGUMENT  ? "java/lang/IllegalArgumentException"
                    : st == JXL_ERR_STATE             ? "java/lang/IllegalStateException"
                    : st == JXL_ERR_OOM               ? "java/lang/OutOfMemoryError"
                                                      : "java/lang/RuntimeException";
    (*e)->ThrowNew(e, (*e)->FindClass(e, cls), c ? jxl_last_error(c) : "jxlatte_amd: no context");
}
#define CHECK(call)                         \
    do {                                    \
        jxl_status st_ = (call);            \
        if (st_ != JXL_OK) {                \
            rethrow(e, c, st_);             \
            return;                         \
        }                                   \
    } while (0)
#define ADDR(buf) ((buf) ? (*e)->GetDirectBufferAddress(e, (buf)) : NULL)

JNIEXPORT jlong JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_create(JNIEnv* e, jclass k, jint device) {
    (void)k;
    jxl_ctx* c = NULL;
    jxl_status st = jxl_ctx_create(device, &c);
    if (st != JXL_OK) {
        rethrow(e, c, st);
        return 0;
    }
    return (jlong)(intptr_t)c;
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_destroy(JNIEnv* e, jclass k, jlong ctx) {
    (void)e; (void)k;
    jxl_ctx_destroy((jxl_ctx*)(intptr_t)ctx);
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_beginFrame(JNIEnv* e, jobject self, jobject params) {
    jxl_ctx* c = ctx_of(e, self);
    jxl_vardct_params p;
    if ((size_t)(*e)->GetDirectBufferCapacity(e, params) < sizeof p) {
        rethrow(e, c, JXL_ERR_INVALID_ARGUMENT);
        return;
    }
    memcpy(&p, ADDR(params), sizeof p);
    CHECK(jxl_vardct_begin_frame(c, &p));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_setWeights(JNIEnv* e, jobject self, jobject weights, jintArray offs) {
    jxl_ctx* c = ctx_of(e, self);
    jint o[51];
    (*e)->GetIntArrayRegion(e, offs, 0, 51, o);
    CHECK(jxl_vardct_set_weights(c, (const float*)ADDR(weights), (size_t)(*e)->GetDirectBufferCapacity(e, weights) / 4, (const int32_t*)o));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_setLFGroup(JNIEnv* e, jobject self, jint lfgY, jint lfgX, jint cellsH,
        jint cellsW, jobject dctSelect, jobject hfMul, jobject sharpness, jobject xFromY, jobject bFromY, jobject blockYX, jint nBlocks,
        jobject lfX, jobject lfY, jobject lfB) {
    jxl_ctx* c = ctx_of(e, self);
    jxl_lfgroup_desc d;
    memset(&d, 0, sizeof d);
    d.lfg_y = lfgY; d.lfg_x = lfgX; d.cells_h = cellsH; d.cells_w = cellsW;
    d.dct_select = (const uint8_t*)ADDR(dctSelect);
    d.hf_mul = (const int32_t*)ADDR(hfMul);
    d.sharpness = (const int32_t*)ADDR(sharpness);
    d.x_from_y = (const int32_t*)ADDR(xFromY);
    d.b_from_y = (const int32_t*)ADDR(bFromY);
    d.block_yx = (const int32_t*)ADDR(blockYX);
    d.n_blocks = nBlocks;
    d.lf[0] = (const float*)ADDR(lfX); d.lf[1] = (const float*)ADDR(lfY); d.lf[2] = (const float*)ADDR(lfB);
    CHECK(jxl_vardct_set_lfgroup(c, &d));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_setLFGroupQuant(JNIEnv* e, jobject self, jint lfgY, jint lfgX,
        jint cellsH, jint cellsW, jobject qX, jobject qY, jobject qB, jint extraPrecision, jfloatArray scaledDequant, jint xFactorLF,
        jint bFactorLF, jboolean adaptiveSmoothing) {
    jxl_ctx* c = ctx_of(e, self);
    jxl_lfquant_desc d;
    memset(&d, 0, sizeof d);
    d.lfg_y = lfgY; d.lfg_x = lfgX; d.cells_h = cellsH; d.cells_w = cellsW;
    d.lf_quant[0] = (const int32_t*)ADDR(qX); d.lf_quant[1] = (const int32_t*)ADDR(qY); d.lf_quant[2] = (const int32_t*)ADDR(qB);
    d.extra_precision = extraPrecision;
    (*e)->GetFloatArrayRegion(e, scaledDequant, 0, 3, d.scaled_dequant);
    d.x_factor_lf = xFactorLF; d.b_factor_lf = bFactorLF; d.adaptive_smoothing = adaptiveSmoothing ? 1 : 0;
    CHECK(jxl_vardct_set_lfgroup_lfquant(c, &d));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_putGroup(JNIEnv* e, jobject self, jint pass, jint group, jobject qx,
        jobject qy, jobject qb, jint sx, jint sy, jint sb) {
    jxl_ctx* c = ctx_of(e, self);
    const int32_t* q[3] = {(const int32_t*)ADDR(qx), (const int32_t*)ADDR(qy), (const int32_t*)ADDR(qb)};
    const int32_t s[3] = {sx, sy, sb};
    CHECK(jxl_vardct_put_group(c, pass, group, q, s));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_putGroupI16(JNIEnv* e, jobject self, jint pass, jint group, jobject qx,
        jobject qy, jobject qb, jint sx, jint sy, jint sb) {
    jxl_ctx* c = ctx_of(e, self);
    const int16_t* q[3] = {(const int16_t*)ADDR(qx), (const int16_t*)ADDR(qy), (const int16_t*)ADDR(qb)};
    const int32_t s[3] = {sx, sy, sb};
    CHECK(jxl_vardct_put_group_i16(c, pass, group, q, s));
}

/* planes of the current frame (sizes from the params the caller passed to beginFrame: (H >> sy) * (W >> sx) samples each) */
JNIEXPORT jobjectArray JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_mapCoeffsI16(JNIEnv* e, jobject self, jint rx, jint ry, jint rb) {
    jxl_ctx* c = ctx_of(e, self);
    int16_t* pl[3];
    int32_t st[3];
    const jint h[3] = {rx, ry, rb};
    jxl_status r = jxl_vardct_map_coeffs_i16(c, pl, st);
    if (r) { rethrow(e, c, r); return NULL; }
    jobjectArray out = (*e)->NewObjectArray(e, 3, (*e)->FindClass(e, "java/nio/ByteBuffer"), NULL);
    for (int i = 0; i < 3; i++)
        (*e)->SetObjectArrayElement(e, out, i, (*e)->NewDirectByteBuffer(e, pl[i], (jlong)st[i] * h[i] * 2));
    return out;
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_commitCoeffsI16(JNIEnv* e, jobject self) {
    jxl_ctx* c = ctx_of(e, self);
    CHECK(jxl_vardct_commit_coeffs_i16(c));
}

JNIEXPORT jobject JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_hostAlloc(JNIEnv* e, jclass k, jlong bytes) {
    void* p = jxl_host_alloc((size_t)bytes);
    return p ? (*e)->NewDirectByteBuffer(e, p, bytes) : NULL;
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_hostFree(JNIEnv* e, jclass k, jobject b) {
    if (b) jxl_host_free((*e)->GetDirectBufferAddress(e, b));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_finishFrame(JNIEnv* e, jobject self, jobject ox, jobject oy, jobject ob,
        jlong stride) {
    jxl_ctx* c = ctx_of(e, self);
    void* out[3] = {ADDR(ox), ADDR(oy), ADDR(ob)};
    CHECK(jxl_vardct_finish_frame(c, out, stride));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_run(JNIEnv* e, jobject self) {
    jxl_ctx* c = ctx_of(e, self);
    CHECK(jxl_vardct_run(c));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_readOutput(JNIEnv* e, jobject self, jobject ox, jobject oy, jobject ob,
        jlong stride) {
    jxl_ctx* c = ctx_of(e, self);
    void* out[3] = {ADDR(ox), ADDR(oy), ADDR(ob)};
    CHECK(jxl_vardct_read_output(c, out, stride));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_prepare(JNIEnv* e, jobject self) {
    jxl_ctx* c = ctx_of(e, self);
    CHECK(jxl_vardct_prepare(c));
}

/* ---- resident colour planes (include/jxlatte_amd.h: jxl_planes_*) ---- */
JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_planesFromFrame(JNIEnv* e, jobject self, jint h, jint w) {
    jxl_ctx* c = ctx_of(e, self);
    CHECK(jxl_planes_from_frame(c, h, w));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_planesUpload(JNIEnv* e, jobject self, jobject p0, jobject p1, jobject p2,
        jint h, jint w) {
    jxl_ctx* c = ctx_of(e, self);
    const float* in[3] = {(const float*)ADDR(p0), (const float*)ADDR(p1), (const float*)ADDR(p2)};
    CHECK(jxl_planes_upload(c, in, h, w));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_planesUpsample(JNIEnv* e, jobject self, jint k, jfloatArray weights) {
    jxl_ctx* c = ctx_of(e, self);
    jfloat* w = (*e)->GetFloatArrayElements(e, weights, NULL);  /* k*k*25 floats: jxl_upsampling_weights */
    const jxl_status st = jxl_planes_upsample(c, k, w);
    (*e)->ReleaseFloatArrayElements(e, weights, w, JNI_ABORT);
    CHECK(st);
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_planesNoise(JNIEnv* e, jobject self, jint groupDim, jlong seed0,
        jfloatArray lut, jfloat bcx, jfloat bcb) {
    jxl_ctx* c = ctx_of(e, self);
    float l[8];
    (*e)->GetFloatArrayRegion(e, lut, 0, 8, l);
    CHECK(jxl_planes_noise(c, groupDim, (uint64_t)seed0, l, bcx, bcb));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_planesXYB(JNIEnv* e, jobject self, jfloatArray matrix, jfloatArray bias,
        jfloatArray cbrtBias, jfloat intensityTarget) {
    jxl_ctx* c = ctx_of(e, self);
    float m[9], b[3], cb[3];
    (*e)->GetFloatArrayRegion(e, matrix, 0, 9, m);
    (*e)->GetFloatArrayRegion(e, bias, 0, 3, b);
    (*e)->GetFloatArrayRegion(e, cbrtBias, 0, 3, cb);
    CHECK(jxl_planes_xyb(c, m, b, cb, intensityTarget));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_planesYCbCr(JNIEnv* e, jobject self) {
    jxl_ctx* c = ctx_of(e, self);
    CHECK(jxl_planes_ycbcr(c));
}

JNIEXPORT jintArray JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_planesShape(JNIEnv* e, jobject self) {
    jxl_ctx* c = ctx_of(e, self);
    int32_t hw[2] = {0, 0};
    jintArray out = (*e)->NewIntArray(e, 2);
    if (jxl_planes_shape(c, &hw[0], &hw[1]) == JXL_OK && out) (*e)->SetIntArrayRegion(e, out, 0, 2, (const jint*)hw);
    return out;
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_planesDownload(JNIEnv* e, jobject self, jobject p0, jobject p1, jobject p2) {
    jxl_ctx* c = ctx_of(e, self);
    float* out[3] = {(float*)ADDR(p0), (float*)ADDR(p1), (float*)ADDR(p2)};
    CHECK(jxl_planes_download(c, out));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_runBatch0(JNIEnv* e, jclass k, jlongArray ctxs) {
    const jsize n = (*e)->GetArrayLength(e, ctxs);
    jlong h[64];
    jxl_ctx* c[64];
    if (n <= 0 || n > 64) {
        rethrow(e, NULL, JXL_ERR_INVALID_ARGUMENT);
        return;
    }
    (*e)->GetLongArrayRegion(e, ctxs, 0, n, h);
    for (jsize i = 0; i < n; i++) c[i] = (jxl_ctx*)(intptr_t)h[i];
    const jxl_status st = jxl_vardct_run_batch(c, (int32_t)n);
    if (st != JXL_OK) rethrow(e, c[0], st);
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_modularApply(JNIEnv* e, jobject self, jobjectArray chans, jintArray widths,
        jintArray heights, jintArray squeezeParams, jint rctType, jint rctBegin, jobjectArray out, jintArray outWidths, jintArray outHeights) {
    jxl_ctx* c = ctx_of(e, self);
    const jsize n = (*e)->GetArrayLength(e, chans), n_out = (*e)->GetArrayLength(e, out);
    const jsize n_sp = (*e)->GetArrayLength(e, squeezeParams) / 4;
    if (n > 256 || n_out > 256 || n_sp > 64) {
        rethrow(e, c, JXL_ERR_INVALID_ARGUMENT);
        return;
    }
    jxl_channel ci[256], co[256];
    jxl_squeeze_param sp[64];
    jint w[256], h[256], ow[256], oh[256], spv[256];
    (*e)->GetIntArrayRegion(e, widths, 0, n, w);
    (*e)->GetIntArrayRegion(e, heights, 0, n, h);
    (*e)->GetIntArrayRegion(e, outWidths, 0, n_out, ow);
    (*e)->GetIntArrayRegion(e, outHeights, 0, n_out, oh);
    (*e)->GetIntArrayRegion(e, squeezeParams, 0, n_sp * 4, spv);
    for (jsize i = 0; i < n; i++) {
        ci[i].width = w[i]; ci[i].height = h[i];
        ci[i].data = (int32_t*)ADDR((*e)->GetObjectArrayElement(e, chans, i));
    }
    for (jsize i = 0; i < n_out; i++) {
        co[i].width = ow[i]; co[i].height = oh[i];
        co[i].data = (int32_t*)ADDR((*e)->GetObjectArrayElement(e, out, i));
    }
    for (jsize i = 0; i < n_sp; i++) {
        sp[i].horizontal = spv[4 * i]; sp[i].in_place = spv[4 * i + 1]; sp[i].begin_c = spv[4 * i + 2]; sp[i].num_c = spv[4 * i + 3];
    }
    CHECK(jxl_modular_apply(c, ci, n, sp, n_sp, rctType, rctBegin, co, n_out));
}
