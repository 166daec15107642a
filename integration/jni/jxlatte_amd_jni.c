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

/* Argument checks (r4): the library trusts the sizes its C callers state; a Java caller states them twice -- as integers and as
 * the capacity of the direct buffers it passes -- and the shim makes the two agree before anything is read or written. */
static void bad_arg(JNIEnv* e, const char* what) {
    if (!(*e)->ExceptionCheck(e)) (*e)->ThrowNew(e, (*e)->FindClass(e, "java/lang/IllegalArgumentException"), what);
}
/* a direct buffer of at least `bytes` bytes */
static int has_room(JNIEnv* e, jobject buf, jlong bytes) {
    return buf && bytes >= 0 && (*e)->GetDirectBufferAddress(e, buf) && (*e)->GetDirectBufferCapacity(e, buf) >= bytes;
}
#define NEED(buf, bytes)                                                        \
    do {                                                                        \
        if (!has_room(e, (buf), (jlong)(bytes))) {                              \
            bad_arg(e, "jxlatte_amd: direct buffer " #buf " missing or too small"); \
            return;                                                             \
        }                                                                       \
    } while (0)
/* an optional buffer: null, or large enough */
#define NEED_OPT(buf, bytes)          \
    do {                              \
        if (buf) NEED(buf, bytes);    \
    } while (0)
static jlong area(jlong h, jlong w) { return h < 0 || w < 0 ? -1 : h * w; }
/* n floats / ints of a Java array into dst; 0 (exception pending) if the array is null or shorter */
static int get_floats(JNIEnv* e, jfloatArray a, jsize n, float* dst) {
    if (!a || (*e)->GetArrayLength(e, a) < n) {
        bad_arg(e, "jxlatte_amd: float array missing or too short");
        return 0;
    }
    (*e)->GetFloatArrayRegion(e, a, 0, n, dst);
    return !(*e)->ExceptionCheck(e);
}
static int get_ints(JNIEnv* e, jintArray a, jsize n, jint* dst) {
    if (!a || (*e)->GetArrayLength(e, a) < n) {
        bad_arg(e, "jxlatte_amd: int array missing or too short");
        return 0;
    }
    (*e)->GetIntArrayRegion(e, a, 0, n, dst);
    return !(*e)->ExceptionCheck(e);
}
#define GETF(arr, n, dst)                              \
    do {                                               \
        if (!get_floats(e, (arr), (n), (dst))) return; \
    } while (0)
#define GETI(arr, n, dst)                            \
    do {                                             \
        if (!get_ints(e, (arr), (n), (dst))) return; \
    } while (0)

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
    /* (a heap ByteBuffer has no direct address: GetDirectBufferAddress returns NULL for it) */
    const void* src = params ? (*e)->GetDirectBufferAddress(e, params) : NULL;
    if (!src || (*e)->GetDirectBufferCapacity(e, params) < (jlong)sizeof p) {
        bad_arg(e, "jxlatte_amd: beginFrame needs a direct buffer holding jxl_vardct_params");
        return;
    }
    memcpy(&p, src, sizeof p);
    CHECK(jxl_vardct_begin_frame(c, &p));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_setWeights(JNIEnv* e, jobject self, jobject weights, jintArray offs) {
    jxl_ctx* c = ctx_of(e, self);
    jint o[51];
    GETI(offs, 51, o);
    NEED(weights, 4);
    CHECK(jxl_vardct_set_weights(c, (const float*)ADDR(weights), (size_t)(*e)->GetDirectBufferCapacity(e, weights) / 4, (const int32_t*)o));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_setLFGroup(JNIEnv* e, jobject self, jint lfgY, jint lfgX, jint cellsH,
        jint cellsW, jobject dctSelect, jobject hfMul, jobject sharpness, jobject xFromY, jobject bFromY, jobject blockYX, jint nBlocks,
        jobject lfX, jobject lfY, jobject lfB) {
    jxl_ctx* c = ctx_of(e, self);
    jxl_lfgroup_desc d;
    int32_t geo[13];
    CHECK(jxl_vardct_geometry(c, geo));
    if (cellsH <= 0 || cellsW <= 0 || cellsH > 256 || cellsW > 256 || nBlocks < 0) {
        bad_arg(e, "jxlatte_amd: LF group size out of range");
        return;
    }
    {   /* every grid the library will read, sized from the stated cell counts and the frame's own subsampling shifts */
        const jlong cells = area(cellsH, cellsW), tiles = area((cellsH + 7) / 8, (cellsW + 7) / 8);
        NEED(dctSelect, cells);
        NEED(hfMul, 4 * cells);
        NEED(sharpness, 4 * cells);
        NEED(xFromY, 4 * tiles);
        NEED(bFromY, 4 * tiles);
        NEED(blockYX, 8 * (jlong)nBlocks);
        NEED_OPT(lfX, 4 * area(cellsH >> geo[7], cellsW >> geo[6]));
        NEED_OPT(lfY, 4 * area(cellsH >> geo[9], cellsW >> geo[8]));
        NEED_OPT(lfB, 4 * area(cellsH >> geo[11], cellsW >> geo[10]));
    }
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
    int32_t geo[13];
    CHECK(jxl_vardct_geometry(c, geo));
    if (cellsH <= 0 || cellsW <= 0 || cellsH > 256 || cellsW > 256) {
        bad_arg(e, "jxlatte_amd: LF group size out of range");
        return;
    }
    NEED(qX, 4 * area(cellsH >> geo[7], cellsW >> geo[6]));
    NEED(qY, 4 * area(cellsH >> geo[9], cellsW >> geo[8]));
    NEED(qB, 4 * area(cellsH >> geo[11], cellsW >> geo[10]));
    memset(&d, 0, sizeof d);
    d.lfg_y = lfgY; d.lfg_x = lfgX; d.cells_h = cellsH; d.cells_w = cellsW;
    d.lf_quant[0] = (const int32_t*)ADDR(qX); d.lf_quant[1] = (const int32_t*)ADDR(qY); d.lf_quant[2] = (const int32_t*)ADDR(qB);
    d.extra_precision = extraPrecision;
    GETF(scaledDequant, 3, d.scaled_dequant);
    d.x_factor_lf = xFactorLF; d.b_factor_lf = bFactorLF; d.adaptive_smoothing = adaptiveSmoothing ? 1 : 0;
    CHECK(jxl_vardct_set_lfgroup_lfquant(c, &d));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_putGroup(JNIEnv* e, jobject self, jint pass, jint group, jobject qx,
        jobject qy, jobject qb, jint sx, jint sy, jint sb) {
    jxl_ctx* c = ctx_of(e, self);
    const int32_t s[3] = {sx, sy, sb};
    int32_t gw[3], gh[3];
    CHECK(jxl_vardct_group_size(c, group, gw, gh));
    {   /* rows of gw samples at the caller's stride: (gh - 1) * stride + gw samples are read from each buffer */
        jobject b[3] = {qx, qy, qb};
        for (int ch = 0; ch < 3; ch++) {
            if (s[ch] < gw[ch]) {
                bad_arg(e, "jxlatte_amd: putGroup stride shorter than the group's rows");
                return;
            }
            if (!has_room(e, b[ch], (jlong)sizeof(int32_t) * (gh[ch] > 0 ? (jlong)(gh[ch] - 1) * s[ch] + gw[ch] : 0))) {
                bad_arg(e, "jxlatte_amd: putGroup coefficient buffer missing or too small for the group");
                return;
            }
        }
    }
    const int32_t* q[3] = {(const int32_t*)ADDR(qx), (const int32_t*)ADDR(qy), (const int32_t*)ADDR(qb)};
    CHECK(jxl_vardct_put_group(c, pass, group, q, s));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_putGroupI16(JNIEnv* e, jobject self, jint pass, jint group, jobject qx,
        jobject qy, jobject qb, jint sx, jint sy, jint sb) {
    jxl_ctx* c = ctx_of(e, self);
    const int32_t s[3] = {sx, sy, sb};
    int32_t gw[3], gh[3];
    CHECK(jxl_vardct_group_size(c, group, gw, gh));
    {   /* rows of gw samples at the caller's stride: (gh - 1) * stride + gw samples are read from each buffer */
        jobject b[3] = {qx, qy, qb};
        for (int ch = 0; ch < 3; ch++) {
            if (s[ch] < gw[ch]) {
                bad_arg(e, "jxlatte_amd: putGroupI16 stride shorter than the group's rows");
                return;
            }
            if (!has_room(e, b[ch], (jlong)sizeof(int16_t) * (gh[ch] > 0 ? (jlong)(gh[ch] - 1) * s[ch] + gw[ch] : 0))) {
                bad_arg(e, "jxlatte_amd: putGroupI16 coefficient buffer missing or too small for the group");
                return;
            }
        }
    }
    const int16_t* q[3] = {(const int16_t*)ADDR(qx), (const int16_t*)ADDR(qy), (const int16_t*)ADDR(qb)};
    CHECK(jxl_vardct_put_group_i16(c, pass, group, q, s));
}

/* planes of the current frame: (H >> sy) rows of (W >> sx) samples each; the sizes come from the library
 * (jxl_vardct_coeff_plane_rows), never from the caller */
JNIEXPORT jobjectArray JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_mapCoeffsI16(JNIEnv* e, jobject self) {
    jxl_ctx* c = ctx_of(e, self);
    int16_t* pl[3];
    int32_t st[3], rows[3];
    jxl_status r = jxl_vardct_map_coeffs_i16(c, pl, st);
    if (r == JXL_OK) r = jxl_vardct_coeff_plane_rows(c, rows);
    if (r) { rethrow(e, c, r); return NULL; }
    jclass bb = (*e)->FindClass(e, "java/nio/ByteBuffer");
    if (!bb) return NULL;  /* exception pending */
    jobjectArray out = (*e)->NewObjectArray(e, 3, bb, NULL);
    if (!out) return NULL;
    for (int i = 0; i < 3; i++) {
        jobject b = (*e)->NewDirectByteBuffer(e, pl[i], (jlong)st[i] * rows[i] * 2);
        if (!b || (*e)->ExceptionCheck(e)) return NULL;
        (*e)->SetObjectArrayElement(e, out, i, b);
    }
    return out;
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_commitCoeffsI16(JNIEnv* e, jobject self) {
    jxl_ctx* c = ctx_of(e, self);
    CHECK(jxl_vardct_commit_coeffs_i16(c));
}

/* the same planes without the zero-fill (JXL_MAP_NO_FILL): the decoder writes every sample of the groups it then names */
JNIEXPORT jobjectArray JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_mapCoeffsI16NoFill(JNIEnv* e, jobject self) {
    jxl_ctx* c = ctx_of(e, self);
    int16_t* pl[3];
    int32_t st[3], rows[3];
    jxl_status r = jxl_vardct_map_coeffs_i16_ex(c, pl, st, JXL_MAP_NO_FILL);
    if (r == JXL_OK) r = jxl_vardct_coeff_plane_rows(c, rows);
    if (r) { rethrow(e, c, r); return NULL; }
    jclass bb = (*e)->FindClass(e, "java/nio/ByteBuffer");
    if (!bb) return NULL;  /* exception pending */
    jobjectArray out = (*e)->NewObjectArray(e, 3, bb, NULL);
    if (!out) return NULL;
    for (int i = 0; i < 3; i++) {
        jobject b = (*e)->NewDirectByteBuffer(e, pl[i], (jlong)st[i] * rows[i] * 2);
        if (!b || (*e)->ExceptionCheck(e)) return NULL;
        (*e)->SetObjectArrayElement(e, out, i, b);
    }
    return out;
}

/* written: one byte per group of the frame (non-zero = its rectangle was fully written); the length is checked by the library */
JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_commitCoeffsI16Groups(JNIEnv* e, jobject self, jbyteArray written) {
    jxl_ctx* c = ctx_of(e, self);
    if (!written) { rethrow(e, c, JXL_ERR_INVALID_ARGUMENT); return; }
    const jsize n = (*e)->GetArrayLength(e, written);
    jbyte* w = (*e)->GetByteArrayElements(e, written, NULL);
    if (!w) return;  /* OutOfMemoryError pending */
    const jxl_status r = jxl_vardct_commit_coeffs_i16_groups(c, (const uint8_t*)w, (int32_t)n);
    (*e)->ReleaseByteArrayElements(e, written, w, JNI_ABORT);
    if (r) rethrow(e, c, r);
}

JNIEXPORT jobject JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_hostAlloc(JNIEnv* e, jclass k, jlong bytes) {
    (void)k;
    void* p = jxl_host_alloc((size_t)bytes);
    return p ? (*e)->NewDirectByteBuffer(e, p, bytes) : NULL;
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_hostFree(JNIEnv* e, jclass k, jobject b) {
    (void)k;
    if (b) jxl_host_free((*e)->GetDirectBufferAddress(e, b));
}

/* The output buffers of readOutput / readOutputBegin / finishFrame against the library's own numbers (jxl_vardct_output_geometry, r6):
 * og[1] rows of og[0] pixels -- the full padded frame, also for chroma-subsampled frames --, og[2] bytes per sample, times 3 when the
 * format interleaves the colours into ox (RGB8 / RGB16: oy and ob may then be null), `stride` pixels apart (>= og[0]).
 * 0: an exception is pending. */
static int out_room(JNIEnv* e, jxl_ctx* c, jobject ox, jobject oy, jobject ob, jlong stride) {
    int32_t og[5];
    jxl_status st_ = jxl_vardct_output_geometry(c, og);
    if (st_ != JXL_OK) {
        rethrow(e, c, st_);
        return 0;
    }
    if (stride < og[0]) { /* the library's own rule (enqueue_output): a row stride is at least a row */
        bad_arg(e, "jxlatte_amd: output stride shorter than a row");
        return 0;
    }
    const jlong need = (jlong)og[2] * (og[3] ? 3 : 1) * ((jlong)(og[1] - 1) * stride + og[0]);
    if (!has_room(e, ox, need) || (og[4] == 3 && (!has_room(e, oy, need) || !has_room(e, ob, need)))) {
        bad_arg(e, "jxlatte_amd: output buffer missing or too small");
        return 0;
    }
    return 1;
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_finishFrame(JNIEnv* e, jobject self, jobject ox, jobject oy, jobject ob,
        jlong stride) {
    jxl_ctx* c = ctx_of(e, self);
    if (!out_room(e, c, ox, oy, ob, stride)) return;
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
    if (!out_room(e, c, ox, oy, ob, stride)) return;
    void* out[3] = {ADDR(ox), ADDR(oy), ADDR(ob)};
    CHECK(jxl_vardct_read_output(c, out, stride));
}

/* readOutput in two halves: the host drives the next frame of this context between them (direct buffers from hostAlloc) */
JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_readOutputBegin(JNIEnv* e, jobject self, jobject ox, jobject oy, jobject ob,
        jlong stride) {
    jxl_ctx* c = ctx_of(e, self);
    if (!out_room(e, c, ox, oy, ob, stride)) return;
    void* out[3] = {ADDR(ox), ADDR(oy), ADDR(ob)};
    CHECK(jxl_vardct_read_output_begin(c, out, stride));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_readOutputWait(JNIEnv* e, jobject self) {
    jxl_ctx* c = ctx_of(e, self);
    CHECK(jxl_vardct_read_output_wait(c));
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
    (void)k;
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
    GETI(widths, n, w);
    GETI(heights, n, h);
    GETI(outWidths, n_out, ow);
    GETI(outHeights, n_out, oh);
    GETI(squeezeParams, n_sp * 4, spv);
    for (jsize i = 0; i < n; i++) {
        jobject b = (*e)->GetObjectArrayElement(e, chans, i);
        if ((*e)->ExceptionCheck(e)) return;
        ci[i].width = w[i]; ci[i].height = h[i];
        if (area(h[i], w[i]) > 0) NEED(b, 4 * area(h[i], w[i]));
        ci[i].data = (int32_t*)ADDR(b);
    }
    for (jsize i = 0; i < n_out; i++) {
        jobject b = (*e)->GetObjectArrayElement(e, out, i);
        if ((*e)->ExceptionCheck(e)) return;
        co[i].width = ow[i]; co[i].height = oh[i];
        if (area(oh[i], ow[i]) > 0) NEED(b, 4 * area(oh[i], ow[i]));
        co[i].data = (int32_t*)ADDR(b);
    }
    for (jsize i = 0; i < n_sp; i++) {
        sp[i].horizontal = spv[4 * i]; sp[i].in_place = spv[4 * i + 1]; sp[i].begin_c = spv[4 * i + 2]; sp[i].num_c = spv[4 * i + 3];
    }
    CHECK(jxl_modular_apply(c, ci, n, sp, n_sp, rctType, rctBegin, co, n_out));
}

/* ---- context / diagnostics ---- */
JNIEXPORT jstring JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_version(JNIEnv* e, jclass k) {
    (void)k;
    return (*e)->NewStringUTF(e, jxl_version());
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_synchronize(JNIEnv* e, jobject self) {
    jxl_ctx* c = ctx_of(e, self);
    CHECK(jxl_ctx_synchronize(c));
}

JNIEXPORT jlong JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_stream(JNIEnv* e, jobject self) {
    return (jlong)(intptr_t)jxl_ctx_stream(ctx_of(e, self));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_setStream(JNIEnv* e, jobject self, jlong hipStream) {
    jxl_ctx* c = ctx_of(e, self);
    CHECK(jxl_ctx_set_stream(c, (void*)(intptr_t)hipStream));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_copyOutputDevice(JNIEnv* e, jobject self, jlong dstDevice) {
    jxl_ctx* c = ctx_of(e, self);
    CHECK(jxl_vardct_copy_output_device(c, (void*)(intptr_t)dstDevice));
}

JNIEXPORT jint JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_outElemSize(JNIEnv* e, jobject self) {
    return jxl_vardct_out_elem_size(ctx_of(e, self));
}

JNIEXPORT jint JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_lastLaunchCount(JNIEnv* e, jobject self) {
    return jxl_vardct_last_launch_count(ctx_of(e, self));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_enableStageTiming(JNIEnv* e, jobject self, jboolean on) {
    jxl_ctx* c = ctx_of(e, self);
    CHECK(jxl_vardct_enable_stage_timing(c, on ? 1 : 0));
}

JNIEXPORT jfloat JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_lastStageMs(JNIEnv* e, jobject self, jint which) {
    jxl_ctx* c = ctx_of(e, self);
    float ms = 0.0f;
    const jxl_status st = jxl_vardct_last_stage_ms(c, which, &ms);
    if (st != JXL_OK) rethrow(e, c, st);
    return ms;
}

JNIEXPORT jintArray JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_coeffPlaneRows(JNIEnv* e, jobject self) {
    jxl_ctx* c = ctx_of(e, self);
    int32_t rows[3] = {0, 0, 0};
    const jxl_status st = jxl_vardct_coeff_plane_rows(c, rows);
    if (st != JXL_OK) { rethrow(e, c, st); return NULL; }
    jintArray out = (*e)->NewIntArray(e, 3);
    if (out) (*e)->SetIntArrayRegion(e, out, 0, 3, (const jint*)rows);
    return out;
}

/* ---- stage entries: one reference function each (host planes in / out as direct buffers) ---- */
JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_stageIdct2d(JNIEnv* e, jobject self, jobject src, jobject dst, jint h, jint w,
        jboolean transposed) {
    jxl_ctx* c = ctx_of(e, self);
    CHECK(jxl_stage_idct2d(c, (const float*)ADDR(src), (float*)ADDR(dst), h, w, transposed ? 1 : 0));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_stageFdct2d(JNIEnv* e, jobject self, jobject src, jobject dst, jint h, jint w) {
    jxl_ctx* c = ctx_of(e, self);
    CHECK(jxl_stage_fdct2d(c, (const float*)ADDR(src), (float*)ADDR(dst), h, w));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_stageGab(JNIEnv* e, jobject self, jobject i0, jobject i1, jobject i2, jobject o0,
        jobject o1, jobject o2, jint h, jint w, jfloatArray w1, jfloatArray w2) {
    jxl_ctx* c = ctx_of(e, self);
    const float* in[3] = {(const float*)ADDR(i0), (const float*)ADDR(i1), (const float*)ADDR(i2)};
    float* out[3] = {(float*)ADDR(o0), (float*)ADDR(o1), (float*)ADDR(o2)};
    float a[3], b[3];
    GETF(w1, 3, a);
    GETF(w2, 3, b);
    NEED(i0, 4 * area(h, w)); NEED(i1, 4 * area(h, w)); NEED(i2, 4 * area(h, w));
    NEED(o0, 4 * area(h, w)); NEED(o1, 4 * area(h, w)); NEED(o2, 4 * area(h, w));
    CHECK(jxl_stage_gab(c, in, out, h, w, a, b));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_stageEpf(JNIEnv* e, jobject self, jobject i0, jobject i1, jobject i2, jobject o0,
        jobject o1, jobject o2, jint h, jint w, jint iterations, jobject invSigma, jfloat invSigmaModular, jfloatArray channelScale,
        jfloat pass0, jfloat pass2, jfloat borderSadMul) {
    jxl_ctx* c = ctx_of(e, self);
    const float* in[3] = {(const float*)ADDR(i0), (const float*)ADDR(i1), (const float*)ADDR(i2)};
    float* out[3] = {(float*)ADDR(o0), (float*)ADDR(o1), (float*)ADDR(o2)};
    float cs[3];
    GETF(channelScale, 3, cs);
    NEED(i0, 4 * area(h, w)); NEED(i1, 4 * area(h, w)); NEED(i2, 4 * area(h, w));
    NEED(o0, 4 * area(h, w)); NEED(o1, 4 * area(h, w)); NEED(o2, 4 * area(h, w));
    NEED_OPT(invSigma, 4 * area(((jlong)h + 7) / 8, ((jlong)w + 7) / 8));
    CHECK(jxl_stage_epf(c, in, out, h, w, iterations, (const float*)ADDR(invSigma), invSigmaModular, cs, pass0, pass2, borderSadMul));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_stageEpfSigma(JNIEnv* e, jobject self, jobject hfMul, jobject sharpness, jint bh,
        jint bw, jfloat globalScale, jfloatArray sharpLut, jobject invSigma) {
    jxl_ctx* c = ctx_of(e, self);
    float lut[8];
    GETF(sharpLut, 8, lut);
    NEED(hfMul, 4 * area(bh, bw)); NEED(sharpness, 4 * area(bh, bw)); NEED(invSigma, 4 * area(bh, bw));
    CHECK(jxl_stage_epf_sigma(c, (const int32_t*)ADDR(hfMul), (const int32_t*)ADDR(sharpness), bh, bw, globalScale, lut, (float*)ADDR(invSigma)));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_stageLfDequant(JNIEnv* e, jobject self, jint lfgY, jint lfgX, jint cellsH,
        jint cellsW, jobject qX, jobject qY, jobject qB, jint extraPrecision, jfloatArray scaledDequant, jint xFactorLF, jint bFactorLF,
        jboolean adaptiveSmoothing, jfloat baseCorrX, jfloat baseCorrB, jint colorFactor, jobject o0, jobject o1, jobject o2) {
    jxl_ctx* c = ctx_of(e, self);
    jxl_lfquant_desc d;
    memset(&d, 0, sizeof d);
    d.lfg_y = lfgY; d.lfg_x = lfgX; d.cells_h = cellsH; d.cells_w = cellsW;
    d.lf_quant[0] = (const int32_t*)ADDR(qX); d.lf_quant[1] = (const int32_t*)ADDR(qY); d.lf_quant[2] = (const int32_t*)ADDR(qB);
    d.extra_precision = extraPrecision;
    GETF(scaledDequant, 3, d.scaled_dequant);
    d.x_factor_lf = xFactorLF; d.b_factor_lf = bFactorLF; d.adaptive_smoothing = adaptiveSmoothing ? 1 : 0;
    NEED(qX, 4 * area(cellsH, cellsW)); NEED(qY, 4 * area(cellsH, cellsW)); NEED(qB, 4 * area(cellsH, cellsW));
    NEED(o0, 4 * area(cellsH, cellsW)); NEED(o1, 4 * area(cellsH, cellsW)); NEED(o2, 4 * area(cellsH, cellsW));
    float* out[3] = {(float*)ADDR(o0), (float*)ADDR(o1), (float*)ADDR(o2)};
    CHECK(jxl_stage_lf_dequant(c, &d, baseCorrX, baseCorrB, colorFactor, out));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_stageXyb(JNIEnv* e, jobject self, jobject p0, jobject p1, jobject p2, jlong n,
        jfloatArray matrix, jfloatArray bias, jfloatArray cbrtBias, jfloat intensityTarget) {
    jxl_ctx* c = ctx_of(e, self);
    float* pl[3] = {(float*)ADDR(p0), (float*)ADDR(p1), (float*)ADDR(p2)};
    float m[9], b[3], cb[3];
    GETF(matrix, 9, m);
    GETF(bias, 3, b);
    GETF(cbrtBias, 3, cb);
    NEED(p0, 4 * n); NEED(p1, 4 * n); NEED(p2, 4 * n);
    CHECK(jxl_stage_xyb(c, pl, n, m, b, cb, intensityTarget));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_stageYcbcr(JNIEnv* e, jobject self, jobject p0, jobject p1, jobject p2, jlong n) {
    jxl_ctx* c = ctx_of(e, self);
    float* pl[3] = {(float*)ADDR(p0), (float*)ADDR(p1), (float*)ADDR(p2)};
    NEED(p0, 4 * n); NEED(p1, 4 * n); NEED(p2, 4 * n);
    CHECK(jxl_stage_ycbcr(c, pl, n));
}

/* transfer + quantise (PNGWriter.java:65,105-111): outF or outI is null */
JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_stageTransfer(JNIEnv* e, jobject self, jobject in, jlong n, jint transfer,
        jint maxValue, jobject outF, jobject outI) {
    jxl_ctx* c = ctx_of(e, self);
    NEED(in, 4 * n); NEED_OPT(outF, 4 * n); NEED_OPT(outI, 4 * n);
    CHECK(jxl_stage_transfer(c, (const float*)ADDR(in), n, transfer, maxValue, (float*)ADDR(outF), (int32_t*)ADDR(outI)));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_stageInvHSqueeze(JNIEnv* e, jobject self, jobject avg, jint aw, jobject res,
        jint rw, jint h, jobject out) {
    jxl_ctx* c = ctx_of(e, self);
    NEED(avg, 4 * area(h, aw)); NEED(res, 4 * area(h, rw)); NEED(out, 4 * area(h, (jlong)aw + rw));
    CHECK(jxl_stage_inv_hsqueeze(c, (const int32_t*)ADDR(avg), aw, (const int32_t*)ADDR(res), rw, h, (int32_t*)ADDR(out)));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_stageInvVSqueeze(JNIEnv* e, jobject self, jobject avg, jint ah, jobject res,
        jint rh, jint w, jobject out) {
    jxl_ctx* c = ctx_of(e, self);
    NEED(avg, 4 * area(ah, w)); NEED(res, 4 * area(rh, w)); NEED(out, 4 * area((jlong)ah + rh, w));
    CHECK(jxl_stage_inv_vsqueeze(c, (const int32_t*)ADDR(avg), ah, (const int32_t*)ADDR(res), rh, w, (int32_t*)ADDR(out)));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_stageRct(JNIEnv* e, jobject self, jobject v0, jobject v1, jobject v2, jlong n,
        jint rctType) {
    jxl_ctx* c = ctx_of(e, self);
    int32_t* v[3] = {(int32_t*)ADDR(v0), (int32_t*)ADDR(v1), (int32_t*)ADDR(v2)};
    NEED(v0, 4 * n); NEED(v1, 4 * n); NEED(v2, 4 * n);
    CHECK(jxl_stage_rct(c, v, n, rctType));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_stageModularToFloat(JNIEnv* e, jobject self, jobject a, jobject b, jlong n,
        jfloat scale, jobject out) {
    jxl_ctx* c = ctx_of(e, self);
    NEED(a, 4 * n); NEED_OPT(b, 4 * n); NEED(out, 4 * n);
    CHECK(jxl_stage_modular_to_float(c, (const int32_t*)ADDR(a), (const int32_t*)ADDR(b), n, scale, (float*)ADDR(out)));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_stageChromaUpsample(JNIEnv* e, jobject self, jobject in, jint h, jint w,
        jint xShift, jint yShift, jobject out) {
    jxl_ctx* c = ctx_of(e, self);
    if (xShift < 0 || xShift > 1 || yShift < 0 || yShift > 1) { bad_arg(e, "jxlatte_amd: chroma shift"); return; }
    NEED(in, 4 * area(h, w)); NEED(out, 4 * area((jlong)h << yShift, (jlong)w << xShift));
    CHECK(jxl_stage_chroma_upsample(c, (const float*)ADDR(in), h, w, xShift, yShift, (float*)ADDR(out)));
}

/* Frame.java:217-260 upsampling weights: packed (the bitstream's / default table) -> k * k * 25 floats */
JNIEXPORT jfloatArray JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_upsamplingWeights(JNIEnv* e, jclass k_, jint k, jfloatArray packed) {
    (void)k_;
    /* k first (2, 4 or 8: Frame.java:217), then the packed table's length for that k: 15, 55 or 210 weights */
    const jsize need = k == 2 ? 15 : k == 4 ? 55 : k == 8 ? 210 : -1;
    if (need < 0 || !packed || (*e)->GetArrayLength(e, packed) < need) {
        bad_arg(e, "jxlatte_amd: upsampling factor or packed weight table");
        return NULL;
    }
    jfloatArray out = (*e)->NewFloatArray(e, k * k * 25);
    if (!out) return NULL;  /* OutOfMemoryError pending */
    jfloat* p = (*e)->GetFloatArrayElements(e, packed, NULL);
    if (!p) return NULL;
    jfloat* o = (*e)->GetFloatArrayElements(e, out, NULL);
    const jxl_status st = o ? jxl_upsampling_weights(k, p, o) : JXL_ERR_OOM;
    if (o) (*e)->ReleaseFloatArrayElements(e, out, o, 0);
    (*e)->ReleaseFloatArrayElements(e, packed, p, JNI_ABORT);
    if (st != JXL_OK) { rethrow(e, NULL, st); return NULL; }
    return out;
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_stageUpsample(JNIEnv* e, jobject self, jobject in, jint h, jint w, jint k,
        jfloatArray weights, jobject out) {
    jxl_ctx* c = ctx_of(e, self);
    if ((k != 2 && k != 4 && k != 8) || !weights || (*e)->GetArrayLength(e, weights) < k * k * 25) { bad_arg(e, "jxlatte_amd: upsampling weights"); return; }
    NEED(in, 4 * area(h, w)); NEED(out, 4 * area((jlong)h * k, (jlong)w * k));
    jfloat* wt = (*e)->GetFloatArrayElements(e, weights, NULL);
    const jxl_status st = wt ? jxl_stage_upsample(c, (const float*)ADDR(in), h, w, k, wt, (float*)ADDR(out)) : JXL_ERR_OOM;
    if (wt) (*e)->ReleaseFloatArrayElements(e, weights, wt, JNI_ABORT);
    CHECK(st);
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_stageNoiseInit(JNIEnv* e, jobject self, jint h, jint w, jint groupDim,
        jlong seed0, jint colors, jobject o0, jobject o1, jobject o2) {
    jxl_ctx* c = ctx_of(e, self);
    float* out[3] = {(float*)ADDR(o0), (float*)ADDR(o1), (float*)ADDR(o2)};
    NEED(o0, 4 * area(h, w)); NEED(o1, 4 * area(h, w)); NEED(o2, 4 * area(h, w));
    CHECK(jxl_stage_noise_init(c, h, w, groupDim, (uint64_t)seed0, colors, out));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_stageNoiseAdd(JNIEnv* e, jobject self, jobject p0, jobject p1, jobject p2,
        jobject n0, jobject n1, jobject n2, jlong n, jfloatArray lut, jfloat baseCorrX, jfloat baseCorrB) {
    jxl_ctx* c = ctx_of(e, self);
    float* pl[3] = {(float*)ADDR(p0), (float*)ADDR(p1), (float*)ADDR(p2)};
    const float* nz[3] = {(const float*)ADDR(n0), (const float*)ADDR(n1), (const float*)ADDR(n2)};
    float l[8];
    GETF(lut, 8, l);
    NEED(p0, 4 * n); NEED(p1, 4 * n); NEED(p2, 4 * n); NEED(n0, 4 * n); NEED(n1, 4 * n); NEED(n2, 4 * n);
    CHECK(jxl_stage_noise_add(c, pl, nz, n, l, baseCorrX, baseCorrB));
}

/* rect: {h, w, canvas_y, canvas_x, frame_y, frame_x, ref_y, ref_x} (JXLCodestreamDecoder.java:26-40, 285-422) */
JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_stageBlend(JNIEnv* e, jobject self, jint mode, jint flags, jboolean isInt,
        jobject canvas, jint ch, jint cw, jobject frame, jint fh, jint fw, jobject ref, jint rh, jint rw, jobject frameAlpha, jobject refAlpha,
        jintArray rect) {
    jxl_ctx* c = ctx_of(e, self);
    jint r[8];
    GETI(rect, 8, r);
    NEED(canvas, 4 * area(ch, cw)); NEED(frame, 4 * area(fh, fw)); NEED_OPT(ref, 4 * area(rh, rw));
    NEED_OPT(frameAlpha, 4 * area(fh, fw)); NEED_OPT(refAlpha, 4 * area(rh, rw));
    jxl_blend_rect br;
    br.h = r[0]; br.w = r[1]; br.canvas_y = r[2]; br.canvas_x = r[3]; br.frame_y = r[4]; br.frame_x = r[5]; br.ref_y = r[6]; br.ref_x = r[7];
    CHECK(jxl_stage_blend(c, mode, (uint32_t)flags, isInt ? 1 : 0, ADDR(canvas), ch, cw, ADDR(frame), fh, fw, ADDR(ref), rh, rw,
                          (const float*)ADDR(frameAlpha), (const float*)ADDR(refAlpha), &br));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_stageOrient(JNIEnv* e, jobject self, jobject in, jint h, jint w, jint orientation,
        jobject out) {
    jxl_ctx* c = ctx_of(e, self);
    NEED(in, 4 * area(h, w)); NEED(out, 4 * area(h, w));
    CHECK(jxl_stage_orient(c, ADDR(in), h, w, orientation, ADDR(out)));
}

/* params: {height, width, n_color, has_alpha, premultiplied, bit_depth, big_endian, is_int[4], tagged_depth[4]} (PNGWriter.java:79-111) */
JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_stagePack(JNIEnv* e, jobject self, jobjectArray planes, jintArray params,
        jobject out) {
    jxl_ctx* c = ctx_of(e, self);
    jint pv[15];
    GETI(params, 15, pv);
    jxl_pack_params p;
    p.height = pv[0]; p.width = pv[1]; p.n_color = pv[2]; p.has_alpha = pv[3]; p.premultiplied = pv[4]; p.bit_depth = pv[5]; p.big_endian = pv[6];
    for (int i = 0; i < 4; i++) { p.is_int[i] = pv[7 + i]; p.tagged_depth[i] = pv[11 + i]; }
    const void* pl[4] = {NULL, NULL, NULL, NULL};
    if (!planes || p.n_color < 1 || p.n_color > 3 || (p.bit_depth != 8 && p.bit_depth != 16)) { bad_arg(e, "jxlatte_amd: pack parameters"); return; }
    const jsize n = (*e)->GetArrayLength(e, planes);
    const jsize want = p.n_color + (p.has_alpha ? 1 : 0);
    if (n < want) { bad_arg(e, "jxlatte_amd: pack: fewer planes than channels"); return; }
    for (jsize i = 0; i < want; i++) {
        jobject b = (*e)->GetObjectArrayElement(e, planes, i);
        NEED(b, 4 * area(p.height, p.width));
        pl[i] = ADDR(b);
    }
    NEED(out, area(p.height, p.width) * want * (p.bit_depth / 8));
    CHECK(jxl_stage_pack(c, pl, &p, ADDR(out)));
}

/* ---- Modular: plan once, run, read channel by channel (ModularStream.applyTransforms, ModularStream.java:110-131) ---- */
JNIEXPORT jintArray JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_modularDefaultSqueezeParams(JNIEnv* e, jclass k, jintArray widths,
        jintArray heights, jint nbMeta) {
    (void)k;
    const jsize n = (*e)->GetArrayLength(e, widths);
    jint w[256], h[256];
    jxl_squeeze_param sp[64];
    if (n > 256) { rethrow(e, NULL, JXL_ERR_INVALID_ARGUMENT); return NULL; }
    if (!get_ints(e, widths, n, w) || !get_ints(e, heights, n, h)) return NULL;
    const int32_t cnt = jxl_modular_default_squeeze_params((const int32_t*)w, (const int32_t*)h, n, nbMeta, sp, 64);
    if (cnt < 0) { rethrow(e, NULL, (jxl_status)cnt); return NULL; }
    jintArray out = (*e)->NewIntArray(e, cnt * 4);
    for (int32_t i = 0; out && i < cnt; i++) {
        const jint v[4] = {sp[i].horizontal, sp[i].in_place, sp[i].begin_c, sp[i].num_c};
        (*e)->SetIntArrayRegion(e, out, i * 4, 4, v);
    }
    return out;
}

/* returns {w0, h0, w1, h1, ...} of the channel list after the forward bookkeeping of the squeeze steps */
JNIEXPORT jintArray JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_modularSqueezedShapes(JNIEnv* e, jclass k, jintArray widths,
        jintArray heights, jintArray squeezeParams) {
    (void)k;
    const jsize n = (*e)->GetArrayLength(e, widths), n_sp = (*e)->GetArrayLength(e, squeezeParams) / 4;
    jint w[256], h[256], spv[256];
    int32_t ow[1024], oh[1024];
    jxl_squeeze_param sp[64];
    if (n > 256 || n_sp > 64) { rethrow(e, NULL, JXL_ERR_INVALID_ARGUMENT); return NULL; }
    if (!get_ints(e, widths, n, w) || !get_ints(e, heights, n, h) || !get_ints(e, squeezeParams, n_sp * 4, spv)) return NULL;
    for (jsize i = 0; i < n_sp; i++) {
        sp[i].horizontal = spv[4 * i]; sp[i].in_place = spv[4 * i + 1]; sp[i].begin_c = spv[4 * i + 2]; sp[i].num_c = spv[4 * i + 3];
    }
    const int32_t cnt = jxl_modular_squeezed_shapes((const int32_t*)w, (const int32_t*)h, n, sp, n_sp, ow, oh, 1024);
    if (cnt < 0) { rethrow(e, NULL, (jxl_status)cnt); return NULL; }
    jintArray out = (*e)->NewIntArray(e, cnt * 2);
    for (int32_t i = 0; out && i < cnt; i++) {
        const jint v[2] = {ow[i], oh[i]};
        (*e)->SetIntArrayRegion(e, out, i * 2, 2, v);
    }
    return out;
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_modularBegin(JNIEnv* e, jobject self, jobjectArray chans, jintArray widths,
        jintArray heights, jintArray squeezeParams, jint rctType, jint rctBegin) {
    jxl_ctx* c = ctx_of(e, self);
    const jsize n = (*e)->GetArrayLength(e, chans), n_sp = (*e)->GetArrayLength(e, squeezeParams) / 4;
    if (n > 256 || n_sp > 64) {
        rethrow(e, c, JXL_ERR_INVALID_ARGUMENT);
        return;
    }
    jxl_channel ci[256];
    jxl_squeeze_param sp[64];
    jint w[256], h[256], spv[256];
    GETI(widths, n, w);
    GETI(heights, n, h);
    GETI(squeezeParams, n_sp * 4, spv);
    for (jsize i = 0; i < n; i++) {
        jobject b = (*e)->GetObjectArrayElement(e, chans, i);
        if ((*e)->ExceptionCheck(e)) return;
        ci[i].width = w[i]; ci[i].height = h[i];
        if (area(h[i], w[i]) > 0) NEED(b, 4 * area(h[i], w[i]));  /* (an empty channel may come without a buffer) */
        ci[i].data = (int32_t*)ADDR(b);
    }
    for (jsize i = 0; i < n_sp; i++) {
        sp[i].horizontal = spv[4 * i]; sp[i].in_place = spv[4 * i + 1]; sp[i].begin_c = spv[4 * i + 2]; sp[i].num_c = spv[4 * i + 3];
    }
    CHECK(jxl_modular_begin(c, ci, n, sp, n_sp, rctType, rctBegin));
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_modularRun(JNIEnv* e, jobject self) {
    jxl_ctx* c = ctx_of(e, self);
    CHECK(jxl_modular_run(c));
}

JNIEXPORT jint JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_modularOutCount(JNIEnv* e, jobject self) {
    return jxl_modular_out_count(ctx_of(e, self));
}

JNIEXPORT jintArray JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_modularOutShape(JNIEnv* e, jobject self, jint idx) {
    jxl_ctx* c = ctx_of(e, self);
    int32_t wh[2] = {0, 0};
    const jxl_status st = jxl_modular_out_shape(c, idx, &wh[0], &wh[1]);
    if (st != JXL_OK) { rethrow(e, c, st); return NULL; }
    jintArray out = (*e)->NewIntArray(e, 2);
    if (out) (*e)->SetIntArrayRegion(e, out, 0, 2, (const jint*)wh);
    return out;
}

JNIEXPORT void JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_modularReadChannel(JNIEnv* e, jobject self, jint idx, jobject dst) {
    jxl_ctx* c = ctx_of(e, self);
    int32_t cw = 0, chh = 0;
    CHECK(jxl_modular_out_shape(c, idx, &cw, &chh));
    NEED(dst, 4 * area(chh, cw));
    CHECK(jxl_modular_read_channel(c, idx, (int32_t*)ADDR(dst)));
}

JNIEXPORT jint JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_modularLastLaunchCount(JNIEnv* e, jobject self) {
    return jxl_modular_last_launch_count(ctx_of(e, self));
}

JNIEXPORT jint JNICALL Java_com_traneptora_jxlatte_gpu_NativeBackend_modularRedoCount(JNIEnv* e, jobject self) {
    return jxl_modular_redo_count(ctx_of(e, self));
}
