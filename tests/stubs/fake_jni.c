/* TEST HARNESS: the JNIEnv entries of tests/stubs/jni.h over plain C structs, so that the entry points of
 * integration/jni/jxlatte_amd_jni.c can be called from ctypes (tests/test_jni_shim.py). "Objects" are heap records; nothing is ever
 * collected (a test process is short-lived). Exceptions: the first ThrowNew is recorded (class name + message) and ExceptionCheck
 * reports it, as a JVM would leave it pending for the Java caller. Not a JVM, no claim about one. */
#define _POSIX_C_SOURCE 200809L
#include <jni.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { K_CLASS = 1, K_SELF, K_STRING, K_DIRECT, K_INTS, K_LONGS, K_BYTES, K_FLOATS, K_OBJECTS };

struct fake_object {
    int kind;
    void* data;     /* arrays: elements; direct buffer: address; class / string: the characters */
    jlong n;        /* arrays: length; direct buffer: capacity */
    jlong ctx;      /* K_SELF: the `long ctx` field of NativeBackend */
};
struct fake_field {
    char name[32];
};

typedef struct {
    const struct JNINativeInterface_* table; /* must be first: JNIEnv* points here */
    int pending;
    char cls[128];
    char msg[512];
} fake_env;

static jobject mk(int kind, void* data, jlong n) {
    jobject o = (jobject)calloc(1, sizeof *o);
    o->kind = kind;
    o->data = data;
    o->n = n;
    return o;
}
static jobject mk_array(int kind, jlong n, size_t elem, const void* src) {
    void* d = calloc((size_t)(n > 0 ? n : 1), elem);
    if (src && n > 0) memcpy(d, src, (size_t)n * elem);
    return mk(kind, d, n);
}

static jclass f_FindClass(JNIEnv* env, const char* name) {
    (void)env;
    return mk(K_CLASS, strdup(name), (jlong)strlen(name));
}
static jclass f_GetObjectClass(JNIEnv* env, jobject obj) {
    (void)obj;
    return f_FindClass(env, "com/traneptora/jxlatte/gpu/NativeBackend");
}
static jfieldID f_GetFieldID(JNIEnv* env, jclass cls, const char* name, const char* sig) {
    (void)env; (void)cls; (void)sig;
    jfieldID f = (jfieldID)calloc(1, sizeof *f);
    strncpy(f->name, name, sizeof f->name - 1);
    return f;
}
static jlong f_GetLongField(JNIEnv* env, jobject obj, jfieldID field) {
    (void)env;
    return obj && obj->kind == K_SELF && field && !strcmp(field->name, "ctx") ? obj->ctx : 0;
}
static jint f_ThrowNew(JNIEnv* env, jclass cls, const char* message) {
    fake_env* fe = (fake_env*)env;
    if (!fe->pending) {
        fe->pending = 1;
        snprintf(fe->cls, sizeof fe->cls, "%s", cls && cls->kind == K_CLASS ? (const char*)cls->data : "?");
        snprintf(fe->msg, sizeof fe->msg, "%s", message ? message : "");
    }
    return 0;
}
static jboolean f_ExceptionCheck(JNIEnv* env) { return ((fake_env*)env)->pending ? JNI_TRUE : JNI_FALSE; }
static jstring f_NewStringUTF(JNIEnv* env, const char* utf) {
    (void)env;
    return mk(K_STRING, strdup(utf ? utf : ""), (jlong)strlen(utf ? utf : ""));
}
static jsize f_GetArrayLength(JNIEnv* env, jarray a) {
    (void)env;
    return a ? (jsize)a->n : 0;
}
static jintArray f_NewIntArray(JNIEnv* env, jsize n) { (void)env; return mk_array(K_INTS, n, sizeof(jint), NULL); }
static jfloatArray f_NewFloatArray(JNIEnv* env, jsize n) { (void)env; return mk_array(K_FLOATS, n, sizeof(jfloat), NULL); }
static jobjectArray f_NewObjectArray(JNIEnv* env, jsize n, jclass cls, jobject init) {
    (void)env; (void)cls;
    jobject a = mk_array(K_OBJECTS, n, sizeof(jobject), NULL);
    for (jsize i = 0; i < n; i++) ((jobject*)a->data)[i] = init;
    return a;
}
static jobject f_GetObjectArrayElement(JNIEnv* env, jobjectArray a, jsize i) {
    (void)env;
    return a && a->kind == K_OBJECTS && i >= 0 && i < a->n ? ((jobject*)a->data)[i] : NULL;
}
static void f_SetObjectArrayElement(JNIEnv* env, jobjectArray a, jsize i, jobject v) {
    (void)env;
    if (a && a->kind == K_OBJECTS && i >= 0 && i < a->n) ((jobject*)a->data)[i] = v;
}
static jbyte* f_GetByteArrayElements(JNIEnv* env, jbyteArray a, jboolean* is_copy) {
    (void)env;
    if (is_copy) *is_copy = JNI_FALSE;
    return a ? (jbyte*)a->data : NULL;
}
static void f_ReleaseByteArrayElements(JNIEnv* env, jbyteArray a, jbyte* p, jint mode) { (void)env; (void)a; (void)p; (void)mode; }
static jfloat* f_GetFloatArrayElements(JNIEnv* env, jfloatArray a, jboolean* is_copy) {
    (void)env;
    if (is_copy) *is_copy = JNI_FALSE;
    return a ? (jfloat*)a->data : NULL;
}
static void f_ReleaseFloatArrayElements(JNIEnv* env, jfloatArray a, jfloat* p, jint mode) { (void)env; (void)a; (void)p; (void)mode; }
/* region access: out of range = ArrayIndexOutOfBoundsException, as the specification says */
static int region_ok(JNIEnv* env, jarray a, int kind, jsize start, jsize len) {
    if (a && a->kind == kind && start >= 0 && len >= 0 && (jlong)start + len <= a->n) return 1;
    f_ThrowNew(env, f_FindClass(env, "java/lang/ArrayIndexOutOfBoundsException"), "array region out of range");
    return 0;
}
static void f_GetIntArrayRegion(JNIEnv* env, jintArray a, jsize s, jsize n, jint* buf) {
    if (region_ok(env, a, K_INTS, s, n)) memcpy(buf, (jint*)a->data + s, (size_t)n * sizeof(jint));
}
static void f_SetIntArrayRegion(JNIEnv* env, jintArray a, jsize s, jsize n, const jint* buf) {
    if (region_ok(env, a, K_INTS, s, n)) memcpy((jint*)a->data + s, buf, (size_t)n * sizeof(jint));
}
static void f_GetLongArrayRegion(JNIEnv* env, jlongArray a, jsize s, jsize n, jlong* buf) {
    if (region_ok(env, a, K_LONGS, s, n)) memcpy(buf, (jlong*)a->data + s, (size_t)n * sizeof(jlong));
}
static void f_GetFloatArrayRegion(JNIEnv* env, jfloatArray a, jsize s, jsize n, jfloat* buf) {
    if (region_ok(env, a, K_FLOATS, s, n)) memcpy(buf, (jfloat*)a->data + s, (size_t)n * sizeof(jfloat));
}
static jobject f_NewDirectByteBuffer(JNIEnv* env, void* address, jlong capacity) { (void)env; return mk(K_DIRECT, address, capacity); }
static void* f_GetDirectBufferAddress(JNIEnv* env, jobject b) { (void)env; return b && b->kind == K_DIRECT ? b->data : NULL; }
static jlong f_GetDirectBufferCapacity(JNIEnv* env, jobject b) { (void)env; return b && b->kind == K_DIRECT ? b->n : -1; }

static const struct JNINativeInterface_ k_table = {
    f_FindClass, f_GetObjectClass, f_GetFieldID, f_GetLongField, f_ThrowNew, f_ExceptionCheck,
    f_NewStringUTF, f_GetArrayLength, f_NewIntArray, f_NewFloatArray, f_NewObjectArray, f_GetObjectArrayElement, f_SetObjectArrayElement,
    f_GetByteArrayElements, f_ReleaseByteArrayElements, f_GetFloatArrayElements, f_ReleaseFloatArrayElements,
    f_GetIntArrayRegion, f_SetIntArrayRegion, f_GetLongArrayRegion, f_GetFloatArrayRegion,
    f_NewDirectByteBuffer, f_GetDirectBufferAddress, f_GetDirectBufferCapacity,
};

/* ---- what the Python side calls ---- */
#define API __attribute__((visibility("default")))
API void* fj_env_new(void) {
    fake_env* fe = (fake_env*)calloc(1, sizeof *fe);
    fe->table = &k_table;
    return fe;
}
API const char* fj_pending_class(void* env) { return ((fake_env*)env)->pending ? ((fake_env*)env)->cls : NULL; }
API const char* fj_pending_message(void* env) { return ((fake_env*)env)->pending ? ((fake_env*)env)->msg : NULL; }
API void fj_clear(void* env) { ((fake_env*)env)->pending = 0; }
API void* fj_self(jlong ctx) {
    jobject o = mk(K_SELF, NULL, 0);
    o->ctx = ctx;
    return o;
}
API void* fj_direct(void* address, jlong capacity) { return mk(K_DIRECT, address, capacity); }
API void* fj_ints(const jint* v, jlong n) { return mk_array(K_INTS, n, sizeof(jint), v); }
API void* fj_longs(const jlong* v, jlong n) { return mk_array(K_LONGS, n, sizeof(jlong), v); }
API void* fj_bytes(const jbyte* v, jlong n) { return mk_array(K_BYTES, n, sizeof(jbyte), v); }
API void* fj_floats(const jfloat* v, jlong n) { return mk_array(K_FLOATS, n, sizeof(jfloat), v); }
API void* fj_objects(jlong n) { return mk_array(K_OBJECTS, n, sizeof(jobject), NULL); }
API void fj_set_object(void* arr, jlong i, void* v) { f_SetObjectArrayElement(NULL, (jobject)arr, (jsize)i, (jobject)v); }
API void* fj_get_object(void* arr, jlong i) { return f_GetObjectArrayElement(NULL, (jobject)arr, (jsize)i); }
API int fj_kind(void* o) { return o ? ((jobject)o)->kind : 0; }
API void* fj_data(void* o) { return o ? ((jobject)o)->data : NULL; }
API jlong fj_length(void* o) { return o ? ((jobject)o)->n : -1; }
