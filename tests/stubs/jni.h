/* TEST STUB -- not the JDK's jni.h and not ABI-compatible with a JVM.
 *
 * This image has no JDK, so integration/jni/jxlatte_amd_jni.c could not even be parsed here. This header declares, in our own words,
 * exactly the JNI types and the JNIEnv function-table entries the shim uses (names, parameter order and C types as the JNI
 * specification gives them), so that
 *   (1) tests/test_jni_shim.py can run `gcc -fsyntax-only -Wall -Wextra` over the shim (it stops 850 lines from rotting), and
 *   (2) tests/stubs/fake_jni.c can implement those entries over plain C structs and the shim's entry points can be CALLED -- through
 *       ctypes, with direct buffers and arrays that are ordinary memory -- to check its size guards and its error mapping against the
 *       real library on the GPU box.
 * The table below holds only those entries and in our own order: a shim object compiled against this file must never be loaded into
 * a JVM. It proves nothing about a JVM. */
#ifndef JXLATTE_AMD_TEST_STUB_JNI_H
#define JXLATTE_AMD_TEST_STUB_JNI_H
#include <stdint.h>

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef float jfloat;
typedef jint jsize;

struct fake_object;
typedef struct fake_object* jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jthrowable;
typedef jobject jarray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jbyteArray;
typedef jarray jfloatArray;
typedef jarray jobjectArray;
struct fake_field;
typedef struct fake_field* jfieldID;

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_FALSE 0
#define JNI_TRUE 1
#define JNI_COMMIT 1
#define JNI_ABORT 2

struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;

struct JNINativeInterface_ {
    /* classes, fields, exceptions */
    jclass (*FindClass)(JNIEnv* env, const char* name);
    jclass (*GetObjectClass)(JNIEnv* env, jobject obj);
    jfieldID (*GetFieldID)(JNIEnv* env, jclass cls, const char* name, const char* sig);
    jlong (*GetLongField)(JNIEnv* env, jobject obj, jfieldID field);
    jint (*ThrowNew)(JNIEnv* env, jclass cls, const char* message);
    jboolean (*ExceptionCheck)(JNIEnv* env);
    /* strings and arrays */
    jstring (*NewStringUTF)(JNIEnv* env, const char* utf);
    jsize (*GetArrayLength)(JNIEnv* env, jarray array);
    jintArray (*NewIntArray)(JNIEnv* env, jsize length);
    jfloatArray (*NewFloatArray)(JNIEnv* env, jsize length);
    jobjectArray (*NewObjectArray)(JNIEnv* env, jsize length, jclass elementClass, jobject initialElement);
    jobject (*GetObjectArrayElement)(JNIEnv* env, jobjectArray array, jsize index);
    void (*SetObjectArrayElement)(JNIEnv* env, jobjectArray array, jsize index, jobject value);
    jbyte* (*GetByteArrayElements)(JNIEnv* env, jbyteArray array, jboolean* isCopy);
    void (*ReleaseByteArrayElements)(JNIEnv* env, jbyteArray array, jbyte* elems, jint mode);
    jfloat* (*GetFloatArrayElements)(JNIEnv* env, jfloatArray array, jboolean* isCopy);
    void (*ReleaseFloatArrayElements)(JNIEnv* env, jfloatArray array, jfloat* elems, jint mode);
    void (*GetIntArrayRegion)(JNIEnv* env, jintArray array, jsize start, jsize len, jint* buf);
    void (*SetIntArrayRegion)(JNIEnv* env, jintArray array, jsize start, jsize len, const jint* buf);
    void (*GetLongArrayRegion)(JNIEnv* env, jlongArray array, jsize start, jsize len, jlong* buf);
    void (*GetFloatArrayRegion)(JNIEnv* env, jfloatArray array, jsize start, jsize len, jfloat* buf);
    /* direct buffers */
    jobject (*NewDirectByteBuffer)(JNIEnv* env, void* address, jlong capacity);
    void* (*GetDirectBufferAddress)(JNIEnv* env, jobject buf);
    jlong (*GetDirectBufferCapacity)(JNIEnv* env, jobject buf);
};

#endif
