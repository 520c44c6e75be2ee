/* jni.h — a DECLARATION STUB of the Java Native Interface header, written by hand for this repository's CPU test suite.
 *
 * The build image has no JDK.  tests/test_jni_binding_cpu.py compiles src/jni/fmhip_jni.cpp against this file and links the
 * result against libfmhip.so, so that every C++ error in the JNI layer and every mismatch between it and include/fmhip.h is
 * caught here, today — not when somebody first builds the layer with a real JDK.  It declares the primitive and reference
 * types of the JNI specification (Java SE, "JNI Types and Data Structures") and the JNIEnv member functions fmhip_jni.cpp
 * uses, with the specification's signatures, and NOTHING else; no function has a body (the shared library built in the test is
 * never loaded into a JVM).  It is not a substitute for <jni.h>: CMakeLists.txt builds the real layer with find_package(JNI).
 */
#ifndef FMHIP_TESTS_JNI_STUB_H
#define FMHIP_TESTS_JNI_STUB_H

#include <stdint.h>

#ifndef __cplusplus
#error "the stub covers the C++ flavour of the interface only (JNIEnv with member functions)"
#endif

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
#define JNI_COMMIT 1

typedef int32_t  jint;
typedef int64_t  jlong;
typedef int8_t   jbyte;
typedef uint8_t  jboolean;
typedef uint16_t jchar;
typedef int16_t  jshort;
typedef float    jfloat;
typedef double   jdouble;
typedef jint     jsize;

class _jobject {};
class _jclass : public _jobject {};
class _jstring : public _jobject {};
class _jarray : public _jobject {};
class _jobjectArray : public _jarray {};
class _jintArray : public _jarray {};
class _jlongArray : public _jarray {};
class _jfloatArray : public _jarray {};
class _jdoubleArray : public _jarray {};
typedef _jobject*      jobject;
typedef _jclass*       jclass;
typedef _jstring*      jstring;
typedef _jarray*       jarray;
typedef _jobjectArray* jobjectArray;
typedef _jintArray*    jintArray;
typedef _jlongArray*   jlongArray;
typedef _jfloatArray*  jfloatArray;
typedef _jdoubleArray* jdoubleArray;

struct JNIEnv_ {
    jsize    GetArrayLength(jarray array);
    jstring  NewStringUTF(const char* utf);
    void     SetObjectArrayElement(jobjectArray array, jsize index, jobject value);

    jint*    GetIntArrayElements(jintArray array, jboolean* isCopy);
    jlong*   GetLongArrayElements(jlongArray array, jboolean* isCopy);
    jfloat*  GetFloatArrayElements(jfloatArray array, jboolean* isCopy);
    jdouble* GetDoubleArrayElements(jdoubleArray array, jboolean* isCopy);
    void     ReleaseIntArrayElements(jintArray array, jint* elems, jint mode);
    void     ReleaseLongArrayElements(jlongArray array, jlong* elems, jint mode);
    void     ReleaseFloatArrayElements(jfloatArray array, jfloat* elems, jint mode);
    void     ReleaseDoubleArrayElements(jdoubleArray array, jdouble* elems, jint mode);

    void     SetIntArrayRegion(jintArray array, jsize start, jsize len, const jint* buf);
    void     SetLongArrayRegion(jlongArray array, jsize start, jsize len, const jlong* buf);
    void     SetDoubleArrayRegion(jdoubleArray array, jsize start, jsize len, const jdouble* buf);
};
typedef JNIEnv_ JNIEnv;

#endif
