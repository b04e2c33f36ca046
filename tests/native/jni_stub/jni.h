/* jni.h -- DECLARATION-ONLY STUB for type-checking mvtopicmodel_amd/java/mvhdp_jni.cpp in an image without a JDK
 * (tests/test_jni_shim.py compiles the shim with -fsyntax-only against it).  It declares the few JNI names the shim
 * uses, with the signatures of the JNI specification; it implements nothing, nothing links against it, and it is not
 * part of any build.  A real build uses $JAVA_HOME/include/jni.h. */
#ifndef MVHDP_TEST_JNI_STUB_H
#define MVHDP_TEST_JNI_STUB_H
#include <stdint.h>
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
typedef int32_t jint; typedef int64_t jlong; typedef double jdouble; typedef uint8_t jboolean; typedef int8_t jbyte; typedef jint jsize;
class _jobject {}; class _jclass : public _jobject {}; class _jarray : public _jobject {};
class _jintArray : public _jarray {}; class _jlongArray : public _jarray {}; class _jdoubleArray : public _jarray {};
class _jbooleanArray : public _jarray {}; class _jobjectArray : public _jarray {}; class _jbyteArray : public _jarray {};
typedef _jobject* jobject; typedef _jclass* jclass; typedef _jarray* jarray; typedef _jintArray* jintArray;
typedef _jlongArray* jlongArray; typedef _jdoubleArray* jdoubleArray; typedef _jbooleanArray* jbooleanArray; typedef _jobjectArray* jobjectArray;
typedef _jbyteArray* jbyteArray;
struct _jfieldID; typedef _jfieldID* jfieldID;
struct JNIEnv {
    jclass FindClass(const char*); jint ThrowNew(jclass, const char*); jsize GetArrayLength(jarray);
    jobject GetObjectArrayElement(jobjectArray, jsize); void DeleteLocalRef(jobject); jclass GetObjectClass(jobject);
    jfieldID GetFieldID(jclass, const char*, const char*);
    void SetLongField(jobject, jfieldID, jlong); void SetIntField(jobject, jfieldID, jint); void SetDoubleField(jobject, jfieldID, jdouble);
    jint* GetIntArrayElements(jintArray, jboolean*); void ReleaseIntArrayElements(jintArray, jint*, jint);
    jlong* GetLongArrayElements(jlongArray, jboolean*); void ReleaseLongArrayElements(jlongArray, jlong*, jint);
    jdouble* GetDoubleArrayElements(jdoubleArray, jboolean*); void ReleaseDoubleArrayElements(jdoubleArray, jdouble*, jint);
    void GetIntArrayRegion(jintArray, jsize, jsize, jint*); void GetDoubleArrayRegion(jdoubleArray, jsize, jsize, jdouble*);
    void GetBooleanArrayRegion(jbooleanArray, jsize, jsize, jboolean*);
    void SetDoubleArrayRegion(jdoubleArray, jsize, jsize, const jdouble*); void SetBooleanArrayRegion(jbooleanArray, jsize, jsize, const jboolean*);
    void GetLongArrayRegion(jlongArray, jsize, jsize, jlong*); void SetLongArrayRegion(jlongArray, jsize, jsize, const jlong*);
    void SetIntArrayRegion(jintArray, jsize, jsize, const jint*);
    void GetByteArrayRegion(jbyteArray, jsize, jsize, jbyte*); void SetByteArrayRegion(jbyteArray, jsize, jsize, const jbyte*);
};
#endif
