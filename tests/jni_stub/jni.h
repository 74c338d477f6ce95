/* jni.h STAND-IN for syntax-checking integration/jni/kmcjni.c where no JDK exists.  It declares only the
 * JNI types and JNIEnv functions that file uses, with the signatures of the Java Native Interface
 * specification (JNI 1.6, "Interface Function Table").  It is NOT a JNI implementation and nothing
 * links against it; the member order is irrelevant because nothing is ever called through it. */
#ifndef KMC_TEST_JNI_STUB_H
#define KMC_TEST_JNI_STUB_H
#include <stdarg.h>
#include <stdint.h>

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef double jdouble;
typedef jint jsize;

struct _jobject;
typedef struct _jobject* jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jobjectArray;
typedef jarray jbyteArray;
typedef jarray jlongArray;
struct _jfieldID;
typedef struct _jfieldID* jfieldID;
struct _jmethodID;
typedef struct _jmethodID* jmethodID;

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL

struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;

struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv* env, const char* name);
    jint (*ThrowNew)(JNIEnv* env, jclass clazz, const char* msg);
    jboolean (*ExceptionCheck)(JNIEnv* env);
    jclass (*GetObjectClass)(JNIEnv* env, jobject obj);
    jmethodID (*GetMethodID)(JNIEnv* env, jclass clazz, const char* name, const char* sig);
    jobject (*NewObject)(JNIEnv* env, jclass clazz, jmethodID methodID, ...);
    void (*CallVoidMethod)(JNIEnv* env, jobject obj, jmethodID methodID, ...);
    jfieldID (*GetFieldID)(JNIEnv* env, jclass clazz, const char* name, const char* sig);
    jobject (*GetObjectField)(JNIEnv* env, jobject obj, jfieldID fieldID);
    jboolean (*GetBooleanField)(JNIEnv* env, jobject obj, jfieldID fieldID);
    jint (*GetIntField)(JNIEnv* env, jobject obj, jfieldID fieldID);
    jlong (*GetLongField)(JNIEnv* env, jobject obj, jfieldID fieldID);
    void (*SetIntField)(JNIEnv* env, jobject obj, jfieldID fieldID, jint val);
    void (*SetLongField)(JNIEnv* env, jobject obj, jfieldID fieldID, jlong val);
    void (*SetDoubleField)(JNIEnv* env, jobject obj, jfieldID fieldID, jdouble val);
    jstring (*NewStringUTF)(JNIEnv* env, const char* utf);
    const char* (*GetStringUTFChars)(JNIEnv* env, jstring str, jboolean* isCopy);
    void (*ReleaseStringUTFChars)(JNIEnv* env, jstring str, const char* chars);
    jsize (*GetArrayLength)(JNIEnv* env, jarray array);
    jobjectArray (*NewObjectArray)(JNIEnv* env, jsize len, jclass clazz, jobject init);
    void (*SetObjectArrayElement)(JNIEnv* env, jobjectArray array, jsize index, jobject val);
    jbyteArray (*NewByteArray)(JNIEnv* env, jsize len);
    void (*SetByteArrayRegion)(JNIEnv* env, jbyteArray array, jsize start, jsize len, const jbyte* buf);
    void (*GetLongArrayRegion)(JNIEnv* env, jlongArray array, jsize start, jsize len, jlong* buf);
    void (*SetLongArrayRegion)(JNIEnv* env, jlongArray array, jsize start, jsize len, const jlong* buf);
};
#endif
