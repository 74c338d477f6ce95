/* kmcjni.c — JNI glue between tlc2.tool.gpu.KmcModelChecker (KmcModelChecker.java) and the C ABI of
 * libkmc.so (include/kmc.h).  No logic lives here: every native method marshals its arguments, makes ONE
 * kmc_* call and turns a non-zero status into a Java exception carrying kmc_last_error().
 *
 * NEVER RUN UNDER A JVM (no JDK in this image).  It is compiled by tests/test_jni_shim.py against
 * tests/jni_stub/jni.h — a stand-in that declares the JNI functions used below with the signatures of the
 * JNI specification — its undefined symbols are exactly kmc_* entry points, and since round 4 it is EXECUTED:
 * tests/jni_stub/fake_jvm.c implements that function table over a toy object model and plays the Java half,
 * so every line below runs against libkmc.so (tests/test_gpu_jni_harness.py on the GPU; the failed-open
 * exception path on any box).  A stand-in is not a JVM: class loading, the real jni.h and KmcModelChecker.java
 * itself remain unexercised.
 * Build against a real JDK:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include kmcjni.c \
 *       -L../../kafka_specification_amd -lkmc -o libkmcjni.so
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "kmc.h"

#define H(handle) ((kmc_handle*)(intptr_t)(handle))

static void throw_kmc(JNIEnv* env, const char* what) {
    char msg[1024];
    const char* err = kmc_last_error();
    strncpy(msg, what, sizeof msg - 1);
    msg[sizeof msg - 1] = 0;
    if (err && *err) {
        strncat(msg, ": ", sizeof msg - strlen(msg) - 1);
        strncat(msg, err, sizeof msg - strlen(msg) - 1);
    }
    jclass ex = (*env)->FindClass(env, "java/lang/IllegalStateException");
    if (ex) (*env)->ThrowNew(env, ex, msg);
}

static jint get_int(JNIEnv* env, jobject o, jclass c, const char* name) {
    return (*env)->GetIntField(env, o, (*env)->GetFieldID(env, c, name, "I"));
}
static jlong get_long(JNIEnv* env, jobject o, jclass c, const char* name) {
    return (*env)->GetLongField(env, o, (*env)->GetFieldID(env, c, name, "J"));
}
static jboolean get_bool(JNIEnv* env, jobject o, jclass c, const char* name) {
    return (*env)->GetBooleanField(env, o, (*env)->GetFieldID(env, c, name, "Z"));
}

JNIEXPORT jlong JNICALL Java_tlc2_tool_gpu_KmcModelChecker_open(JNIEnv* env, jclass self, jobject jc) {
    (void)self;
    kmc_config c;
    memset(&c, 0, sizeof c);
    jclass cc = (*env)->GetObjectClass(env, jc);
    c.model = get_int(env, jc, cc, "model");
    c.n_replicas = get_int(env, jc, cc, "nReplicas");
    c.log_size = get_int(env, jc, cc, "logSize");
    c.max_records = get_int(env, jc, cc, "maxRecords");
    c.max_leader_epoch = get_int(env, jc, cc, "maxLeaderEpoch");
    c.n_log_records = get_int(env, jc, cc, "nLogRecords");
    c.max_id = get_long(env, jc, cc, "maxId");
    c.invariant_mask = (uint32_t)get_int(env, jc, cc, "invariantMask");
    c.check_deadlock = get_bool(env, jc, cc, "checkDeadlock");
    c.continue_on_violation = get_bool(env, jc, cc, "continueOnViolation");
    c.keep_trace = get_bool(env, jc, cc, "keepTrace");
    c.device = get_int(env, jc, cc, "device");
    c.n_shards = 1;
    c.table_capacity = (uint64_t)get_long(env, jc, cc, "tableCapacity");
    c.frontier_capacity = (uint64_t)get_long(env, jc, cc, "frontierCapacity");
    c.hash_seed = (uint64_t)get_long(env, jc, cc, "hashSeed");
    c.max_levels = (uint64_t)get_long(env, jc, cc, "maxLevels");
    if ((*env)->ExceptionCheck(env)) return 0;   /* a field name did not resolve */
    kmc_handle* h = NULL;
    if (kmc_open(&c, &h) != KMC_OK) {
        throw_kmc(env, "kmc_open");
        return 0;
    }
    return (jlong)(intptr_t)h;
}

typedef struct {
    JNIEnv* env;
    jobject progress;
    jmethodID level;
} progress_ctx;

static void on_level(const kmc_level_info* i, void* user) {
    progress_ctx* p = (progress_ctx*)user;
    if (!p->progress || (*p->env)->ExceptionCheck(p->env)) return;
    (*p->env)->CallVoidMethod(p->env, p->progress, p->level, (jlong)i->depth, (jlong)i->new_states,
                              (jlong)i->generated_total, (jlong)i->distinct_total, (jdouble)i->seconds);
}

static int make_progress(JNIEnv* env, jobject jp, progress_ctx* ctx) {
    ctx->env = env;
    ctx->progress = jp;
    ctx->level = NULL;
    if (jp) {
        ctx->level = (*env)->GetMethodID(env, (*env)->GetObjectClass(env, jp), "level", "(JJJJD)V");
        if (!ctx->level) return 0;
    }
    return 1;
}

JNIEXPORT void JNICALL Java_tlc2_tool_gpu_KmcModelChecker_run(JNIEnv* env, jclass self, jlong handle, jobject jp) {
    (void)self;
    progress_ctx ctx;
    if (!make_progress(env, jp, &ctx)) return;
    if (kmc_run(H(handle), on_level, &ctx) != KMC_OK) throw_kmc(env, "kmc_run");
}

JNIEXPORT jobject JNICALL Java_tlc2_tool_gpu_KmcModelChecker_result(JNIEnv* env, jclass self, jlong handle) {
    (void)self;
    kmc_result r;
    if (kmc_result_get(H(handle), &r) != KMC_OK) {
        throw_kmc(env, "kmc_result_get");
        return NULL;
    }
    jclass rc = (*env)->FindClass(env, "tlc2/tool/gpu/KmcModelChecker$Result");
    if (!rc) return NULL;
    jobject o = (*env)->NewObject(env, rc, (*env)->GetMethodID(env, rc, "<init>", "()V"));
    if (!o) return NULL;
#define SETJ(name, v) (*env)->SetLongField(env, o, (*env)->GetFieldID(env, rc, name, "J"), (jlong)(v))
#define SETI(name, v) (*env)->SetIntField(env, o, (*env)->GetFieldID(env, rc, name, "I"), (jint)(v))
#define SETD(name, v) (*env)->SetDoubleField(env, o, (*env)->GetFieldID(env, rc, name, "D"), (jdouble)(v))
    SETJ("generated", r.generated);
    SETJ("distinct", r.distinct);
    SETJ("depth", r.depth);
    SETJ("queueLeft", r.queue_left);
    SETI("verdict", r.verdict);
    SETI("violatedInvariant", r.violated_invariant);
    SETJ("violationDepth", r.violation_depth);
    SETD("secondsTotal", r.seconds_total);
    SETD("secondsExpand", r.seconds_expand);
    jlong tmp[KMC_MAX_KINDS];
    jlongArray vc = (jlongArray)(*env)->GetObjectField(env, o, (*env)->GetFieldID(env, rc, "violationCount", "[J"));
    for (int k = 0; k < 4; ++k) tmp[k] = (jlong)r.violation_count[k];
    if (vc) (*env)->SetLongArrayRegion(env, vc, 0, 4, tmp);
    jlongArray ag = (jlongArray)(*env)->GetObjectField(env, o, (*env)->GetFieldID(env, rc, "actionGenerated", "[J"));
    for (int k = 0; k < KMC_MAX_KINDS; ++k) tmp[k] = (jlong)r.action_generated[k];
    if (ag) (*env)->SetLongArrayRegion(env, ag, 0, KMC_MAX_KINDS, tmp);
    return o;
}

JNIEXPORT jobjectArray JNICALL Java_tlc2_tool_gpu_KmcModelChecker_trace(JNIEnv* env, jclass self, jlong handle,
                                                                       jint model) {
    (void)self;
    kmc_handle* h = H(handle);
    uint64_t n = 0;
    if (kmc_trace(h, NULL, NULL, 0, &n) != KMC_OK) {   /* first call: the length only */
        throw_kmc(env, "kmc_trace");
        return NULL;
    }
    const uint64_t cb = kmc_canon_bytes(h);
    uint8_t* states = (uint8_t*)malloc((size_t)(n * cb) + 1);
    int32_t* kinds = (int32_t*)malloc((size_t)n * sizeof(int32_t) + 4);
    jobjectArray out = NULL;
    if (states && kinds && kmc_trace(h, states, kinds, n, &n) == KMC_OK) {
        jclass tc = (*env)->FindClass(env, "tlc2/tool/gpu/KmcModelChecker$TraceState");
        jmethodID ctor = tc ? (*env)->GetMethodID(env, tc, "<init>", "(Ljava/lang/String;[B)V") : NULL;
        out = ctor ? (*env)->NewObjectArray(env, (jsize)n, tc, NULL) : NULL;
        for (uint64_t k = 0; out && k < n; ++k) {
            jbyteArray b = (*env)->NewByteArray(env, (jsize)cb);
            if (!b) { out = NULL; break; }
            (*env)->SetByteArrayRegion(env, b, 0, (jsize)cb, (const jbyte*)(states + k * cb));
            /* kinds[k] < 0 marks the initial state: its action stays null */
            jstring name = kinds[k] >= 0 ? (*env)->NewStringUTF(env, kmc_action_name(model, kinds[k])) : NULL;
            jobject ts = (*env)->NewObject(env, tc, ctor, name, b);
            if (!ts) { out = NULL; break; }
            (*env)->SetObjectArrayElement(env, out, (jsize)k, ts);
        }
    } else {
        throw_kmc(env, "kmc_trace");
    }
    free(states);
    free(kinds);
    return out;
}

JNIEXPORT jboolean JNICALL Java_tlc2_tool_gpu_KmcModelChecker_contains(JNIEnv* env, jclass self, jlong handle,
                                                                       jlongArray packed) {
    (void)self;
    kmc_handle* h = H(handle);
    const uint64_t w = kmc_state_words(h);
    if ((uint64_t)(*env)->GetArrayLength(env, packed) != w) {
        throw_kmc(env, "contains: wrong number of state words");
        return 0;
    }
    uint64_t words[16];
    jlong tmp[16];
    (*env)->GetLongArrayRegion(env, packed, 0, (jsize)w, tmp);
    for (uint64_t k = 0; k < w && k < 16; ++k) words[k] = (uint64_t)tmp[k];
    int32_t present = 0;
    if (kmc_contains(h, words, &present) != KMC_OK) throw_kmc(env, "kmc_contains");
    return (jboolean)(present != 0);
}

JNIEXPORT void JNICALL Java_tlc2_tool_gpu_KmcModelChecker_checkpoint(JNIEnv* env, jclass self, jlong handle, jstring path) {
    (void)self;
    const char* p = (*env)->GetStringUTFChars(env, path, NULL);
    if (!p) return;
    const int rc = kmc_checkpoint_save(H(handle), p);
    (*env)->ReleaseStringUTFChars(env, path, p);
    if (rc != KMC_OK) throw_kmc(env, "kmc_checkpoint_save");
}

JNIEXPORT void JNICALL Java_tlc2_tool_gpu_KmcModelChecker_recover(JNIEnv* env, jclass self, jlong handle, jstring path,
                                                                  jobject jp) {
    (void)self;
    const char* p = (*env)->GetStringUTFChars(env, path, NULL);
    if (!p) return;
    int rc = kmc_checkpoint_load(H(handle), p);
    (*env)->ReleaseStringUTFChars(env, path, p);
    if (rc != KMC_OK) {
        throw_kmc(env, "kmc_checkpoint_load");
        return;
    }
    progress_ctx ctx;
    if (!make_progress(env, jp, &ctx)) return;
    if (kmc_resume(H(handle), on_level, &ctx) != KMC_OK) throw_kmc(env, "kmc_resume");
}

JNIEXPORT void JNICALL Java_tlc2_tool_gpu_KmcModelChecker_close(JNIEnv* env, jclass self, jlong handle) {
    (void)env;
    (void)self;
    kmc_close(H(handle));
}
