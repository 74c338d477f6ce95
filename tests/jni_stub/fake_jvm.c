/* fake_jvm.c — a STAND-IN for the JVM side of integration/jni/kmcjni.c.  TEST INFRASTRUCTURE ONLY.
 *
 * No JDK exists in this image, so the JNI glue has never met a JVM.  What CAN be done without one: give kmcjni.c a JNIEnv
 * whose function table (tests/jni_stub/jni.h: the JNI functions the glue uses, with the specification's signatures) is
 * implemented by a few hundred lines of C over a toy object model, and play the Java half — KmcModelChecker.java's Config /
 * Result / Progress / TraceState and a caller of its eight native methods — from a C main().  Every line of kmcjni.c then
 * EXECUTES against libkmc.so: field marshalling, the progress call-back, result objects, trace arrays, exceptions.
 *
 * What this is not: a JVM.  Object layout, class loading, exceptions as control flow, GC pinning and the real jni.h's member
 * order are not exercised; KmcModelChecker.java itself is still never compiled.  INTEGRATION.md section 2 says so.
 *
 *   gcc -std=gnu11 -I tests/jni_stub -I include tests/jni_stub/fake_jvm.c integration/jni/kmcjni.c \
 *       -L kafka_specification_amd -lkmc -Wl,-rpath,$PWD/kafka_specification_amd -o tests/_jni_harness
 *   tests/_jni_harness MODEL_ID N L R E INV_MASK [trace] [levels=K ckpt=PATH]      -> one JSON object on stdout
 */
#include <jni.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kmc.h"

/* the natives under test (kmcjni.c) */
jlong Java_tlc2_tool_gpu_KmcModelChecker_open(JNIEnv*, jclass, jobject);
void Java_tlc2_tool_gpu_KmcModelChecker_run(JNIEnv*, jclass, jlong, jobject);
jobject Java_tlc2_tool_gpu_KmcModelChecker_result(JNIEnv*, jclass, jlong);
jobjectArray Java_tlc2_tool_gpu_KmcModelChecker_trace(JNIEnv*, jclass, jlong, jint);
jboolean Java_tlc2_tool_gpu_KmcModelChecker_contains(JNIEnv*, jclass, jlong, jlongArray);
void Java_tlc2_tool_gpu_KmcModelChecker_checkpoint(JNIEnv*, jclass, jlong, jstring);
void Java_tlc2_tool_gpu_KmcModelChecker_recover(JNIEnv*, jclass, jlong, jstring, jobject);
void Java_tlc2_tool_gpu_KmcModelChecker_close(JNIEnv*, jclass, jlong);

/* ---- a toy object model ------------------------------------------------------------------------------------------------ */
enum { K_OBJECT, K_CLASS, K_STRING, K_LONGS, K_BYTES, K_OBJECTS };
typedef struct {
    char name[40], sig[24];
    union { jint i; jlong j; jdouble d; jboolean z; jobject o; } v;
} Field;
struct _jobject {
    int kind;
    jobject cls;          /* objects: their class */
    char name[72];        /* classes: binary name; strings: unused */
    Field fields[24];
    int nfields;
    jsize len;            /* arrays */
    jlong* longs;
    jbyte* bytes;
    jobject* objects;
    char* utf;            /* strings */
};
struct _jfieldID { char name[40], sig[24]; };
struct _jmethodID { char name[40], sig[40]; };

static struct {
    int pending;               /* an exception is pending */
    char ex_class[72], ex_msg[1100];
    jobject classes[16];
    int nclasses;
    struct _jfieldID fids[64];
    int nfids;
    struct _jmethodID mids[16];
    int nmids;
    /* what the "Java" Progress object does with a level */
    int levels_seen;
    jlong last_depth, last_distinct;
} VM;

static jobject new_obj(int kind) {
    jobject o = (jobject)calloc(1, sizeof *o);
    o->kind = kind;
    return o;
}
static Field* field_of(jobject o, jfieldID f, int create) {
    for (int i = 0; i < o->nfields; ++i)
        if (!strcmp(o->fields[i].name, f->name)) return &o->fields[i];
    if (!create || o->nfields >= 24) return NULL;
    Field* n = &o->fields[o->nfields++];
    strncpy(n->name, f->name, sizeof n->name - 1);
    strncpy(n->sig, f->sig, sizeof n->sig - 1);
    return n;
}
static void raise_(const char* cls, const char* msg) {
    VM.pending = 1;
    strncpy(VM.ex_class, cls, sizeof VM.ex_class - 1);
    strncpy(VM.ex_msg, msg, sizeof VM.ex_msg - 1);
}

/* ---- the JNI functions kmcjni.c uses ----------------------------------------------------------------------------------- */
static jclass FindClass(JNIEnv* env, const char* name) {
    (void)env;
    for (int i = 0; i < VM.nclasses; ++i)
        if (!strcmp(VM.classes[i]->name, name)) return VM.classes[i];
    static const char* known[] = {"java/lang/IllegalStateException", "tlc2/tool/gpu/KmcModelChecker$Result",
                                  "tlc2/tool/gpu/KmcModelChecker$TraceState", "tlc2/tool/gpu/KmcModelChecker$Config",
                                  "tlc2/tool/gpu/KmcModelChecker$Progress"};
    for (unsigned k = 0; k < sizeof known / sizeof *known; ++k)
        if (!strcmp(known[k], name)) {
            jobject c = new_obj(K_CLASS);
            strncpy(c->name, name, sizeof c->name - 1);
            return VM.classes[VM.nclasses++] = c;
        }
    raise_("java/lang/NoClassDefFoundError", name);
    return NULL;
}
static jint ThrowNew(JNIEnv* env, jclass c, const char* msg) { (void)env; raise_(c->name, msg); return 0; }
static jboolean ExceptionCheck(JNIEnv* env) { (void)env; return (jboolean)VM.pending; }
static jclass GetObjectClass(JNIEnv* env, jobject o) { (void)env; return o->cls; }
static jmethodID GetMethodID(JNIEnv* env, jclass c, const char* name, const char* sig) {
    (void)env;
    /* the methods the Java half really has (KmcModelChecker.java) */
    const int ok = (strstr(c->name, "$Result") && !strcmp(name, "<init>") && !strcmp(sig, "()V")) ||
                   (strstr(c->name, "$TraceState") && !strcmp(name, "<init>") && !strcmp(sig, "(Ljava/lang/String;[B)V")) ||
                   (strstr(c->name, "$Progress") && !strcmp(name, "level") && !strcmp(sig, "(JJJJD)V"));
    if (!ok) { raise_("java/lang/NoSuchMethodError", name); return NULL; }
    for (int i = 0; i < VM.nmids; ++i)
        if (!strcmp(VM.mids[i].name, name) && !strcmp(VM.mids[i].sig, sig)) return &VM.mids[i];
    struct _jmethodID* m = &VM.mids[VM.nmids++];
    strncpy(m->name, name, sizeof m->name - 1);
    strncpy(m->sig, sig, sizeof m->sig - 1);
    return m;
}
static jobject new_longs(jsize n) {
    jobject a = new_obj(K_LONGS);
    a->len = n;
    a->longs = (jlong*)calloc((size_t)n + 1, sizeof(jlong));
    return a;
}
static jobject NewObject(JNIEnv* env, jclass c, jmethodID m, ...) {
    (void)env;
    jobject o = new_obj(K_OBJECT);
    o->cls = c;
    if (strstr(c->name, "$Result")) {   /* the field initialisers of KmcModelChecker.Result */
        struct _jfieldID vc = {"violationCount", "[J"}, ag = {"actionGenerated", "[J"};
        field_of(o, &vc, 1)->v.o = new_longs(4);
        field_of(o, &ag, 1)->v.o = new_longs(16);
    } else if (strstr(c->name, "$TraceState")) {   /* TraceState(String action, byte[] canonical) */
        va_list ap;
        va_start(ap, m);
        struct _jfieldID fa = {"action", "Ljava/lang/String;"}, fc = {"canonical", "[B"};
        field_of(o, &fa, 1)->v.o = va_arg(ap, jobject);
        field_of(o, &fc, 1)->v.o = va_arg(ap, jobject);
        va_end(ap);
    }
    return o;
}
static void CallVoidMethod(JNIEnv* env, jobject obj, jmethodID m, ...) {   /* Progress.level(JJJJD) */
    (void)env; (void)obj;
    if (strcmp(m->name, "level")) { raise_("java/lang/NoSuchMethodError", m->name); return; }
    va_list ap;
    va_start(ap, m);
    const jlong depth = va_arg(ap, jlong), fresh = va_arg(ap, jlong), generated = va_arg(ap, jlong), distinct = va_arg(ap, jlong);
    const jdouble seconds = va_arg(ap, jdouble);
    va_end(ap);
    (void)fresh; (void)generated; (void)seconds;
    VM.levels_seen++;
    VM.last_depth = depth;
    VM.last_distinct = distinct;
}
static jfieldID GetFieldID(JNIEnv* env, jclass c, const char* name, const char* sig) {
    (void)env; (void)c;
    for (int i = 0; i < VM.nfids; ++i)
        if (!strcmp(VM.fids[i].name, name) && !strcmp(VM.fids[i].sig, sig)) return &VM.fids[i];
    if (VM.nfids >= 64) { raise_("java/lang/OutOfMemoryError", "field ids"); return NULL; }
    struct _jfieldID* f = &VM.fids[VM.nfids++];
    strncpy(f->name, name, sizeof f->name - 1);
    strncpy(f->sig, sig, sizeof f->sig - 1);
    return f;
}
/* a field read of a name / type the object does not have is what a JVM answers with NoSuchFieldError (at GetFieldID time) */
static Field* must(jobject o, jfieldID f) {
    Field* x = field_of(o, f, 0);
    if (!x || strcmp(x->sig, f->sig)) { raise_("java/lang/NoSuchFieldError", f->name); return NULL; }
    return x;
}
static jobject GetObjectField(JNIEnv* env, jobject o, jfieldID f) { (void)env; Field* x = must(o, f); return x ? x->v.o : NULL; }
static jboolean GetBooleanField(JNIEnv* env, jobject o, jfieldID f) { (void)env; Field* x = must(o, f); return x ? x->v.z : 0; }
static jint GetIntField(JNIEnv* env, jobject o, jfieldID f) { (void)env; Field* x = must(o, f); return x ? x->v.i : 0; }
static jlong GetLongField(JNIEnv* env, jobject o, jfieldID f) { (void)env; Field* x = must(o, f); return x ? x->v.j : 0; }
static void SetIntField(JNIEnv* env, jobject o, jfieldID f, jint v) { (void)env; field_of(o, f, 1)->v.i = v; }
static void SetLongField(JNIEnv* env, jobject o, jfieldID f, jlong v) { (void)env; field_of(o, f, 1)->v.j = v; }
static void SetDoubleField(JNIEnv* env, jobject o, jfieldID f, jdouble v) { (void)env; field_of(o, f, 1)->v.d = v; }
static jstring NewStringUTF(JNIEnv* env, const char* utf) {
    (void)env;
    jobject s = new_obj(K_STRING);
    s->utf = strdup(utf);
    return s;
}
static const char* GetStringUTFChars(JNIEnv* env, jstring s, jboolean* copy) { (void)env; if (copy) *copy = 0; return s->utf; }
static void ReleaseStringUTFChars(JNIEnv* env, jstring s, const char* c) { (void)env; (void)s; (void)c; }
static jsize GetArrayLength(JNIEnv* env, jarray a) { (void)env; return a->len; }
static jobjectArray NewObjectArray(JNIEnv* env, jsize n, jclass c, jobject init) {
    (void)env; (void)c;
    jobject a = new_obj(K_OBJECTS);
    a->len = n;
    a->objects = (jobject*)calloc((size_t)n + 1, sizeof(jobject));
    for (jsize i = 0; i < n; ++i) a->objects[i] = init;
    return a;
}
static void SetObjectArrayElement(JNIEnv* env, jobjectArray a, jsize i, jobject v) {
    (void)env;
    if (i < 0 || i >= a->len) { raise_("java/lang/ArrayIndexOutOfBoundsException", "object array"); return; }
    a->objects[i] = v;
}
static jbyteArray NewByteArray(JNIEnv* env, jsize n) {
    (void)env;
    jobject a = new_obj(K_BYTES);
    a->len = n;
    a->bytes = (jbyte*)calloc((size_t)n + 1, 1);
    return a;
}
static void SetByteArrayRegion(JNIEnv* env, jbyteArray a, jsize s, jsize n, const jbyte* b) {
    (void)env;
    if (s < 0 || n < 0 || s + n > a->len) { raise_("java/lang/ArrayIndexOutOfBoundsException", "byte array"); return; }
    memcpy(a->bytes + s, b, (size_t)n);
}
static void GetLongArrayRegion(JNIEnv* env, jlongArray a, jsize s, jsize n, jlong* b) {
    (void)env;
    if (s < 0 || n < 0 || s + n > a->len) { raise_("java/lang/ArrayIndexOutOfBoundsException", "long array"); return; }
    memcpy(b, a->longs + s, (size_t)n * sizeof(jlong));
}
static void SetLongArrayRegion(JNIEnv* env, jlongArray a, jsize s, jsize n, const jlong* b) {
    (void)env;
    if (s < 0 || n < 0 || s + n > a->len) { raise_("java/lang/ArrayIndexOutOfBoundsException", "long array"); return; }
    memcpy(a->longs + s, b, (size_t)n * sizeof(jlong));
}

static const struct JNINativeInterface_ TABLE = {
    FindClass, ThrowNew, ExceptionCheck, GetObjectClass, GetMethodID, NewObject, CallVoidMethod, GetFieldID, GetObjectField,
    GetBooleanField, GetIntField, GetLongField, SetIntField, SetLongField, SetDoubleField, NewStringUTF, GetStringUTFChars,
    ReleaseStringUTFChars, GetArrayLength, NewObjectArray, SetObjectArrayElement, NewByteArray, SetByteArrayRegion,
    GetLongArrayRegion, SetLongArrayRegion};

/* ---- the "Java" side: what a caller of KmcModelChecker does --------------------------------------------------------------- */
static void set_i(JNIEnv* env, jobject o, const char* n, jint v) { SetIntField(env, o, GetFieldID(env, o->cls, n, "I"), v); }
static void set_j(JNIEnv* env, jobject o, const char* n, jlong v) { SetLongField(env, o, GetFieldID(env, o->cls, n, "J"), v); }
static void set_z(JNIEnv* env, jobject o, const char* n, jboolean v) {
    struct _jfieldID f;
    memset(&f, 0, sizeof f);
    strncpy(f.name, n, sizeof f.name - 1);
    strcpy(f.sig, "Z");
    field_of(o, &f, 1)->v.z = v;
    (void)env;
}
static jlong long_field(JNIEnv* env, jobject o, const char* n) { return GetLongField(env, o, GetFieldID(env, o->cls, n, "J")); }
static jint int_field(JNIEnv* env, jobject o, const char* n) { return GetIntField(env, o, GetFieldID(env, o->cls, n, "I")); }

static void json_escape(const char* s) {
    for (; *s; ++s) {
        if (*s == '"' || *s == '\\') putchar('\\');
        putchar(*s == '\n' ? ' ' : *s);
    }
}
static int report_exception(void) {
    printf("{\"exception\": \"%s\", \"message\": \"", VM.ex_class);
    json_escape(VM.ex_msg);
    printf("\"}\n");
    return 3;
}

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: %s MODEL_ID N L R E INV_MASK [trace] [levels=K] [ckpt=PATH] [device=D]\n", argv[0]); return 2; }
    JNIEnv envp = &TABLE;
    JNIEnv* env = &envp;
    int trace = 0, levels = 0, device = 0;
    const char* ckpt = NULL;
    for (int i = 7; i < argc; ++i) {
        if (!strcmp(argv[i], "trace")) trace = 1;
        else if (!strncmp(argv[i], "levels=", 7)) levels = atoi(argv[i] + 7);
        else if (!strncmp(argv[i], "ckpt=", 5)) ckpt = argv[i] + 5;
        else if (!strncmp(argv[i], "device=", 7)) device = atoi(argv[i] + 7);
    }
    /* new KmcModelChecker.Config() { ... } */
    jobject cfg = new_obj(K_OBJECT);
    cfg->cls = FindClass(env, "tlc2/tool/gpu/KmcModelChecker$Config");
    const jint model = atoi(argv[1]);
    set_i(env, cfg, "model", model);
    set_i(env, cfg, "nReplicas", atoi(argv[2]));
    set_i(env, cfg, "logSize", atoi(argv[3]));
    set_i(env, cfg, "maxRecords", atoi(argv[4]));
    set_i(env, cfg, "maxLeaderEpoch", atoi(argv[5]));
    set_i(env, cfg, "nLogRecords", 2);
    set_j(env, cfg, "maxId", 10);
    set_i(env, cfg, "invariantMask", atoi(argv[6]));
    set_z(env, cfg, "checkDeadlock", 0);
    set_z(env, cfg, "continueOnViolation", 0);
    set_z(env, cfg, "keepTrace", (jboolean)trace);
    set_i(env, cfg, "device", device);
    set_j(env, cfg, "tableCapacity", 1 << 20);
    set_j(env, cfg, "frontierCapacity", 1 << 18);
    set_j(env, cfg, "hashSeed", 0);
    set_j(env, cfg, "maxLevels", levels);
    jobject progress = new_obj(K_OBJECT);
    progress->cls = FindClass(env, "tlc2/tool/gpu/KmcModelChecker$Progress");

    jlong h = Java_tlc2_tool_gpu_KmcModelChecker_open(env, NULL, cfg);
    if (VM.pending) return report_exception();
    Java_tlc2_tool_gpu_KmcModelChecker_run(env, NULL, h, progress);
    if (VM.pending) return report_exception();
    int resumed_levels = -1;
    if (ckpt && levels) {   /* checkpoint the level-limited search, recover it into a second handle without the limit */
        jstring path = NewStringUTF(env, ckpt);
        Java_tlc2_tool_gpu_KmcModelChecker_checkpoint(env, NULL, h, path);
        if (VM.pending) return report_exception();
        Java_tlc2_tool_gpu_KmcModelChecker_close(env, NULL, h);
        set_j(env, cfg, "maxLevels", 0);
        h = Java_tlc2_tool_gpu_KmcModelChecker_open(env, NULL, cfg);
        if (VM.pending) return report_exception();
        const int before = VM.levels_seen;
        Java_tlc2_tool_gpu_KmcModelChecker_recover(env, NULL, h, path, progress);
        if (VM.pending) return report_exception();
        resumed_levels = VM.levels_seen - before;
    }
    jobject res = Java_tlc2_tool_gpu_KmcModelChecker_result(env, NULL, h);
    if (VM.pending) return report_exception();
    /* FPSet.contains on the initial state (packed by the library itself) */
    uint64_t init[16] = {0};
    kmc_init_state((kmc_handle*)(intptr_t)h, init);
    const jsize W = (jsize)kmc_state_words((kmc_handle*)(intptr_t)h);
    jobject packed = new_longs(W);
    for (jsize k = 0; k < W; ++k) packed->longs[k] = (jlong)init[k];
    const jboolean has_init = Java_tlc2_tool_gpu_KmcModelChecker_contains(env, NULL, h, packed);
    if (VM.pending) return report_exception();
    packed->longs[0] ^= 0x5555;   /* some other bit pattern */
    const jboolean has_other = Java_tlc2_tool_gpu_KmcModelChecker_contains(env, NULL, h, packed);
    if (VM.pending) return report_exception();

    struct _jfieldID fvc = {"violationCount", "[J"}, fag = {"actionGenerated", "[J"}, fd1 = {"secondsTotal", "D"};
    jobject vc = GetObjectField(env, res, &fvc), ag = GetObjectField(env, res, &fag);
    printf("{\"generated\": %lld, \"distinct\": %lld, \"depth\": %lld, \"queue_left\": %lld, \"verdict\": %d, "
           "\"violated_invariant\": %d, \"violation_depth\": %lld, \"seconds_total\": %.6f, \"levels_seen_by_progress\": %d, "
           "\"last_progress_depth\": %lld, \"last_progress_distinct\": %lld, \"resumed_levels\": %d, \"contains_init\": %d, "
           "\"contains_other\": %d, \"violation_count\": [%lld, %lld, %lld, %lld], \"action_generated\": [",
           (long long)long_field(env, res, "generated"), (long long)long_field(env, res, "distinct"),
           (long long)long_field(env, res, "depth"), (long long)long_field(env, res, "queueLeft"), int_field(env, res, "verdict"),
           int_field(env, res, "violatedInvariant"), (long long)long_field(env, res, "violationDepth"),
           field_of(res, &fd1, 0) ? field_of(res, &fd1, 0)->v.d : -1.0, VM.levels_seen, (long long)VM.last_depth,
           (long long)VM.last_distinct, resumed_levels, has_init, has_other, (long long)vc->longs[0], (long long)vc->longs[1],
           (long long)vc->longs[2], (long long)vc->longs[3]);
    for (int k = 0; k < 16; ++k) printf("%s%lld", k ? ", " : "", (long long)ag->longs[k]);
    printf("], \"trace\": [");
    if (trace && int_field(env, res, "verdict") == KMC_V_INVARIANT) {
        jobject tr = Java_tlc2_tool_gpu_KmcModelChecker_trace(env, NULL, h, model);
        if (VM.pending) { printf("]}\n"); return report_exception(); }
        struct _jfieldID fa = {"action", "Ljava/lang/String;"}, fc = {"canonical", "[B"};
        for (jsize k = 0; tr && k < tr->len; ++k) {
            jobject ts = tr->objects[k], act = GetObjectField(env, ts, &fa), bytes = GetObjectField(env, ts, &fc);
            printf("%s{\"action\": ", k ? ", " : "");
            if (act) printf("\"%s\"", act->utf); else printf("null");
            printf(", \"canonical\": \"");
            for (jsize b = 0; b < bytes->len; ++b) printf("%02x", (unsigned)(uint8_t)bytes->bytes[b]);
            printf("\"}");
        }
    }
    printf("]}\n");
    Java_tlc2_tool_gpu_KmcModelChecker_close(env, NULL, h);
    return 0;
}
