/*
 * JNI front of libkmc.so for TLC (tlc2.tool.*) — what BASELINE's north star asks for: AbstractChecker and
 * the tlc2.TLC command line stay in Java; the worker loop, the fingerprint set and the state queue are
 * replaced by one native call that runs the whole breadth-first search on the GPU.
 *
 * NOT COMPILED OR TESTED HERE: this image has no JDK and TLC is not part of the reference repository.
 * The native half (kmcjni.c) is compiled against a stub jni.h by the test-suite; this class is the
 * matching Java half a TLC maintainer would drop into tlc2/tool/gpu/.  Names of TLC classes are [TLC-recall].
 */
package tlc2.tool.gpu;

public final class KmcModelChecker {
    static { System.loadLibrary("kmcjni"); }      // libkmcjni.so, linked against libkmc.so

    /** mirrors struct kmc_config (include/kmc.h); filled from the parsed ModelConfig (.cfg) */
    public static final class Config {
        public int model;                 // KMC_* model id, chosen from the root module's name
        public int nReplicas, logSize, maxRecords, maxLeaderEpoch, nLogRecords;
        public long maxId;
        public int invariantMask;         // bit k = k-th invariant of the model, in kmc_model_invariant_name order
        public boolean checkDeadlock, continueOnViolation, keepTrace;
        public int device;
        public long tableCapacity, frontierCapacity, hashSeed, maxLevels;
    }

    /** mirrors struct kmc_result */
    public static final class Result {
        public long generated, distinct, depth, queueLeft;
        public int verdict, violatedInvariant;
        public long violationDepth;
        public long[] violationCount = new long[4];
        public long[] actionGenerated = new long[16];
        public double secondsTotal, secondsExpand;
    }

    /** called once per BFS level from inside run(), on the calling thread */
    public interface Progress {
        void level(long depth, long newStates, long generatedTotal, long distinctTotal, double seconds);
    }

    /** one state of a counterexample: the action that produced it (null for the initial state) + canonical bytes */
    public static final class TraceState {
        public final String action;
        public final byte[] canonical;    // byte layout: include/kmc.h, "states as data"
        public TraceState(String action, byte[] canonical) { this.action = action; this.canonical = canonical; }
    }

    public static native long open(Config c);                    // kmc_open; throws IllegalStateException(kmc_last_error)
    public static native void run(long handle, Progress p);      // kmc_run
    public static native Result result(long handle);             // kmc_result_get
    public static native TraceState[] trace(long handle, int model);   // kmc_trace (needs keepTrace) + kmc_action_name
    public static native boolean contains(long handle, long[] packedState);   // kmc_contains (FPSet.contains)
    public static native void checkpoint(long handle, String path);           // kmc_checkpoint_save
    public static native void recover(long handle, String path, Progress p);  // kmc_checkpoint_load + kmc_resume
    public static native void close(long handle);                // kmc_close

    private KmcModelChecker() {}
}
