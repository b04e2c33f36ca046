package org.madgik.MVTopicModel;

/**
 * JNI binding of libmvhdp.so (include/mvhdp.h): the MI355X replacement of the
 * FastQMVWVWorkerRunnable / FastQMVWVUpdaterRunnable threads that
 * FastQMVWVParallelTopicModel.estimate() submits every iteration.
 *
 * NOT compiled in the build image (no JDK there); kept as the reference-side
 * binding a maintainer adds.  See INTEGRATION.md for the patch to estimate().
 *
 * One instance = one model shard on one GPU.  Not thread-safe, like estimate().
 */
public final class NativeSampler implements AutoCloseable {

    static {
        System.loadLibrary("mvhdp_jni");   // libmvhdp_jni.so, linked against libmvhdp.so
    }

    public static final int SWEEP_REUSE_TREES = 0x1;
    public static final int SWEEP_NO_APPLY = 0x2;
    public static final int SWEEP_EXACT_CHAIN = 0x4;
    public static final int SWEEP_FROZEN = 0x10;       // the inferencer's call (nst = 1, nut = 0)
    public static final int SWEEP_LIVE = 0x20;         // the updater threads' own discipline: atomics on the shared counts
    public static final int SWEEP_SEGMENT_APPLY = 0x40; // deterministic: segments sampled one after the other, deltas applied in between
    public static int sweepLiveSegments(int n) { return (n & 0xff) << 16; }

    /** What one sweep reports: the three branch counters of the worker plus bookkeeping. */
    public static final class SweepStats {
        public long tokens, changed, newMassCnt, topicDocMassCnt, wordFTreeMassCnt, oovSkipped, abortedDocs, exactFallbacks;
        public int activatedTopic, activatedModality;
        public long activationKey;
        public double sweepKernelMs, totalMs;
        public int activations;          // topics activated during the call (a live / segmented sweep activates at every segment border)
    }

    private long handle;   // mvhdp_handle

    public NativeSampler(int numTopics, int[] numTypes, int device, long docIdBase) {
        handle = nCreate(numTopics, numTypes, device, docIdBase);
    }

    /** tokens = FeatureSequence indices of every entity's view m, docOff = CSR offsets (size by getLength()). */
    public void setCorpus(int m, long[] docOff, int[] tokens) { nSetCorpus(handle, m, docOff, tokens); }
    public void setAssignments(int m, int[] z) { nSetAssignments(handle, m, z); }
    public void getAssignments(int m, int[] z) { nGetAssignments(handle, m, z); }

    public void setHyper(double[][] alpha, double[] alphaSum, double[] beta, double[] betaSum, double[] gamma,
                         double[][] p_a, double[][] p_b, boolean[] inactive) {
        nSetHyper(handle, alpha, alphaSum, beta, betaSum, gamma, p_a, p_b, inactive);
    }

    public void buildCounts() { nBuildCounts(handle); }
    public void buildTrees() { nBuildTrees(handle); }
    public void getCounts(int m, int[] typeTopicCountsFlat, int[] tokensPerTopic) { nGetCounts(handle, m, typeTopicCountsFlat, tokensPerTopic); }
    public void getDocTopicHist(int m, int[] histFlat, int histLen, int[] docLengthCounts) { nGetDocTopicHist(handle, m, histFlat, histLen, docLengthCounts); }
    public void getAlpha(double[] alphaFlat, boolean[] inactive) { nGetAlpha(handle, alphaFlat, inactive); }

    /** pOverride: [D*M*M] view weights drawn as WorkerRunnable lines 327-337, or null to draw them on the device. */
    public SweepStats sweep(int sweepIdx, long seed, int flags, double[] pOverride) {
        SweepStats st = new SweepStats();
        nSweep(handle, sweepIdx, seed, flags, pOverride, st);
        return st;
    }

    /** Multi-GPU: after the all-reduce of the delta buffer; (topic, modality) = winner of the MIN-reduced activation key. */
    public void applyDelta(int activatedTopic, int activatedModality) { nApplyDelta(handle, activatedTopic, activatedModality); }
    public double[] modelLogLikelihood(int numModalities) { double[] ll = new double[numModalities]; nModelLogLikelihood(handle, ll); return ll; }

    /** Safe from a finalizer or a shutdown hook: after the HIP runtime has gone the library frees host memory only. */
    @Override
    public void close() {
        if (handle != 0) { nDestroy(handle); handle = 0; }
    }

    private static native long nCreate(int numTopics, int[] numTypes, int device, long docIdBase);
    private static native void nDestroy(long h);
    private static native void nSetCorpus(long h, int m, long[] docOff, int[] tokens);
    private static native void nSetAssignments(long h, int m, int[] z);
    private static native void nGetAssignments(long h, int m, int[] z);
    private static native void nSetHyper(long h, double[][] alpha, double[] alphaSum, double[] beta, double[] betaSum,
                                         double[] gamma, double[][] p_a, double[][] p_b, boolean[] inactive);
    private static native void nBuildCounts(long h);
    private static native void nBuildTrees(long h);
    private static native void nGetCounts(long h, int m, int[] typeTopicCountsFlat, int[] tokensPerTopic);
    private static native void nGetDocTopicHist(long h, int m, int[] histFlat, int histLen, int[] docLengthCounts);
    private static native void nGetAlpha(long h, double[] alphaFlat, boolean[] inactive);
    private static native void nSweep(long h, int sweepIdx, long seed, int flags, double[] pOverride, SweepStats out);
    private static native void nApplyDelta(long h, int topic, int modality);
    private static native void nModelLogLikelihood(long h, double[] out);
}
