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
    public static final int SWEEP_ASYNC_EXCHANGE = 0x100;  // Group.sweep with SWEEP_LIVE: the all-reduce of sweep t beside sweep t+1 (then Group.drain())
    public static final int SWEEP_SEGMENT_OVERLAP = 0x80; // with SEGMENT_APPLY: the deltas of segment s are applied while segment s+1 samples (s+2 sees them)
    public static int sweepLiveSegments(int n) { return (n & 0xff) << 16; }
    public static int sweepOnlySegment(int s) { return ((s + 1) & 0xff) << 24; }   // only segment s of the n segments

    /** What one sweep reports: the three branch counters of the worker plus bookkeeping. */
    public static final class SweepStats {
        public long tokens, changed, newMassCnt, topicDocMassCnt, wordFTreeMassCnt, oovSkipped, abortedDocs, exactFallbacks;
        public int activatedTopic, activatedModality;
        public long activationKey;
        public double sweepKernelMs, totalMs;
        public int activations;          // topics activated during the call (a segmented sweep: one per segment border; a live sweep in its live-rows form: chunk by chunk, any number)
    }

    private long handle;   // mvhdp_handle

    public NativeSampler(int numTopics, int[] numTypes, int device, long docIdBase) {
        handle = nCreate(numTopics, numTypes, device, docIdBase);
    }

    /** tokens = FeatureSequence indices of every entity's view m, docOff = CSR offsets (size by getLength()). */
    public void setCorpus(int m, long[] docOff, int[] tokens) { nSetCorpus(handle, m, docOff, tokens); }
    public void setAssignments(int m, int[] z) { nSetAssignments(handle, m, z); }
    public void getAssignments(int m, int[] z) { nGetAssignments(handle, m, z); }
    /** present[d] = data.get(d).Assignments[m] != null: an instance with an EMPTY FeatureSequence is not a missing view (lines 3348-3373, 2873-2886). */
    public void setViewPresence(int m, boolean[] present) { nSetViewPresence(handle, m, present); }

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
    /** countHistogram of optimizeBeta (lines 2295-2309): hist[c] = (type, topic) pairs of view m holding count c. */
    public void getCountHistogram(int m, int[] hist) { nGetCountHistogram(handle, m, hist); }
    /** optimizeP (lines 2706-2792): sums[m*M+i] = sum over the entities, in entity order, of pDistr_Mean[m][i][doc]. */
    public double[] viewOverlapSums(int numModalities) { double[] s = new double[numModalities * numModalities]; nViewOverlapSums(handle, s); return s; }
    /** optimizeGamma's document level (lines 2415-2433): {qs, qw}, every entity drawing from its own counter-based stream. */
    public double[] gammaDocStatistics(int m, double gammaM, long seed, int round) { double[] o = new double[2]; nGammaDocStatistics(handle, m, gammaM, seed, round, o); return o; }
    /** optimizeDP's view-table simulation PTM:2454-2488 over topicDocCounts[m] (hist [K][histLen], of the whole model): mk [K] and active [K] are filled. */
    public void dpTableStatistics(int m, int[] hist, int histLen, double[] conc, long seed, int round, double[] mk, byte[] active) { nDpTableStatistics(handle, m, hist, histLen, conc, seed, round, mk, active); }

    /**
     * n sweeps (indices firstIdx .. firstIdx+n-1) enqueued back to back, one synchronisation at the end: the iteration loop
     * of estimate() (lines 1146-1239) without a host round trip per iteration.  Same integers as n calls of sweep().
     */
    public SweepStats[] sweepMany(int firstIdx, int n, long seed, int flags) {
        long[] flat = new long[n * 8];
        nSweepMany(handle, firstIdx, n, seed, flags, flat);
        SweepStats[] out = new SweepStats[n];
        for (int i = 0; i < n; i++) out[i] = statsFromFlat(flat, i);
        return out;
    }

    static SweepStats statsFromFlat(long[] flat, int i) {
        SweepStats st = new SweepStats();
        st.tokens = flat[8 * i]; st.changed = flat[8 * i + 1]; st.newMassCnt = flat[8 * i + 2]; st.topicDocMassCnt = flat[8 * i + 3];
        st.wordFTreeMassCnt = flat[8 * i + 4]; st.oovSkipped = flat[8 * i + 5]; st.abortedDocs = flat[8 * i + 6]; st.exactFallbacks = flat[8 * i + 7];
        return st;
    }

    /**
     * The sweep's own choices (mvhdp_tuning): none of them changes a result.  learntWalkStep / treeBranchShare are what the
     * library's walk-threshold search has found -- read them from one sampler and hand them to another (a document shard, a
     * resumed chain) and it does not search again.
     */
    public static final class Tuning {
        public int forcePrimary, narrow = -1, walkFixed, singleStream, live16 = -1;
        public int[] learntWalkStep = {-1, -1, -1};
        public double primaryMinShare;
        public double[] walkTheta = new double[8], treeBranchShare = new double[8];
    }

    public Tuning getTuning() {
        int[] iv = new int[8]; double[] dv = new double[17];
        nGetTuning(handle, iv, dv);
        Tuning t = new Tuning();
        t.forcePrimary = iv[0]; t.narrow = iv[1]; t.walkFixed = iv[2]; t.singleStream = iv[3]; t.live16 = iv[4];
        t.learntWalkStep = new int[] {iv[5], iv[6], iv[7]};
        t.primaryMinShare = dv[0];
        System.arraycopy(dv, 1, t.walkTheta, 0, 8); System.arraycopy(dv, 9, t.treeBranchShare, 0, 8);
        return t;
    }

    public void setTuning(Tuning t) {
        int[] iv = {t.forcePrimary, t.narrow, t.walkFixed, t.singleStream, t.live16, t.learntWalkStep[0], t.learntWalkStep[1], t.learntWalkStep[2]};
        double[] dv = new double[17];
        dv[0] = t.primaryMinShare;
        System.arraycopy(t.walkTheta, 0, dv, 1, 8); System.arraycopy(t.treeBranchShare, 0, dv, 9, 8);
        nSetTuning(handle, iv, dv);
    }

    /**
     * Document shards on several GPUs with the exchange step inside the library (mvhdp_group_*): what the queue mesh between
     * sampler and updater threads and the CyclicBarrier do inside the reference's one JVM (lines 1042-1049, 1232).  Every member
     * holds a contiguous range of entities (docIdBase = global index of its first entity) and a full replica of the model.
     */
    public static final class Group implements AutoCloseable {
        private long g;
        private final int members;

        /** One JVM drives all its GPUs: one NativeSampler per device. */
        public Group(NativeSampler[] samplers) {
            long[] hs = new long[samplers.length];
            for (int i = 0; i < hs.length; i++) hs[i] = samplers[i].handle;
            g = nGroupCreate(hs);
            members = hs.length;
        }

        private Group(long g) { this.g = g; this.members = 1; }

        /** One JVM per GPU: rank 0 calls uniqueId(), the launcher hands the 128 bytes to every rank, every rank calls this. */
        public static Group ofRank(NativeSampler sampler, byte[] id, int rank, int nranks) {
            return new Group(nGroupCreateRank(sampler.handle, id, rank, nranks));
        }

        public static byte[] uniqueId() { byte[] id = new byte[128]; nGroupUniqueId(id); return id; }

        /** buildInitialTypeTopicCounts (lines 600-652) over all shards. */
        public void buildCounts() { nGroupBuildCounts(g); }

        /** One sweep of the whole model; flags: SWEEP_LIVE / SWEEP_SEGMENT_APPLY (+ sweepLiveSegments), SWEEP_EXACT_CHAIN. */
        public SweepStats[] sweep(int sweepIdx, long seed, int flags) {
            long[] flat = new long[members * 8];
            int[] act = new int[3];
            double exchangeMs = nGroupSweep(g, sweepIdx, seed, flags, flat, act);
            SweepStats[] out = new SweepStats[members];
            for (int i = 0; i < members; i++) {
                out[i] = statsFromFlat(flat, i);
                out[i].activatedTopic = act[0]; out[i].activatedModality = act[1]; out[i].activations = act[2];
                out[i].totalMs = exchangeMs;
            }
            return out;
        }

        /** After SWEEP_ASYNC_EXCHANGE sweeps: lands what is on the wire, every replica is the global model again. */
        public void drain() { nGroupDrain(g); }

        /** This rank cannot go on: its next sweep contributes nothing and fails on EVERY rank together (nobody waits in a collective). */
        public void abort() { nGroupAbort(g); }

        // The steps either side of the sweep for the sharded model (what estimate() does every optimizeInterval and every tenth
        // iteration, lines 1173-1210 and 1296-1320): statistics of the replicated counts are any member's, statistics over the
        // entities are put together from the members in entity order.  setHyper: call NativeSampler.setHyper on every member.
        public double[] modelLogLikelihood(int numModalities) { double[] ll = new double[numModalities]; nGroupModelLogLikelihood(g, ll); return ll; }
        public void getDocTopicHist(int m, int[] histFlat, int histLen, int[] docLengthCounts) { nGroupGetDocTopicHist(g, m, histFlat, histLen, docLengthCounts); }
        public void getCountHistogram(int m, int[] hist) { nGroupGetCountHistogram(g, m, hist); }
        public double[] viewOverlapSums(int numModalities) { double[] s = new double[numModalities * numModalities]; nGroupViewOverlapSums(g, s); return s; }
        public double[] gammaDocStatistics(int m, double gammaM, long seed, int round) { double[] o = new double[2]; nGroupGammaDocStatistics(g, m, gammaM, seed, round, o); return o; }

        @Override
        public void close() { if (g != 0) { nGroupDestroy(g); g = 0; } }
    }

    /**
     * Safe from a finalizer or a shutdown hook (after the HIP runtime has gone the library frees host memory only) and when it
     * races a second close: the native side drops the handle from its registry under a lock and ignores a handle it does not know.
     */
    @Override
    public synchronized void close() {
        if (handle != 0) { long h = handle; handle = 0; nDestroy(h); }
    }

    private static native long nCreate(int numTopics, int[] numTypes, int device, long docIdBase);
    private static native void nDestroy(long h);
    private static native void nSetCorpus(long h, int m, long[] docOff, int[] tokens);
    private static native void nSetAssignments(long h, int m, int[] z);
    private static native void nGetAssignments(long h, int m, int[] z);
    private static native void nSetViewPresence(long h, int m, boolean[] present);
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
    private static native void nSweepMany(long h, int firstIdx, int n, long seed, int flags, long[] statsFlat);
    private static native void nGetTuning(long h, int[] ints, double[] doubles);
    private static native void nSetTuning(long h, int[] ints, double[] doubles);
    private static native long nGroupCreate(long[] handles);
    private static native void nGroupUniqueId(byte[] id);
    private static native long nGroupCreateRank(long h, byte[] id, int rank, int nranks);
    private static native void nGroupDestroy(long g);
    private static native void nGroupBuildCounts(long g);
    private static native double nGroupSweep(long g, int sweepIdx, long seed, int flags, long[] statsFlat, int[] act);
    private static native void nGetCountHistogram(long h, int m, int[] hist);
    private static native void nViewOverlapSums(long h, double[] sums);
    private static native void nGammaDocStatistics(long h, int m, double gammaM, long seed, int round, double[] out);
    private static native void nDpTableStatistics(long h, int m, int[] hist, int histLen, double[] conc, long seed, int round, double[] mk, byte[] active);
    private static native void nGroupAbort(long g);
    private static native void nGroupDrain(long g);
    private static native void nGroupModelLogLikelihood(long g, double[] out);
    private static native void nGroupGetDocTopicHist(long g, int m, int[] histFlat, int histLen, int[] docLengthCounts);
    private static native void nGroupGetCountHistogram(long g, int m, int[] hist);
    private static native void nGroupViewOverlapSums(long g, double[] sums);
    private static native void nGroupGammaDocStatistics(long g, int m, double gammaM, long seed, int round, double[] out);
}
