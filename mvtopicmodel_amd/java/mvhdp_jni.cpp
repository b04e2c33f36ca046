// mvhdp_jni.cpp — JNI shim between org.madgik.MVTopicModel.NativeSampler and the C ABI
// of libmvhdp.so (include/mvhdp.h).  NOT compiled in the build image (no JDK there):
//   g++ -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude
//       mvtopicmodel_amd/java/mvhdp_jni.cpp -Lmvtopicmodel_amd/lib -lmvhdp -o libmvhdp_jni.so      (one command line)
// (tests/test_jni_shim.py type-checks it against a declaration-only jni.h stub.)
//
// Arrays cross with Get<Type>ArrayElements / Release<Type>ArrayElements, never with GetPrimitiveArrayCritical: every
// mvhdp_* call may block (hipMalloc, synchronous copies, a whole sweep), and JNI forbids blocking -- or any other JNI
// call -- inside a critical region (it stalls the collector for every Java thread and can deadlock it).  The elements
// calls may copy; the library copies once more into HBM, and the Java arrays are free to move meanwhile.  Every array
// length is checked against the shape the handle was created with before the library sees a pointer.
// A negative status becomes a RuntimeException carrying mvhdp_last_error().
#include <jni.h>

#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <set>
#include <vector>

#include "mvhdp.h"

namespace {

struct Shard {                       // what the jlong handle points to: the library handle plus the shape for the checks
    mvhdp_handle h = nullptr;
    int K = 0, M = 0;
    int V[MVHDP_MAX_MODALITIES] = {};
    jlong D = -1;
    jlong N[MVHDP_MAX_MODALITIES] = {};
    int pins = 0;                    // JNI calls currently inside the library on this handle (guarded by g_reg_mutex)
};

// Every Shard (and group) the shim has handed to Java is listed here; a jlong that is not listed -- closed already, e.g. by a
// finalizer racing an explicit close() -- is refused instead of dereferenced.  A call that found its object PINS it for its
// duration: close() takes the object out of the registry at once (no new call can find it) and then waits until the calls already
// inside the library have returned before it frees anything -- a close() racing a sweep on another thread cannot pull the handle
// from under it.
std::mutex g_reg_mutex;
std::condition_variable g_reg_cv;
std::set<void*> g_shards, g_groups;

struct Group {                        // an mvhdp_group, how many members it has in this process, and the model's shape for the checks
    mvhdp_group g = nullptr;
    int members = 0;
    int K = 0, M = 0;                 // of the member Shards (validate_members: every member holds the same shape)
    std::vector<Shard*> shards;       // the members, PINNED from nGroupCreate to nGroupDestroy: nDestroy of a member (a finalizer, another thread)
                                      // waits for its pins, so a member cannot be destroyed under a group that still names it
    int pins = 0;
};

template <class T>
struct Pin {
    T* s = nullptr;
    Pin(std::set<void*>& reg, jlong p)
    {
        std::lock_guard<std::mutex> lk(g_reg_mutex);
        if (reg.count(reinterpret_cast<void*>(p))) { s = reinterpret_cast<T*>(p); s->pins++; }
    }
    ~Pin()
    {
        if (!s) return;
        std::lock_guard<std::mutex> lk(g_reg_mutex);
        if (--s->pins == 0) g_reg_cv.notify_all();
    }
    Pin(const Pin&) = delete;
    Pin& operator=(const Pin&) = delete;
};
struct ShardPin : Pin<Shard> { explicit ShardPin(jlong p) : Pin<Shard>(g_shards, p) {} };
struct GroupPin : Pin<Group> { explicit GroupPin(jlong p) : Pin<Group>(g_groups, p) {} };

// close(): out of the registry under the lock, then wait for the calls that are still inside
template <class T>
T* unregister_and_drain(std::set<void*>& reg, jlong p)
{
    std::unique_lock<std::mutex> lk(g_reg_mutex);
    auto it = reg.find(reinterpret_cast<void*>(p));
    if (it == reg.end()) return nullptr;
    T* s = reinterpret_cast<T*>(*it);
    reg.erase(it);
    g_reg_cv.wait(lk, [&] { return s->pins == 0; });
    return s;
}

void throw_msg(JNIEnv* env, const char* cls, const char* msg)
{
    jclass c = env->FindClass(cls);
    if (c) env->ThrowNew(c, msg);
}

void throw_rt(JNIEnv* env, mvhdp_handle h, int rc, const char* what)
{
    char msg[640];
    snprintf(msg, sizeof msg, "%s failed (%d): %s", what, rc, mvhdp_last_error(h));
    throw_msg(env, "java/lang/RuntimeException", msg);
}

bool bad_len(JNIEnv* env, jarray a, jlong want, const char* what)
{
    if (a && env->GetArrayLength(a) == want) return false;
    char msg[256];
    snprintf(msg, sizeof msg, "%s: array of length %lld expected, got %lld", what, (long long)want, a ? (long long)env->GetArrayLength(a) : -1LL);
    throw_msg(env, "java/lang/IllegalArgumentException", msg);
    return true;
}

// RAII over Get/Release<Type>ArrayElements (mode 0: copy back and free; JNI_ABORT: input only)
struct Ints {
    JNIEnv* env; jintArray a; jint* p; jint mode;
    Ints(JNIEnv* e, jintArray arr, jint m) : env(e), a(arr), p(arr ? e->GetIntArrayElements(arr, nullptr) : nullptr), mode(m) {}
    ~Ints() { if (a && p) env->ReleaseIntArrayElements(a, p, mode); }
    bool failed() const { return a && !p; }
};
struct Longs {
    JNIEnv* env; jlongArray a; jlong* p; jint mode;
    Longs(JNIEnv* e, jlongArray arr, jint m) : env(e), a(arr), p(arr ? e->GetLongArrayElements(arr, nullptr) : nullptr), mode(m) {}
    ~Longs() { if (a && p) env->ReleaseLongArrayElements(a, p, mode); }
    bool failed() const { return a && !p; }
};
struct Doubles {
    JNIEnv* env; jdoubleArray a; jdouble* p; jint mode;
    Doubles(JNIEnv* e, jdoubleArray arr, jint m) : env(e), a(arr), p(arr ? e->GetDoubleArrayElements(arr, nullptr) : nullptr), mode(m) {}
    ~Doubles() { if (a && p) env->ReleaseDoubleArrayElements(a, p, mode); }
    bool failed() const { return a && !p; }
};

static_assert(sizeof(jint) == sizeof(int32_t) && sizeof(jlong) == sizeof(int64_t) && sizeof(jdouble) == sizeof(double), "JNI primitive sizes");

}  // namespace

extern "C" {

JNIEXPORT jlong JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nCreate(JNIEnv* env, jclass, jint K, jintArray numTypes, jint device, jlong docIdBase)
{
    const jsize M = numTypes ? env->GetArrayLength(numTypes) : 0;
    if (M < 1 || M > MVHDP_MAX_MODALITIES) { throw_msg(env, "java/lang/IllegalArgumentException", "numTypes: 1..8 modalities"); return 0; }
    mvhdp_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.num_topics = K;
    cfg.num_modalities = M;
    env->GetIntArrayRegion(numTypes, 0, M, cfg.num_types);
    cfg.device = device;
    cfg.doc_id_base = docIdBase;
    Shard* s = new Shard();
    int rc = mvhdp_create(&cfg, &s->h);
    if (rc != MVHDP_OK) { delete s; throw_rt(env, nullptr, rc, "mvhdp_create"); return 0; }
    s->K = K; s->M = M;
    for (int m = 0; m < M; m++) s->V[m] = cfg.num_types[m];
    { std::lock_guard<std::mutex> lk(g_reg_mutex); g_shards.insert(s); }
    return reinterpret_cast<jlong>(s);
}

// Safe at any time, including from a finalizer or shutdown hook after the HIP runtime is gone (mvhdp_destroy then
// releases host memory only) and when called twice: the handle leaves the registry under the lock, a second call finds
// nothing and touches nothing.
JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nDestroy(JNIEnv*, jclass, jlong p)
{
    Shard* s = unregister_and_drain<Shard>(g_shards, p);
    if (!s) return;
    mvhdp_destroy(s->h);
    delete s;
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nSetCorpus(JNIEnv* env, jclass, jlong p, jint m, jlongArray docOff, jintArray tokens)
{
    ShardPin pin_(p); Shard* s = pin_.s;
    if (!s) { throw_msg(env, "java/lang/IllegalStateException", "NativeSampler is closed"); return; }
    if (m < 0 || m >= s->M || !docOff || env->GetArrayLength(docOff) < 1) { throw_msg(env, "java/lang/IllegalArgumentException", "setCorpus: bad view or docOff"); return; }
    const jsize D = env->GetArrayLength(docOff) - 1;
    int rc;
    jlong N = 0;
    {
        Longs o(env, docOff, JNI_ABORT);
        if (o.failed()) return;                                   // OutOfMemoryError pending
        N = o.p[D];
        if (N < 0 || (N > 0 && bad_len(env, tokens, N, "setCorpus tokens"))) return;
        Ints t(env, N > 0 ? tokens : nullptr, JNI_ABORT);
        if (t.failed()) return;
        rc = mvhdp_set_corpus(s->h, m, D, reinterpret_cast<const int64_t*>(o.p), reinterpret_cast<const int32_t*>(t.p));
    }
    if (rc) { throw_rt(env, s->h, rc, "mvhdp_set_corpus"); return; }
    s->D = D; s->N[m] = N;
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nSetAssignments(JNIEnv* env, jclass, jlong p, jint m, jintArray z)
{
    ShardPin pin_(p); Shard* s = pin_.s;
    if (!s) { throw_msg(env, "java/lang/IllegalStateException", "NativeSampler is closed"); return; }
    if (m < 0 || m >= s->M) { throw_msg(env, "java/lang/IllegalArgumentException", "setAssignments: bad view"); return; }
    if (s->N[m] > 0 && bad_len(env, z, s->N[m], "setAssignments")) return;
    int rc;
    { Ints a(env, s->N[m] > 0 ? z : nullptr, JNI_ABORT); if (a.failed()) return;
      rc = mvhdp_set_assignments(s->h, m, reinterpret_cast<const int32_t*>(a.p)); }
    if (rc) throw_rt(env, s->h, rc, "mvhdp_set_assignments");
}

// which entities HAVE the view (Assignments[m] != null) even when its FeatureSequence is empty: mvhdp_set_view_presence
JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nSetViewPresence(JNIEnv* env, jclass, jlong p, jint m, jbooleanArray present)
{
    ShardPin pin_(p); Shard* s = pin_.s;
    if (!s) { throw_msg(env, "java/lang/IllegalStateException", "NativeSampler is closed"); return; }
    if (m < 0 || m >= s->M) { throw_msg(env, "java/lang/IllegalArgumentException", "setViewPresence: bad view"); return; }
    int rc;
    if (!present) rc = mvhdp_set_view_presence(s->h, m, nullptr);
    else {
        if (bad_len(env, present, s->D, "setViewPresence")) return;
        std::vector<uint8_t> v(static_cast<size_t>(s->D));
        env->GetBooleanArrayRegion(present, 0, (jsize)s->D, reinterpret_cast<jboolean*>(v.data()));
        rc = mvhdp_set_view_presence(s->h, m, v.data());
    }
    if (rc) throw_rt(env, s->h, rc, "mvhdp_set_view_presence");
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGetAssignments(JNIEnv* env, jclass, jlong p, jint m, jintArray z)
{
    ShardPin pin_(p); Shard* s = pin_.s;
    if (!s) { throw_msg(env, "java/lang/IllegalStateException", "NativeSampler is closed"); return; }
    if (m < 0 || m >= s->M) { throw_msg(env, "java/lang/IllegalArgumentException", "getAssignments: bad view"); return; }
    if (s->N[m] == 0) return;
    if (bad_len(env, z, s->N[m], "getAssignments")) return;
    int rc;
    { Ints a(env, z, 0); if (a.failed()) return; rc = mvhdp_get_assignments(s->h, m, reinterpret_cast<int32_t*>(a.p)); }
    if (rc) throw_rt(env, s->h, rc, "mvhdp_get_assignments");
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nSetHyper(JNIEnv* env, jclass, jlong p, jobjectArray alpha, jdoubleArray alphaSum,
        jdoubleArray beta, jdoubleArray betaSum, jdoubleArray gamma, jobjectArray p_a, jobjectArray p_b, jbooleanArray inactive)
{
    ShardPin pin_(p); Shard* s = pin_.s;
    if (!s) { throw_msg(env, "java/lang/IllegalStateException", "NativeSampler is closed"); return; }
    const jsize M = s->M, K1 = s->K + 1;
    if (bad_len(env, alpha, M, "setHyper alpha") || bad_len(env, p_a, M, "setHyper p_a") || bad_len(env, p_b, M, "setHyper p_b") ||
        bad_len(env, alphaSum, M, "setHyper alphaSum") || bad_len(env, beta, M, "setHyper beta") ||
        bad_len(env, betaSum, M, "setHyper betaSum") || bad_len(env, gamma, M, "setHyper gamma")) return;
    if (inactive && bad_len(env, inactive, s->K, "setHyper inactive")) return;
    std::vector<double> a(static_cast<size_t>(M) * K1);
    mvhdp_hyper hy;
    std::memset(&hy, 0, sizeof hy);
    for (jsize m = 0; m < M; m++) {
        jdoubleArray r = static_cast<jdoubleArray>(env->GetObjectArrayElement(alpha, m));
        jdoubleArray pa = static_cast<jdoubleArray>(env->GetObjectArrayElement(p_a, m));
        jdoubleArray pb = static_cast<jdoubleArray>(env->GetObjectArrayElement(p_b, m));
        if (bad_len(env, r, K1, "setHyper alpha[m] (K+1 entries, PTM:196)") || bad_len(env, pa, M, "setHyper p_a[m]") || bad_len(env, pb, M, "setHyper p_b[m]")) return;
        env->GetDoubleArrayRegion(r, 0, K1, a.data() + static_cast<size_t>(m) * K1);
        env->GetDoubleArrayRegion(pa, 0, M, hy.p_a[m]);
        env->GetDoubleArrayRegion(pb, 0, M, hy.p_b[m]);
        env->DeleteLocalRef(r); env->DeleteLocalRef(pa); env->DeleteLocalRef(pb);
    }
    env->GetDoubleArrayRegion(alphaSum, 0, M, hy.alpha_sum);
    env->GetDoubleArrayRegion(beta, 0, M, hy.beta);
    env->GetDoubleArrayRegion(betaSum, 0, M, hy.beta_sum);
    env->GetDoubleArrayRegion(gamma, 0, M, hy.gamma);
    std::vector<uint8_t> ina;
    if (inactive) {
        ina.resize(static_cast<size_t>(s->K));
        env->GetBooleanArrayRegion(inactive, 0, s->K, reinterpret_cast<jboolean*>(ina.data()));
        hy.inactive = ina.data();
    }
    hy.alpha = a.data();
    int rc = mvhdp_set_hyper(s->h, &hy);
    if (rc) throw_rt(env, s->h, rc, "mvhdp_set_hyper");
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nBuildCounts(JNIEnv* env, jclass, jlong p)
{ ShardPin pin_(p); Shard* s = pin_.s; if (!s) { throw_msg(env, "java/lang/IllegalStateException", "NativeSampler is closed"); return; }
  int rc = mvhdp_build_counts(s->h); if (rc) throw_rt(env, s->h, rc, "mvhdp_build_counts"); }

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nBuildTrees(JNIEnv* env, jclass, jlong p)
{ ShardPin pin_(p); Shard* s = pin_.s; if (!s) { throw_msg(env, "java/lang/IllegalStateException", "NativeSampler is closed"); return; }
  int rc = mvhdp_build_trees(s->h); if (rc) throw_rt(env, s->h, rc, "mvhdp_build_trees"); }

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGetCounts(JNIEnv* env, jclass, jlong p, jint m, jintArray nwk, jintArray nk)
{
    ShardPin pin_(p); Shard* s = pin_.s;
    if (!s) { throw_msg(env, "java/lang/IllegalStateException", "NativeSampler is closed"); return; }
    if (m < 0 || m >= s->M) { throw_msg(env, "java/lang/IllegalArgumentException", "getCounts: bad view"); return; }
    if ((nwk && bad_len(env, nwk, (jlong)s->V[m] * s->K, "getCounts typeTopicCounts")) || (nk && bad_len(env, nk, s->K, "getCounts tokensPerTopic"))) return;
    int rc;
    { Ints a(env, nwk, 0), b(env, nk, 0); if (a.failed() || b.failed()) return;
      rc = mvhdp_get_counts(s->h, m, reinterpret_cast<int32_t*>(a.p), reinterpret_cast<int32_t*>(b.p)); }
    if (rc) throw_rt(env, s->h, rc, "mvhdp_get_counts");
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGetDocTopicHist(JNIEnv* env, jclass, jlong p, jint m, jintArray hist, jint histLen, jintArray lens)
{
    ShardPin pin_(p); Shard* s = pin_.s;
    if (!s) { throw_msg(env, "java/lang/IllegalStateException", "NativeSampler is closed"); return; }
    if (m < 0 || m >= s->M || histLen < 1) { throw_msg(env, "java/lang/IllegalArgumentException", "getDocTopicHist: bad view or length"); return; }
    if (hist && bad_len(env, hist, (jlong)s->K * histLen, "getDocTopicHist hist")) return;
    int rc;
    { Ints a(env, hist, 0), b(env, lens, 0); if (a.failed() || b.failed()) return;
      rc = mvhdp_get_doc_topic_hist(s->h, m, reinterpret_cast<int32_t*>(a.p), histLen, reinterpret_cast<int32_t*>(b.p), lens ? env->GetArrayLength(lens) : 0); }
    if (rc) throw_rt(env, s->h, rc, "mvhdp_get_doc_topic_hist");
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGetAlpha(JNIEnv* env, jclass, jlong p, jdoubleArray alphaFlat, jbooleanArray inactive)
{
    ShardPin pin_(p); Shard* s = pin_.s;
    if (!s) { throw_msg(env, "java/lang/IllegalStateException", "NativeSampler is closed"); return; }
    if (bad_len(env, alphaFlat, (jlong)s->M * (s->K + 1), "getAlpha alpha") || bad_len(env, inactive, s->K, "getAlpha inactive")) return;
    std::vector<double> a(static_cast<size_t>(s->M) * (s->K + 1));
    std::vector<uint8_t> ina(static_cast<size_t>(s->K));
    int rc = mvhdp_get_alpha(s->h, a.data(), ina.data());
    if (rc) { throw_rt(env, s->h, rc, "mvhdp_get_alpha"); return; }
    env->SetDoubleArrayRegion(alphaFlat, 0, (jsize)a.size(), a.data());
    env->SetBooleanArrayRegion(inactive, 0, s->K, reinterpret_cast<const jboolean*>(ina.data()));
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nSweep(JNIEnv* env, jclass, jlong p, jint sweepIdx, jlong seed, jint flags, jdoubleArray pOverride, jobject out)
{
    ShardPin pin_(p); Shard* s = pin_.s;
    if (!s) { throw_msg(env, "java/lang/IllegalStateException", "NativeSampler is closed"); return; }
    if (pOverride && bad_len(env, pOverride, s->D * s->M * s->M, "sweep pOverride [D][M][M]")) return;
    mvhdp_sweep_stats st;
    int rc;
    {
        // the view weights are copied out first; the sweep itself runs with no Java array held
        std::vector<double> pcopy;
        if (pOverride) { pcopy.resize(static_cast<size_t>(env->GetArrayLength(pOverride))); env->GetDoubleArrayRegion(pOverride, 0, (jsize)pcopy.size(), pcopy.data()); }
        rc = mvhdp_sweep(s->h, static_cast<uint32_t>(sweepIdx), static_cast<uint64_t>(seed), static_cast<uint32_t>(flags),
                         pOverride ? pcopy.data() : nullptr, nullptr, &st);
    }
    if (rc) { throw_rt(env, s->h, rc, "mvhdp_sweep"); return; }
    jclass c = env->GetObjectClass(out);
    auto setL = [&](const char* f, jlong v) { env->SetLongField(out, env->GetFieldID(c, f, "J"), v); };
    auto setI = [&](const char* f, jint v) { env->SetIntField(out, env->GetFieldID(c, f, "I"), v); };
    auto setD = [&](const char* f, jdouble v) { env->SetDoubleField(out, env->GetFieldID(c, f, "D"), v); };
    setL("tokens", st.tokens); setL("changed", st.changed); setL("newMassCnt", st.new_mass_cnt);
    setL("topicDocMassCnt", st.topic_doc_mass_cnt); setL("wordFTreeMassCnt", st.word_ftree_mass_cnt);
    setL("oovSkipped", st.oov_skipped); setL("abortedDocs", st.aborted_docs); setL("exactFallbacks", st.exact_fallbacks);
    setI("activatedTopic", st.activated_topic); setI("activatedModality", st.activated_modality);
    setL("activationKey", st.activation_key); setD("sweepKernelMs", st.sweep_kernel_ms); setD("totalMs", st.total_ms);
    setI("activations", st.activations);
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nApplyDelta(JNIEnv* env, jclass, jlong p, jint topic, jint modality)
{ ShardPin pin_(p); Shard* s = pin_.s; if (!s) { throw_msg(env, "java/lang/IllegalStateException", "NativeSampler is closed"); return; }
  int rc = mvhdp_apply_delta(s->h, topic, modality); if (rc) throw_rt(env, s->h, rc, "mvhdp_apply_delta"); }

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nModelLogLikelihood(JNIEnv* env, jclass, jlong p, jdoubleArray out)
{
    ShardPin pin_(p); Shard* s = pin_.s;
    if (!s) { throw_msg(env, "java/lang/IllegalStateException", "NativeSampler is closed"); return; }
    if (bad_len(env, out, s->M, "modelLogLikelihood")) return;
    double ll[MVHDP_MAX_MODALITIES];
    int rc = mvhdp_model_log_likelihood(s->h, ll);
    if (rc) { throw_rt(env, s->h, rc, "mvhdp_model_log_likelihood"); return; }
    env->SetDoubleArrayRegion(out, 0, s->M, ll);
}

// statistics of n sweeps as a flat long array, 8 per sweep: tokens, changed, newMassCnt, topicDocMassCnt, wordFTreeMassCnt,
// oovSkipped, abortedDocs, exactFallbacks
static void stats_to_longs(const mvhdp_sweep_stats& st, jlong* o)
{
    o[0] = st.tokens; o[1] = st.changed; o[2] = st.new_mass_cnt; o[3] = st.topic_doc_mass_cnt;
    o[4] = st.word_ftree_mass_cnt; o[5] = st.oov_skipped; o[6] = st.aborted_docs; o[7] = st.exact_fallbacks;
}

// the iteration loop PTM:1146-1239 without a host round trip per iteration (mvhdp_sweep_many)
JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nSweepMany(JNIEnv* env, jclass, jlong p, jint firstIdx, jint n, jlong seed, jint flags, jlongArray statsFlat)
{
    ShardPin pin_(p); Shard* s = pin_.s;
    if (!s) { throw_msg(env, "java/lang/IllegalStateException", "NativeSampler is closed"); return; }
    if (n < 0 || (statsFlat && bad_len(env, statsFlat, (jlong)n * 8, "sweepMany stats [n][8]"))) return;
    std::vector<mvhdp_sweep_stats> st(static_cast<size_t>(n > 0 ? n : 1));
    int rc = mvhdp_sweep_many(s->h, static_cast<uint32_t>(firstIdx), n, static_cast<uint64_t>(seed), static_cast<uint32_t>(flags), st.data());
    if (rc) { throw_rt(env, s->h, rc, "mvhdp_sweep_many"); return; }
    if (statsFlat) {
        std::vector<jlong> flat(static_cast<size_t>(n) * 8);
        for (int i = 0; i < n; i++) stats_to_longs(st[i], flat.data() + (size_t)i * 8);
        env->SetLongArrayRegion(statsFlat, 0, (jsize)flat.size(), flat.data());
    }
}

// tuning block (mvhdp_tuning): ints = {forcePrimary, narrow, walkFixed, singleStream, live16, learntStep0, learntStep1, learntStep2},
// doubles = {primaryMinShare, walkTheta[8], treeBranchShare[8]}
JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGetTuning(JNIEnv* env, jclass, jlong p, jintArray ints, jdoubleArray doubles)
{
    ShardPin pin_(p); Shard* s = pin_.s;
    if (!s) { throw_msg(env, "java/lang/IllegalStateException", "NativeSampler is closed"); return; }
    if (bad_len(env, ints, 8, "getTuning ints") || bad_len(env, doubles, 17, "getTuning doubles")) return;
    mvhdp_tuning t;
    int rc = mvhdp_get_tuning(s->h, &t);
    if (rc) { throw_rt(env, s->h, rc, "mvhdp_get_tuning"); return; }
    const jint iv[8] = {t.force_primary, t.narrow, t.walk_fixed, t.single_stream, t.live16, t.learnt_walk_step[0], t.learnt_walk_step[1], t.learnt_walk_step[2]};
    jdouble dv[17];
    dv[0] = t.primary_min_share;
    for (int m = 0; m < 8; m++) { dv[1 + m] = t.walk_theta[m]; dv[9 + m] = t.tree_branch_share[m]; }
    env->SetIntArrayRegion(ints, 0, 8, iv);
    env->SetDoubleArrayRegion(doubles, 0, 17, dv);
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nSetTuning(JNIEnv* env, jclass, jlong p, jintArray ints, jdoubleArray doubles)
{
    ShardPin pin_(p); Shard* s = pin_.s;
    if (!s) { throw_msg(env, "java/lang/IllegalStateException", "NativeSampler is closed"); return; }
    if (bad_len(env, ints, 8, "setTuning ints") || bad_len(env, doubles, 17, "setTuning doubles")) return;
    jint iv[8]; jdouble dv[17];
    env->GetIntArrayRegion(ints, 0, 8, iv);
    env->GetDoubleArrayRegion(doubles, 0, 17, dv);
    mvhdp_tuning t;
    std::memset(&t, 0, sizeof t);
    t.force_primary = iv[0]; t.narrow = iv[1]; t.walk_fixed = iv[2]; t.single_stream = iv[3]; t.live16 = iv[4];
    t.learnt_walk_step[0] = iv[5]; t.learnt_walk_step[1] = iv[6]; t.learnt_walk_step[2] = iv[7]; t.learnt_walk_step[3] = -1;
    t.live_overlap = -1;                                     // (the library's default: overlapped segments in live sweeps)
    t.primary_min_share = dv[0];
    for (int m = 0; m < 8; m++) { t.walk_theta[m] = dv[1 + m]; t.tree_branch_share[m] = dv[9 + m]; }
    int rc = mvhdp_set_tuning(s->h, &t);
    if (rc) throw_rt(env, s->h, rc, "mvhdp_set_tuning");
}

// ---- document shards on several GPUs: mvhdp_group_* (the reference's queue mesh + barrier, PTM:1042-1049, PTM:1232) ----
static void throw_group(JNIEnv* env, mvhdp_group g, int rc, const char* what)
{
    char msg[640];
    snprintf(msg, sizeof msg, "%s failed (%d): %s", what, rc, mvhdp_group_last_error(g));
    throw_msg(env, "java/lang/RuntimeException", msg);
}

// one JVM drives all its GPUs: one NativeSampler per device
JNIEXPORT jlong JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGroupCreate(JNIEnv* env, jclass, jlongArray handles)
{
    const jsize n = handles ? env->GetArrayLength(handles) : 0;
    if (n < 1) { throw_msg(env, "java/lang/IllegalArgumentException", "groupCreate: no members"); return 0; }
    std::vector<jlong> hp(static_cast<size_t>(n));
    env->GetLongArrayRegion(handles, 0, n, hp.data());
    std::vector<mvhdp_handle> hs;
    Group* gr = new Group();
    auto unpin_all = [&]() { std::lock_guard<std::mutex> lk(g_reg_mutex); for (Shard* s : gr->shards) if (--s->pins == 0) g_reg_cv.notify_all(); gr->shards.clear(); };
    for (jsize i = 0; i < n; i++) {
        Shard* s = nullptr;
        { std::lock_guard<std::mutex> lk(g_reg_mutex); if (g_shards.count(reinterpret_cast<void*>(hp[i]))) { s = reinterpret_cast<Shard*>(hp[i]); s->pins++; } }
        if (!s) { unpin_all(); delete gr; throw_msg(env, "java/lang/IllegalStateException", "groupCreate: a member is closed"); return 0; }
        gr->shards.push_back(s);
        hs.push_back(s->h);
    }
    int rc = mvhdp_group_create(n, hs.data(), &gr->g);
    if (rc != MVHDP_OK) { unpin_all(); delete gr; throw_group(env, nullptr, rc, "mvhdp_group_create"); return 0; }
    gr->members = n; gr->K = gr->shards[0]->K; gr->M = gr->shards[0]->M;
    { std::lock_guard<std::mutex> lk(g_reg_mutex); g_groups.insert(gr); }
    return reinterpret_cast<jlong>(gr);
}

// one JVM per GPU: rank 0 makes the 128-byte id, the launcher hands it to the other ranks
JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGroupUniqueId(JNIEnv* env, jclass, jbyteArray id)
{
    if (bad_len(env, id, MVHDP_UNIQUE_ID_BYTES, "groupUniqueId")) return;
    uint8_t buf[MVHDP_UNIQUE_ID_BYTES];
    int rc = mvhdp_group_unique_id(buf);
    if (rc) { throw_group(env, nullptr, rc, "mvhdp_group_unique_id"); return; }
    env->SetByteArrayRegion(id, 0, MVHDP_UNIQUE_ID_BYTES, reinterpret_cast<const jbyte*>(buf));
}

JNIEXPORT jlong JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGroupCreateRank(JNIEnv* env, jclass, jlong p, jbyteArray id, jint rank, jint nranks)
{
    Shard* s = nullptr;
    { std::lock_guard<std::mutex> lk(g_reg_mutex); if (g_shards.count(reinterpret_cast<void*>(p))) { s = reinterpret_cast<Shard*>(p); s->pins++; } }   // (kept until nGroupDestroy)
    if (!s) { throw_msg(env, "java/lang/IllegalStateException", "NativeSampler is closed"); return 0; }
    auto unpin = [&]() { std::lock_guard<std::mutex> lk(g_reg_mutex); if (--s->pins == 0) g_reg_cv.notify_all(); };
    if (bad_len(env, id, MVHDP_UNIQUE_ID_BYTES, "groupCreateRank id")) { unpin(); return 0; }
    uint8_t buf[MVHDP_UNIQUE_ID_BYTES];
    env->GetByteArrayRegion(id, 0, MVHDP_UNIQUE_ID_BYTES, reinterpret_cast<jbyte*>(buf));
    Group* gr = new Group();
    int rc = mvhdp_group_create_rank(s->h, buf, rank, nranks, &gr->g);
    if (rc != MVHDP_OK) { unpin(); delete gr; throw_group(env, nullptr, rc, "mvhdp_group_create_rank"); return 0; }
    gr->members = 1; gr->K = s->K; gr->M = s->M; gr->shards.push_back(s);
    { std::lock_guard<std::mutex> lk(g_reg_mutex); g_groups.insert(gr); }
    return reinterpret_cast<jlong>(gr);
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGroupDestroy(JNIEnv*, jclass, jlong p)
{
    Group* gr = unregister_and_drain<Group>(g_groups, p);
    if (!gr) return;
    mvhdp_group_destroy(gr->g);
    { std::lock_guard<std::mutex> lk(g_reg_mutex); for (Shard* s : gr->shards) if (--s->pins == 0) g_reg_cv.notify_all(); }   // the members may be closed now
    delete gr;
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGroupBuildCounts(JNIEnv* env, jclass, jlong p)
{
    GroupPin gpin_(p); Group* gr = gpin_.s;
    if (!gr) { throw_msg(env, "java/lang/IllegalStateException", "group is closed"); return; }
    int rc = mvhdp_group_build_counts(gr->g);
    if (rc) throw_group(env, gr->g, rc, "mvhdp_group_build_counts");
}

// statsFlat: [members][8] as nSweepMany; act: {activatedTopic, activatedModality, activations} of the sweep (the same on every replica);
// returns the device milliseconds of the exchange (collectives + updates + tree rebuilds)
JNIEXPORT jdouble JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGroupSweep(JNIEnv* env, jclass, jlong p, jint sweepIdx, jlong seed, jint flags, jlongArray statsFlat, jintArray act)
{
    GroupPin gpin_(p); Group* gr = gpin_.s;
    if (!gr) { throw_msg(env, "java/lang/IllegalStateException", "group is closed"); return 0.0; }
    if ((statsFlat && bad_len(env, statsFlat, (jlong)gr->members * 8, "groupSweep stats [members][8]")) || (act && bad_len(env, act, 3, "groupSweep act"))) return 0.0;
    std::vector<mvhdp_sweep_stats> st(static_cast<size_t>(gr->members));
    int rc = mvhdp_group_sweep(gr->g, static_cast<uint32_t>(sweepIdx), static_cast<uint64_t>(seed), static_cast<uint32_t>(flags), st.data());
    if (rc) { throw_group(env, gr->g, rc, "mvhdp_group_sweep"); return 0.0; }
    if (statsFlat) {
        std::vector<jlong> flat(static_cast<size_t>(gr->members) * 8);
        for (int i = 0; i < gr->members; i++) stats_to_longs(st[i], flat.data() + (size_t)i * 8);
        env->SetLongArrayRegion(statsFlat, 0, (jsize)flat.size(), flat.data());
    }
    if (act) { const jint a[3] = {st[0].activated_topic, st[0].activated_modality, st[0].activations}; env->SetIntArrayRegion(act, 0, 3, a); }
    mvhdp_group_info info;
    return mvhdp_group_get_info(gr->g, &info) == MVHDP_OK ? info.last_exchange_ms : 0.0;
}

// ---- the steps either side of the sweep: one handle, and a sharded model (mvhdp_group_*) ----
// countHistogram of optimizeBeta PTM:2295-2309
JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGetCountHistogram(JNIEnv* env, jclass, jlong p, jint m, jintArray hist)
{
    ShardPin pin_(p); Shard* s = pin_.s;
    if (!s) { throw_msg(env, "java/lang/IllegalStateException", "NativeSampler is closed"); return; }
    if (m < 0 || m >= s->M || !hist || env->GetArrayLength(hist) < 1) { throw_msg(env, "java/lang/IllegalArgumentException", "getCountHistogram: bad view or array"); return; }
    int rc;
    { Ints a(env, hist, 0); if (a.failed()) return; rc = mvhdp_get_count_histogram(s->h, m, reinterpret_cast<int32_t*>(a.p), env->GetArrayLength(hist)); }
    if (rc) throw_rt(env, s->h, rc, "mvhdp_get_count_histogram");
}

// optimizeP PTM:2706-2792: sums [M*M]
JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nViewOverlapSums(JNIEnv* env, jclass, jlong p, jdoubleArray sums)
{
    ShardPin pin_(p); Shard* s = pin_.s;
    if (!s) { throw_msg(env, "java/lang/IllegalStateException", "NativeSampler is closed"); return; }
    if (bad_len(env, sums, (jlong)s->M * s->M, "viewOverlapSums")) return;
    double v[MVHDP_MAX_MODALITIES * MVHDP_MAX_MODALITIES];
    int rc = mvhdp_view_overlap_sums(s->h, v);
    if (rc) { throw_rt(env, s->h, rc, "mvhdp_view_overlap_sums"); return; }
    env->SetDoubleArrayRegion(sums, 0, s->M * s->M, v);
}

// optimizeGamma's document level PTM:2415-2433: out = {qs, qw}
JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGammaDocStatistics(JNIEnv* env, jclass, jlong p, jint m, jdouble gammaM, jlong seed, jint round, jdoubleArray out)
{
    ShardPin pin_(p); Shard* s = pin_.s;
    if (!s) { throw_msg(env, "java/lang/IllegalStateException", "NativeSampler is closed"); return; }
    if (bad_len(env, out, 2, "gammaDocStatistics")) return;
    double v[2];
    int rc = mvhdp_gamma_doc_statistics(s->h, m, gammaM, static_cast<uint64_t>(seed), static_cast<uint32_t>(round), &v[0], &v[1]);
    if (rc) { throw_rt(env, s->h, rc, "mvhdp_gamma_doc_statistics"); return; }
    env->SetDoubleArrayRegion(out, 0, 2, v);
}

// optimizeDP's view-table simulation PTM:2454-2488: hist [K][histLen] (of the whole model), conc [K] = gamma[m] * alpha[m][t]; mk [K], active [K] filled
JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nDpTableStatistics(JNIEnv* env, jclass, jlong p, jint m, jintArray hist, jint histLen, jdoubleArray conc,
                                                                                      jlong seed, jint round, jdoubleArray mk, jbyteArray active)
{
    ShardPin pin_(p); Shard* s = pin_.s;
    if (!s) { throw_msg(env, "java/lang/IllegalStateException", "NativeSampler is closed"); return; }
    if (m < 0 || m >= s->M || histLen < 1) { throw_msg(env, "java/lang/IllegalArgumentException", "dpTableStatistics: bad view or histLen"); return; }
    if (bad_len(env, hist, (jlong)s->K * histLen, "dpTableStatistics hist [K][histLen]") || bad_len(env, conc, s->K, "dpTableStatistics conc") ||
        bad_len(env, mk, s->K, "dpTableStatistics mk") || bad_len(env, active, s->K, "dpTableStatistics active")) return;
    std::vector<double> cv(static_cast<size_t>(s->K)), mv(static_cast<size_t>(s->K));
    std::vector<uint8_t> av(static_cast<size_t>(s->K));
    env->GetDoubleArrayRegion(conc, 0, s->K, cv.data());
    int rc;
    { Ints a(env, hist, 0); if (a.failed()) return;
      rc = mvhdp_dp_table_statistics(s->h, m, reinterpret_cast<const int32_t*>(a.p), histLen, cv.data(), static_cast<uint64_t>(seed), static_cast<uint32_t>(round), mv.data(), av.data()); }
    if (rc) { throw_rt(env, s->h, rc, "mvhdp_dp_table_statistics"); return; }
    env->SetDoubleArrayRegion(mk, 0, s->K, mv.data());
    env->SetByteArrayRegion(active, 0, s->K, reinterpret_cast<const jbyte*>(av.data()));
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGroupDrain(JNIEnv* env, jclass, jlong p)
{
    GroupPin gpin_(p); Group* gr = gpin_.s;
    if (!gr) { throw_msg(env, "java/lang/IllegalStateException", "group is closed"); return; }
    int rc = mvhdp_group_drain(gr->g);
    if (rc) throw_group(env, gr->g, rc, "mvhdp_group_drain");
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGroupAbort(JNIEnv* env, jclass, jlong p)
{
    GroupPin gpin_(p); Group* gr = gpin_.s;
    if (!gr) { throw_msg(env, "java/lang/IllegalStateException", "group is closed"); return; }
    int rc = mvhdp_group_abort(gr->g);
    if (rc) throw_group(env, gr->g, rc, "mvhdp_group_abort");
}

// modelLogLikelihood PTM:3322-3452 of the sharded model: out [M]
JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGroupModelLogLikelihood(JNIEnv* env, jclass, jlong p, jdoubleArray out)
{
    GroupPin gpin_(p); Group* gr = gpin_.s;
    if (!gr) { throw_msg(env, "java/lang/IllegalStateException", "group is closed"); return; }
    if (bad_len(env, out, gr->M, "groupModelLogLikelihood: one entry per view")) return;
    double ll[MVHDP_MAX_MODALITIES] = {};
    int rc = mvhdp_group_log_likelihood(gr->g, ll);
    if (rc) { throw_group(env, gr->g, rc, "mvhdp_group_log_likelihood"); return; }
    env->SetDoubleArrayRegion(out, 0, env->GetArrayLength(out), ll);
}

// topicDocCounts / docLengthCounts over every entity of every member: hist [K*histLen]
JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGroupGetDocTopicHist(JNIEnv* env, jclass, jlong p, jint m, jintArray hist, jint histLen, jintArray lens)
{
    GroupPin gpin_(p); Group* gr = gpin_.s;
    if (!gr) { throw_msg(env, "java/lang/IllegalStateException", "group is closed"); return; }
    // (the library zero-fills and sums K * histLen entries whatever it is handed: the shape is checked HERE, against the members' K and M)
    if (m < 0 || m >= gr->M || histLen < 1) { throw_msg(env, "java/lang/IllegalArgumentException", "groupGetDocTopicHist: bad view or histLen"); return; }
    if (hist && bad_len(env, hist, (jlong)gr->K * histLen, "groupGetDocTopicHist hist [K][histLen]")) return;
    int rc;
    { Ints a(env, hist, 0), b(env, lens, 0); if (a.failed() || b.failed()) return;
      rc = mvhdp_group_doc_topic_hist(gr->g, m, reinterpret_cast<int32_t*>(a.p), histLen, reinterpret_cast<int32_t*>(b.p), lens ? env->GetArrayLength(lens) : 0); }
    if (rc) throw_group(env, gr->g, rc, "mvhdp_group_doc_topic_hist");
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGroupGetCountHistogram(JNIEnv* env, jclass, jlong p, jint m, jintArray hist)
{
    GroupPin gpin_(p); Group* gr = gpin_.s;
    if (!gr) { throw_msg(env, "java/lang/IllegalStateException", "group is closed"); return; }
    if (m < 0 || m >= gr->M || !hist || env->GetArrayLength(hist) < 1) { throw_msg(env, "java/lang/IllegalArgumentException", "groupGetCountHistogram: bad view or empty array"); return; }
    int rc;
    { Ints a(env, hist, 0); if (a.failed()) return; rc = mvhdp_group_count_histogram(gr->g, m, reinterpret_cast<int32_t*>(a.p), env->GetArrayLength(hist)); }
    if (rc) throw_group(env, gr->g, rc, "mvhdp_group_count_histogram");
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGroupViewOverlapSums(JNIEnv* env, jclass, jlong p, jdoubleArray sums)
{
    GroupPin gpin_(p); Group* gr = gpin_.s;
    if (!gr) { throw_msg(env, "java/lang/IllegalStateException", "group is closed"); return; }
    const jsize n = gr->M * gr->M;
    if (bad_len(env, sums, n, "groupViewOverlapSums [M][M]")) return;
    double v[MVHDP_MAX_MODALITIES * MVHDP_MAX_MODALITIES] = {};
    int rc = mvhdp_group_view_overlap_sums(gr->g, v);
    if (rc) { throw_group(env, gr->g, rc, "mvhdp_group_view_overlap_sums"); return; }
    env->SetDoubleArrayRegion(sums, 0, n, v);
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGroupGammaDocStatistics(JNIEnv* env, jclass, jlong p, jint m, jdouble gammaM, jlong seed, jint round, jdoubleArray out)
{
    GroupPin gpin_(p); Group* gr = gpin_.s;
    if (!gr) { throw_msg(env, "java/lang/IllegalStateException", "group is closed"); return; }
    if (m < 0 || m >= gr->M) { throw_msg(env, "java/lang/IllegalArgumentException", "groupGammaDocStatistics: bad view"); return; }
    if (bad_len(env, out, 2, "groupGammaDocStatistics")) return;
    double v[2];
    int rc = mvhdp_group_gamma_doc_statistics(gr->g, m, gammaM, static_cast<uint64_t>(seed), static_cast<uint32_t>(round), &v[0], &v[1]);
    if (rc) { throw_group(env, gr->g, rc, "mvhdp_group_gamma_doc_statistics"); return; }
    env->SetDoubleArrayRegion(out, 0, 2, v);
}

}  // extern "C"
