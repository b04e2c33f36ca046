// mvhdp_jni.cpp — JNI shim between org.madgik.MVTopicModel.NativeSampler and the C ABI
// of libmvhdp.so (include/mvhdp.h).  NOT compiled in the build image (no jni.h there):
//   g++ -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
//       mvtopicmodel_amd/java/mvhdp_jni.cpp -Lmvtopicmodel_amd/lib -lmvhdp -o libmvhdp_jni.so
// Arrays cross with Get/ReleasePrimitiveArrayCritical around the library's own copies;
// a negative status becomes a RuntimeException carrying mvhdp_last_error().
#include <jni.h>

#include <cstring>
#include <vector>

#include "mvhdp.h"

namespace {

void throw_rt(JNIEnv* env, mvhdp_handle h, int rc, const char* what)
{
    char msg[512];
    snprintf(msg, sizeof msg, "%s failed (%d): %s", what, rc, mvhdp_last_error(h));
    env->ThrowNew(env->FindClass("java/lang/RuntimeException"), msg);
}

struct Crit {   // RAII for GetPrimitiveArrayCritical
    JNIEnv* env; jarray arr; void* p; jint mode;
    Crit(JNIEnv* e, jarray a, jint m = 0) : env(e), arr(a), p(a ? e->GetPrimitiveArrayCritical(a, nullptr) : nullptr), mode(m) {}
    ~Crit() { if (arr) env->ReleasePrimitiveArrayCritical(arr, p, mode); }
};

inline mvhdp_handle H(jlong h) { return reinterpret_cast<mvhdp_handle>(h); }

}  // namespace

extern "C" {

JNIEXPORT jlong JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nCreate(JNIEnv* env, jclass, jint K, jintArray numTypes, jint device, jlong docIdBase)
{
    mvhdp_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.num_topics = K;
    cfg.num_modalities = env->GetArrayLength(numTypes);
    env->GetIntArrayRegion(numTypes, 0, cfg.num_modalities, cfg.num_types);
    cfg.device = device;
    cfg.doc_id_base = docIdBase;
    mvhdp_handle h = nullptr;
    int rc = mvhdp_create(&cfg, &h);
    if (rc != MVHDP_OK) { throw_rt(env, nullptr, rc, "mvhdp_create"); return 0; }
    return reinterpret_cast<jlong>(h);
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nDestroy(JNIEnv*, jclass, jlong h) { mvhdp_destroy(H(h)); }

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nSetCorpus(JNIEnv* env, jclass, jlong h, jint m, jlongArray docOff, jintArray tokens)
{
    jsize D = env->GetArrayLength(docOff) - 1;
    int rc;
    { Crit o(env, docOff, JNI_ABORT), t(env, tokens, JNI_ABORT);
      rc = mvhdp_set_corpus(H(h), m, D, static_cast<const int64_t*>(o.p), static_cast<const int32_t*>(t.p)); }
    if (rc) throw_rt(env, H(h), rc, "mvhdp_set_corpus");
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nSetAssignments(JNIEnv* env, jclass, jlong h, jint m, jintArray z)
{
    int rc;
    { Crit a(env, z, JNI_ABORT); rc = mvhdp_set_assignments(H(h), m, static_cast<const int32_t*>(a.p)); }
    if (rc) throw_rt(env, H(h), rc, "mvhdp_set_assignments");
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGetAssignments(JNIEnv* env, jclass, jlong h, jint m, jintArray z)
{
    int rc;
    { Crit a(env, z); rc = mvhdp_get_assignments(H(h), m, static_cast<int32_t*>(a.p)); }
    if (rc) throw_rt(env, H(h), rc, "mvhdp_get_assignments");
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nSetHyper(JNIEnv* env, jclass, jlong h, jobjectArray alpha, jdoubleArray alphaSum,
        jdoubleArray beta, jdoubleArray betaSum, jdoubleArray gamma, jobjectArray p_a, jobjectArray p_b, jbooleanArray inactive)
{
    const jsize M = env->GetArrayLength(alpha);
    jdoubleArray row0 = static_cast<jdoubleArray>(env->GetObjectArrayElement(alpha, 0));
    const jsize K1 = env->GetArrayLength(row0);           // K+1
    std::vector<double> a(static_cast<size_t>(M) * K1);
    mvhdp_hyper hy;
    std::memset(&hy, 0, sizeof hy);
    for (jsize m = 0; m < M; m++) {
        jdoubleArray r = static_cast<jdoubleArray>(env->GetObjectArrayElement(alpha, m));
        env->GetDoubleArrayRegion(r, 0, K1, a.data() + static_cast<size_t>(m) * K1);
        jdoubleArray pa = static_cast<jdoubleArray>(env->GetObjectArrayElement(p_a, m));
        jdoubleArray pb = static_cast<jdoubleArray>(env->GetObjectArrayElement(p_b, m));
        env->GetDoubleArrayRegion(pa, 0, M, hy.p_a[m]);
        env->GetDoubleArrayRegion(pb, 0, M, hy.p_b[m]);
    }
    env->GetDoubleArrayRegion(alphaSum, 0, M, hy.alpha_sum);
    env->GetDoubleArrayRegion(beta, 0, M, hy.beta);
    env->GetDoubleArrayRegion(betaSum, 0, M, hy.beta_sum);
    env->GetDoubleArrayRegion(gamma, 0, M, hy.gamma);
    std::vector<uint8_t> ina;
    if (inactive) {
        ina.resize(K1 - 1);
        env->GetBooleanArrayRegion(inactive, 0, K1 - 1, reinterpret_cast<jboolean*>(ina.data()));
        hy.inactive = ina.data();
    }
    hy.alpha = a.data();
    int rc = mvhdp_set_hyper(H(h), &hy);
    if (rc) throw_rt(env, H(h), rc, "mvhdp_set_hyper");
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nBuildCounts(JNIEnv* env, jclass, jlong h)
{ int rc = mvhdp_build_counts(H(h)); if (rc) throw_rt(env, H(h), rc, "mvhdp_build_counts"); }

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nBuildTrees(JNIEnv* env, jclass, jlong h)
{ int rc = mvhdp_build_trees(H(h)); if (rc) throw_rt(env, H(h), rc, "mvhdp_build_trees"); }

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGetCounts(JNIEnv* env, jclass, jlong h, jint m, jintArray nwk, jintArray nk)
{
    int rc;
    { Crit a(env, nwk), b(env, nk); rc = mvhdp_get_counts(H(h), m, static_cast<int32_t*>(a.p), static_cast<int32_t*>(b.p)); }
    if (rc) throw_rt(env, H(h), rc, "mvhdp_get_counts");
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGetDocTopicHist(JNIEnv* env, jclass, jlong h, jint m, jintArray hist, jint histLen, jintArray lens)
{
    int rc;
    { Crit a(env, hist), b(env, lens);
      rc = mvhdp_get_doc_topic_hist(H(h), m, static_cast<int32_t*>(a.p), histLen, static_cast<int32_t*>(b.p), lens ? env->GetArrayLength(lens) : 0); }
    if (rc) throw_rt(env, H(h), rc, "mvhdp_get_doc_topic_hist");
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nGetAlpha(JNIEnv* env, jclass, jlong h, jdoubleArray alphaFlat, jbooleanArray inactive)
{
    int rc;
    { Crit a(env, alphaFlat), b(env, inactive); rc = mvhdp_get_alpha(H(h), static_cast<double*>(a.p), static_cast<uint8_t*>(b.p)); }
    if (rc) throw_rt(env, H(h), rc, "mvhdp_get_alpha");
}

JNIEXPORT void JNICALL Java_org_madgik_MVTopicModel_NativeSampler_nSweep(JNIEnv* env, jclass, jlong h, jint sweepIdx, jlong seed, jint flags, jdoubleArray pOverride, jobject out)
{
    mvhdp_sweep_stats st;
    int rc;
    { Crit p(env, pOverride, JNI_ABORT);
      rc = mvhdp_sweep(H(h), static_cast<uint32_t>(sweepIdx), static_cast<uint64_t>(seed), static_cast<uint32_t>(flags),
                       static_cast<const double*>(p.p), nullptr, &st); }
    if (rc) { throw_rt(env, H(h), rc, "mvhdp_sweep"); return; }
    jclass c = env->GetObjectClass(out);
    auto setL = [&](const char* f, jlong v) { env->SetLongField(out, env->GetFieldID(c, f, "J"), v); };
    auto setI = [&](const char* f, jint v) { env->SetIntField(out, env->GetFieldID(c, f, "I"), v); };
    auto setD = [&](const char* f, jdouble v) { env->SetDoubleField(out, env->GetFieldID(c, f, "D"), v); };
    setL("tokens", st.tokens); setL("changed", st.changed); setL("newMassCnt", st.new_mass_cnt);
    setL("topicDocMassCnt", st.topic_doc_mass_cnt); setL("wordFTreeMassCnt", st.word_ftree_mass_cnt);
    setL("oovSkipped", st.oov_skipped); setL("abortedDocs", st.aborted_docs); setL("exactFallbacks", st.exact_fallbacks);
    setI("activatedTopic", st.activated_topic); setI("activatedModality", st.activated_modality);
    setL("activationKey", st.activation_key); setD("sweepKernelMs", st.sweep_kernel_ms); setD("totalMs", st.total_ms);
}

}  // extern "C"
