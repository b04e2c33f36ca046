// FastQMVWVTopicInferencer.cpp — see the header.  Host logic only.
#include "FastQMVWVTopicInferencer.h"

#include <algorithm>
#include <cstring>
#include <random>
#include <stdexcept>
#include <unordered_map>

namespace mvtm {

std::string formatDocumentTopics(mvhdp_handle h, const std::vector<std::string>& names, int K, const std::vector<double>& w,
                                 double threshold, int max)
{
    if (max < 0 || max > K) max = K;                                                 // PTM:2834-2836, INF:346-348
    std::string out = "#doc name topic proportion ...\n";                            // PTM:2823, INF:334
    const int64_t D = (int64_t)names.size();
    const int64_t batch = std::max<int64_t>(1, (int64_t)(64 << 20) / (K * 8));      // 64 MB of proportions at a time
    std::vector<double> prop;
    std::vector<int> order((size_t)K);
    for (int64_t d0 = 0; d0 < D; d0 += batch) {
        const int64_t d1 = std::min(D, d0 + batch);
        prop.assign((size_t)(d1 - d0) * K, 0.0);
        int rc = mvhdp_doc_topic_proportions(h, w.data(), d0, d1, prop.data());
        if (rc != MVHDP_OK) throw std::runtime_error(std::string("mvhdp_doc_topic_proportions: ") + mvhdp_last_error(h));
        for (int64_t doc = d0; doc < d1; doc++) {
            const double* pr = prop.data() + (size_t)(doc - d0) * K;
            for (int k = 0; k < K; k++) order[k] = k;
            // Arrays.sort(IDSorter[]) PTM:2902 with MALLET 2.0.8's IDSorter.compareTo (class file): descending weight,
            // equal weights by DESCENDING id
            std::sort(order.begin(), order.end(), [&](int a, int b) { return pr[a] > pr[b] || (pr[a] == pr[b] && a > b); });
            std::string builder = std::to_string(doc) + "\t" + names[(size_t)doc] + "\t";     // PTM:2862-2869
            for (int i = 0; i < max; i++) {
                if (pr[order[i]] < threshold) break;                                  // PTM:2905
                builder += std::to_string(order[i]) + "\t" + FastQMVWVParallelTopicModel::javaDoubleToString(pr[order[i]]) + "\t";
                out += builder; out += "\n";                                          // PTM:2909: the whole builder, every time
            }
        }
    }
    return out;
}

FastQMVWVTopicInferencer::FastQMVWVTopicInferencer(const std::vector<int>& numTypes_, const std::vector<std::vector<double>>& alpha_,
                                                   const std::vector<double>& alphaSum_, const std::vector<std::vector<int32_t>>& typeTopicCounts_,
                                                   const std::vector<std::vector<int32_t>>& tokensPerTopic_, const std::vector<double>& beta_,
                                                   const std::vector<double>& betaSum_, const std::vector<double>& gamma_, int numTopics_,
                                                   int8_t numModalities_, const std::vector<std::vector<double>>& p_a_,
                                                   const std::vector<std::vector<double>>& p_b_, const std::vector<double>& discrWeightPerModality_,
                                                   const std::vector<std::vector<double>>& pMean_)
    : numTopics(numTopics_), numModalities(numModalities_), numTypes(numTypes_), alpha(alpha_), alphaSum(alphaSum_), beta(beta_),
      betaSum(betaSum_), gamma(gamma_), typeTopicCounts(typeTopicCounts_), tokensPerTopic(tokensPerTopic_), p_a(p_a_), p_b(p_b_),
      pMean(pMean_), discrWeightPerModality(discrWeightPerModality_)
{
    const int M = numModalities;
    if ((int)numTypes.size() != M || (int)alpha.size() != M || (int)typeTopicCounts.size() != M || (int)tokensPerTopic.size() != M)
        throw std::invalid_argument("FastQMVWVTopicInferencer: per-modality arrays do not match numModalities");
    for (int m = 0; m < M; m++)
        if ((int64_t)typeTopicCounts[m].size() != (int64_t)numTypes[m] * numTopics || (int)tokensPerTopic[m].size() != numTopics)
            throw std::invalid_argument("FastQMVWVTopicInferencer: count arrays do not match numTypes x numTopics");
    if (discrWeightPerModality.empty()) discrWeightPerModality.assign(M, 1.0);
    if (pMean.empty()) { pMean.assign(M, std::vector<double>(M, 0.0)); for (int m = 0; m < M; m++) pMean[m][m] = 1.0; }
    // initInferencer() INF:557-586 builds the trees; here they are built on the device when the corpus arrives
}

FastQMVWVTopicInferencer::~FastQMVWVTopicInferencer()
{
    if (h_) mvhdp_destroy(h_);
}

void FastQMVWVTopicInferencer::check(int rc, const char* what)
{
    if (rc != MVHDP_OK)
        throw std::runtime_error(std::string(what) + ": " + mvhdp_last_error(h_) + " (code " + std::to_string(rc) + ")");
}

std::string FastQMVWVTopicInferencer::inferTopicDistributionsOnNewDocs(const std::vector<InstanceList>& training)
{
    const int M = numModalities, K = numTopics;
    if ((int)training.size() != M) throw std::invalid_argument("inferTopicDistributionsOnNewDocs: one InstanceList per modality");
    std::unordered_map<std::string, int> entityPosition;             // INF:116
    data.clear();
    for (int m = 0; m < M; m++) {                                     // INF:118-165
        // numTypes[m] keeps the trained model's size (INF:121): new alphabet entries are out-of-vocabulary
        betaSum[m] = beta[m] * numTypes[m];                           // INF:127
        for (const Instance& instance : training[m].instances) {
            TopicAssignment t;
            t.present = true;
            t.tokens = instance.features;
            t.source = instance.source;
            t.topics.assign(instance.features.size(), 0);             // new int[tokens.size()] INF:136
            const std::string& entityId = instance.name;              // INF:141
            auto it = entityPosition.find(entityId);
            if (m != 0 && it != entityPosition.end()) {               // INF:145-149
                data[it->second].Assignments[m] = std::move(t);
            } else {                                                  // INF:151-157
                MixTopicModelTopicAssignment mt;
                mt.EntityId = entityId;
                mt.Assignments.assign(M, TopicAssignment());
                mt.Assignments[m] = std::move(t);
                data.push_back(std::move(mt));
                entityPosition[entityId] = (int)data.size() - 1;
            }
        }
    }

    if (h_) { mvhdp_destroy(h_); h_ = nullptr; }
    mvhdp_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.num_topics = K; cfg.num_modalities = M; cfg.device = device_;
    for (int m = 0; m < M; m++) cfg.num_types[m] = numTypes[m];
    if (mvhdp_create(&cfg, &h_) != MVHDP_OK) throw std::runtime_error(std::string("mvhdp_create: ") + mvhdp_last_error(nullptr));

    const int64_t D = (int64_t)data.size();
    for (int m = 0; m < M; m++) {
        std::vector<int64_t> off(D + 1, 0);
        for (int64_t d = 0; d < D; d++) off[d + 1] = off[d] + (int64_t)data[d].Assignments[m].tokens.size();
        std::vector<int32_t> tok((size_t)off[D]);
        for (int64_t d = 0; d < D; d++) std::copy(data[d].Assignments[m].tokens.begin(), data[d].Assignments[m].tokens.end(), tok.begin() + off[d]);
        check(mvhdp_set_corpus(h_, m, D, off.data(), tok.data()), "mvhdp_set_corpus");
        check(mvhdp_set_counts(h_, m, typeTopicCounts[m].data(), tokensPerTopic[m].data()), "mvhdp_set_counts");   // the trained model
    }
    for (int i = 0; i < M; i++) {                                     // INF:216-219
        std::fill(p_a[i].begin(), p_a[i].end(), 0.2);
        std::fill(p_b[i].begin(), p_b[i].end(), 1.0);
    }
    {
        std::vector<double> a((size_t)M * (K + 1));
        for (int m = 0; m < M; m++) std::copy(alpha[m].begin(), alpha[m].end(), a.begin() + (size_t)m * (K + 1));
        mvhdp_hyper hy;
        std::memset(&hy, 0, sizeof hy);
        hy.alpha = a.data();
        hy.inactive = nullptr;                                        // INF:228: a fresh, empty inActiveTopicIndex
        for (int m = 0; m < M; m++) {
            hy.alpha_sum[m] = alphaSum[m]; hy.beta[m] = beta[m]; hy.beta_sum[m] = betaSum[m]; hy.gamma[m] = gamma[m];
            for (int j = 0; j < M; j++) { hy.p_a[m][j] = p_a[m][j]; hy.p_b[m][j] = p_b[m][j]; }
        }
        check(mvhdp_set_hyper(h_, &hy), "mvhdp_set_hyper");
    }
    check(mvhdp_build_inference_trees(h_), "mvhdp_build_inference_trees");         // INF:557-586
    const uint64_t seed = (randomSeed == -1) ? (uint64_t)std::random_device{}() : (uint64_t)(int64_t)randomSeed;
    check(mvhdp_init_assignments_from_trees(h_, seed), "mvhdp_init_assignments_from_trees");   // INF:169-199

    iterationStats.clear();
    for (int iteration = 1; iteration <= numIterations; iteration++) {             // INF:258-288: nst = 1, nut = 0
        mvhdp_sweep_stats st;
        check(mvhdp_sweep(h_, (uint32_t)iteration, seed, MVHDP_SWEEP_FROZEN, nullptr, nullptr, &st), "mvhdp_sweep");
        iterationStats.push_back(st);
    }
    for (int m = 0; m < M; m++) {                                     // back into the arrays getFeatures() returns
        int64_t N = 0;
        for (int64_t d = 0; d < D; d++) N += (int64_t)data[d].Assignments[m].tokens.size();
        std::vector<int32_t> z((size_t)std::max<int64_t>(N, 1));
        check(mvhdp_get_assignments(h_, m, z.data()), "mvhdp_get_assignments");
        int64_t o = 0;
        for (int64_t d = 0; d < D; d++) {
            TopicAssignment& ta = data[d].Assignments[m];
            std::copy(z.begin() + o, z.begin() + o + (int64_t)ta.topics.size(), ta.topics.begin());
            o += (int64_t)ta.topics.size();
        }
    }
    return printDocumentTopicsToString(0.03, -1);                     // INF:326-329
}

std::vector<double> FastQMVWVTopicInferencer::docTopicProportions()
{
    if (!h_) throw std::runtime_error("docTopicProportions() before inferTopicDistributionsOnNewDocs()");
    const int M = numModalities, K = numTopics;
    std::vector<double> w((size_t)M);
    for (int m = 0; m < M; m++) w[m] = (m == 0 ? 1 : discrWeightPerModality[m]) * pMean[0][m];   // INF:407
    std::vector<double> out((size_t)data.size() * K);
    if (!data.empty()) check(mvhdp_doc_topic_proportions(h_, w.data(), 0, (int64_t)data.size(), out.data()), "mvhdp_doc_topic_proportions");
    return out;
}

std::string FastQMVWVTopicInferencer::printDocumentTopicsToString(double threshold, int max)
{
    if (!h_) throw std::runtime_error("printDocumentTopics() before inferTopicDistributionsOnNewDocs()");
    const int M = numModalities;
    std::vector<double> w((size_t)M);
    for (int m = 0; m < M; m++) w[m] = (m == 0 ? 1 : discrWeightPerModality[m]) * pMean[0][m];
    std::vector<std::string> names;
    names.reserve(data.size());
    for (auto& e : data) names.push_back(e.EntityId);
    return formatDocumentTopics(h_, names, numTopics, w, threshold, max);
}

}  // namespace mvtm

// ---------------------------------------------------------------------------
// extern "C" hooks (include/mvtm_host.h)
// ---------------------------------------------------------------------------
using mvtm::FastQMVWVParallelTopicModel;
using mvtm::FastQMVWVTopicInferencer;

extern "C" {
extern const char* mvtm_last_error(void);
void mvtm_set_last_error(const char* msg);

// getInferencer() PTM:3457-3463; discr_weight[M] / p_mean[M][M] replace the model's when non-NULL
void* mvtm_model_get_inferencer(void* p, const double* discr_weight, const double* p_mean)
{
    auto* model = (FastQMVWVParallelTopicModel*)p;
    try {
        const int M = model->numModalities;
        if (discr_weight) model->discrWeightPerModality.assign(discr_weight, discr_weight + M);
        if (p_mean) {
            model->pMean.assign(M, std::vector<double>(M));
            for (int i = 0; i < M; i++) for (int j = 0; j < M; j++) model->pMean[i][j] = p_mean[i * M + j];
        }
        return model->getInferencer().release();
    } catch (const std::exception& e) { mvtm_set_last_error(e.what()); return nullptr; }
}

void mvtm_inferencer_delete(void* p) { delete (FastQMVWVTopicInferencer*)p; }

int mvtm_inferencer_configure(void* p, int numIterations, int randomSeed, int device)
{
    auto* inf = (FastQMVWVTopicInferencer*)p;
    inf->setNumIterations(numIterations); inf->setRandomSeed(randomSeed); inf->setDevice(device);
    return 0;
}

// per view v: n_inst[v] instances with names name_ids[v][i], features tokens[v][off[v][i]..off[v][i+1]).
// The returned text (printDocumentTopics(out, 0.03, -1)) is copied to text_out when it fits; the return value is its length.
int64_t mvtm_inferencer_infer(void* p, int M, const int64_t* n_inst, const int64_t* const* name_ids,
                              const int64_t* const* off, const int32_t* const* tokens, char* text_out, int64_t cap)
{
    auto* inf = (FastQMVWVTopicInferencer*)p;
    try {
        std::vector<mvtm::InstanceList> training(M);
        for (int v = 0; v < M; v++) {
            training[v].instances.resize((size_t)n_inst[v]);
            for (int64_t i = 0; i < n_inst[v]; i++) {
                training[v].instances[i].name = std::to_string(name_ids[v][i]);
                training[v].instances[i].features.assign(tokens[v] + off[v][i], tokens[v] + off[v][i + 1]);
            }
        }
        const std::string text = inf->inferTopicDistributionsOnNewDocs(training);
        if (text_out && cap > 0) {
            const int64_t n = std::min<int64_t>((int64_t)text.size(), cap - 1);
            std::memcpy(text_out, text.data(), (size_t)n);
            text_out[n] = 0;
        }
        return (int64_t)text.size();
    } catch (const std::exception& e) { mvtm_set_last_error(e.what()); return -1; }
}

int64_t mvtm_inferencer_num_entities(void* p) { return (int64_t)((FastQMVWVTopicInferencer*)p)->data.size(); }

int64_t mvtm_inferencer_view_tokens(void* p, int m)
{
    int64_t n = 0;
    for (auto& e : ((FastQMVWVTopicInferencer*)p)->data) n += (int64_t)e.Assignments[m].tokens.size();
    return n;
}

int mvtm_inferencer_get_view(void* p, int m, int64_t* entity_ids, int64_t* off, int32_t* tokens, int32_t* topics)
{
    auto* inf = (FastQMVWVTopicInferencer*)p;
    int64_t o = 0, d = 0;
    if (off) off[0] = 0;
    for (auto& e : inf->data) {
        const auto& ta = e.Assignments[m];
        if (entity_ids) entity_ids[d] = std::stoll(e.EntityId);
        if (tokens) std::copy(ta.tokens.begin(), ta.tokens.end(), tokens + o);
        if (topics) std::copy(ta.topics.begin(), ta.topics.end(), topics + o);
        o += (int64_t)ta.tokens.size();
        d++;
        if (off) off[d] = o;
    }
    return 0;
}

int mvtm_inferencer_doc_topics(void* p, double* out)
{
    auto* inf = (FastQMVWVTopicInferencer*)p;
    try {
        const std::vector<double> v = inf->docTopicProportions();
        std::copy(v.begin(), v.end(), out);
        return 0;
    } catch (const std::exception& e) { mvtm_set_last_error(e.what()); return -1; }
}

int64_t mvtm_inferencer_print_document_topics(void* p, double threshold, int max, char* text_out, int64_t cap)
{
    auto* inf = (FastQMVWVTopicInferencer*)p;
    try {
        const std::string text = inf->printDocumentTopicsToString(threshold, max);
        if (text_out && cap > 0) {
            const int64_t n = std::min<int64_t>((int64_t)text.size(), cap - 1);
            std::memcpy(text_out, text.data(), (size_t)n);
            text_out[n] = 0;
        }
        return (int64_t)text.size();
    } catch (const std::exception& e) { mvtm_set_last_error(e.what()); return -1; }
}

int mvtm_inferencer_get_stats(void* p, int i, mvhdp_sweep_stats* st)
{
    auto* inf = (FastQMVWVTopicInferencer*)p;
    if (i < 0 || i >= (int)inf->iterationStats.size()) return -1;
    *st = inf->iterationStats[(size_t)i];
    return 0;
}
}
