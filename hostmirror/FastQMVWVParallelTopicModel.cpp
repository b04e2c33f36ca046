// FastQMVWVParallelTopicModel.cpp — see the header.  Host logic only; every
// per-token operation happens in libmvhdp's kernels.
#include "FastQMVWVParallelTopicModel.h"
#include "FastQMVWVTopicInferencer.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <random>
#include <stdexcept>
#include <charconv>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <unordered_map>

#include <zlib.h>

#include "java_random.h"

namespace mvtm {

FastQMVWVParallelTopicModel::FastQMVWVParallelTopicModel(int numberOfTopics, int8_t numModalities_, double alpha_, double beta_,
                                                         bool useCycleProposals, const std::string& SQLConnectionString,
                                                         bool useTypeVectors, double vectorsLambda, bool trainTypeVectors)
    : numTopics(numberOfTopics), numModalities(numModalities_)
{
    if (numberOfTopics < 1 || numberOfTopics > MVHDP_MAX_TOPICS) throw std::invalid_argument("numberOfTopics out of range");
    if (numModalities_ < 1 || numModalities_ > MVHDP_MAX_MODALITIES) throw std::invalid_argument("numModalities out of range");
    if (useTypeVectors || trainTypeVectors)
        throw std::invalid_argument("the embedding mix (useTypeVectors/trainTypeVectors) is outside the accelerated path");
    (void)useCycleProposals; (void)SQLConnectionString; (void)vectorsLambda;
    const int M = numModalities, K = numTopics;
    alphaSum.assign(M, 0); beta.assign(M, 0); betaSum.assign(M, 0); gamma.assign(M, 0);
    alpha.assign(M, std::vector<double>(K + 1, 0.0));
    totalTokens.assign(M, 0); totalDocsPerModality.assign(M, 0); numTypes.assign(M, 0);
    tokensPerTopic.assign(M, std::vector<int32_t>(K, 0));
    for (int m = 0; m < M; m++) {                       // PTM:207-214
        alphaSum[m] = K * alpha_;
        std::fill(alpha[m].begin(), alpha[m].end(), alpha_);
        beta[m] = beta_;
        gamma[m] = 1;
    }
    tablesCnt.assign(M, 0.0);                          // PTM:238-239
    gammaView.assign(M, 0.0);
    p_a.assign(M, std::vector<double>(M, 0.0));        // PTM:228-229
    p_b.assign(M, std::vector<double>(M, 0.0));
}

FastQMVWVParallelTopicModel::~FastQMVWVParallelTopicModel()
{
}

void FastQMVWVParallelTopicModel::check(int rc, const char* what)
{
    if (rc != MVHDP_OK)
        throw std::runtime_error(std::string(what) + ": " + dev_.lastError() + " (code " + std::to_string(rc) + ")");
}

void FastQMVWVParallelTopicModel::addInstances(const std::vector<InstanceList>& training, const std::string& batchId, int vectorSize)
{
    (void)batchId; (void)vectorSize;
    const int M = numModalities, K = numTopics;
    if ((int)training.size() != M) throw std::invalid_argument("addInstances: one InstanceList per modality");
    std::unordered_map<std::string, int> entityPosition;             // PTM:398
    typeTotals.assign(M, {});
    data.clear();

    for (int m = 0; m < M; m++) {                                     // PTM:410-463
        numTypes[m] = training[m].alphabetSize;                       // PTM:413
        if (alphabet.size() != (size_t)M) alphabet.assign(M, {});
        alphabet[m] = training[m].alphabet;                           // PTM:412
        if (numTypes[m] < 1) throw std::invalid_argument("addInstances: empty alphabet");
        typeTotals[m].assign(numTypes[m], 0);
        betaSum[m] = beta[m] * numTypes[m];                           // PTM:420
        for (const Instance& instance : training[m].instances) {
            TopicAssignment t;
            t.present = true;
            t.tokens = instance.features;
            t.source = instance.source;
            t.topics.assign(instance.features.size(), 0);             // new int[tokens.size()] PTM:430
            const std::string& entityId = instance.name;              // PTM:437
            auto it = entityPosition.find(entityId);
            if (m != 0 && it != entityPosition.end()) {               // PTM:443-447
                data[it->second].Assignments[m] = std::move(t);
            } else {                                                  // PTM:449-455
                MixTopicModelTopicAssignment mt;
                mt.EntityId = entityId;
                mt.Assignments.assign(M, TopicAssignment());
                mt.Assignments[m] = std::move(t);
                data.push_back(std::move(mt));
                entityPosition[entityId] = (int)data.size() - 1;
            }
        }
    }

    // PTM:403-408, 465-515: random initial assignments from MALLET Randoms(randomSeed)
    int64_t seed = randomSeed;
    if (randomSeed == -1) seed = (int64_t)std::random_device{}();
    Randoms random(seed);
    std::vector<int32_t> activeTopics;
    for (MixTopicModelTopicAssignment& entity : data) {
        for (int m = 0; m < M; m++) {
            if (m == 0) activeTopics.clear();                         // PTM:470-472
            TopicAssignment& document = entity.Assignments[m];
            if (!document.present) continue;
            for (size_t position = 0; position < document.tokens.size(); position++) {
                int type = document.tokens[position];
                int topic;
                if (m == 0) {                                         // PTM:499-501
                    topic = random.nextInt(K);
                    activeTopics.push_back(topic);
                } else if (!activeTopics.empty()) {                   // PTM:502-504
                    int ind = random.nextInt((int32_t)activeTopics.size());
                    topic = activeTopics[ind];
                } else {                                              // PTM:505-507
                    topic = random.nextInt(K);
                }
                document.topics[position] = topic;
                if (type >= 0 && type < numTypes[m]) typeTotals[m][type]++;   // PTM:511
            }
        }
    }

    initializeHistograms();                                           // PTM:527
    // initSpace PTM:575-598: the model arrays live on the device
    maxTypeCount.assign(M, 0);
    for (int m = 0; m < M; m++)
        for (int c : typeTotals[m]) maxTypeCount[m] = std::max(maxTypeCount[m], c);

    mvhdp_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.num_topics = K; cfg.num_modalities = M; cfg.device = device_; cfg.doc_id_base = docIdBase_;
    for (int m = 0; m < M; m++) cfg.num_types[m] = numTypes[m];
    // one handle, or numShards_ document shards behind an mvhdp_group (device_model.h): contiguous entity ranges by token count
    const int64_t D = (int64_t)data.size();
    {
        std::vector<int64_t> entityTokens((size_t)D, 0);
        for (int64_t d = 0; d < D; d++)
            for (int m = 0; m < M; m++) entityTokens[(size_t)d] += (int64_t)data[d].Assignments[m].tokens.size();
        int rc = dev_.create(cfg, numShards_, entityTokens);
        if (rc != MVHDP_OK) throw std::runtime_error(dev_.lastError());
        h_ = dev_.first();
    }

    // flatten MixTopicModelTopicAssignment -> CSR per view
    for (int m = 0; m < M; m++) {
        std::vector<int64_t> off(D + 1, 0);
        for (int64_t d = 0; d < D; d++) off[d + 1] = off[d] + (int64_t)data[d].Assignments[m].tokens.size();
        std::vector<int32_t> tok((size_t)off[D]), z((size_t)off[D]);
        for (int64_t d = 0; d < D; d++) {
            const TopicAssignment& ta = data[d].Assignments[m];
            std::copy(ta.tokens.begin(), ta.tokens.end(), tok.begin() + off[d]);
            std::copy(ta.topics.begin(), ta.topics.end(), z.begin() + off[d]);
        }
        check(dev_.setCorpus(m, D, off.data(), tok.data()), "mvhdp_set_corpus");
        check(dev_.setAssignments(m, z.data()), "mvhdp_set_assignments");
    }
    pushHyper();
    check(dev_.buildCounts(), "mvhdp_build_counts");              // PTM:529
    check(dev_.buildTrees(), "mvhdp_build_trees");                // PTM:531
    syncFromDevice(true);
}

void FastQMVWVParallelTopicModel::initializeHistograms()
{
    const int M = numModalities;                                      // PTM:849-897
    histogramSize.assign(M, 0);
    std::fill(totalTokens.begin(), totalTokens.end(), 0);
    std::fill(totalDocsPerModality.begin(), totalDocsPerModality.end(), 0);
    for (const auto& entity : data)
        for (int i = 0; i < M; i++) {
            const TopicAssignment& document = entity.Assignments[i];
            if (document.present) {
                int seqLen = (int)document.tokens.size();
                histogramSize[i] = std::max(histogramSize[i], seqLen);
                totalTokens[i] += seqLen;
                totalDocsPerModality[i]++;                            // PTM:624
            }
        }
    docLengthCounts.assign(M, {});
    topicDocCounts.assign(M, {});
    for (int m = 0; m < M; m++) docLengthCounts[m].assign(histogramSize[m] + 1, 0);
}

void FastQMVWVParallelTopicModel::pushHyper()
{
    const int M = numModalities, K = numTopics;
    std::vector<double> a((size_t)M * (K + 1));
    std::vector<uint8_t> ina((size_t)K, 0);
    for (int m = 0; m < M; m++) std::copy(alpha[m].begin(), alpha[m].end(), a.begin() + (size_t)m * (K + 1));
    for (int t : inActiveTopicIndex) if (t >= 0 && t < K) ina[t] = 1;
    mvhdp_hyper hy;
    std::memset(&hy, 0, sizeof hy);
    hy.alpha = a.data();
    hy.inactive = ina.data();
    for (int m = 0; m < M; m++) {
        hy.alpha_sum[m] = alphaSum[m]; hy.beta[m] = beta[m]; hy.beta_sum[m] = betaSum[m]; hy.gamma[m] = gamma[m];
        for (int j = 0; j < M; j++) { hy.p_a[m][j] = p_a[m][j]; hy.p_b[m][j] = p_b[m][j]; }
    }
    check(dev_.setHyper(&hy), "mvhdp_set_hyper");
}

void FastQMVWVParallelTopicModel::syncFromDevice(bool histograms)
{
    const int M = numModalities, K = numTopics;
    const int64_t D = (int64_t)data.size();
    typeTopicCounts.resize(M);
    for (int m = 0; m < M; m++) {
        typeTopicCounts[m].resize((size_t)numTypes[m] * K);
        check(dev_.getCounts(m, typeTopicCounts[m].data(), tokensPerTopic[m].data()), "mvhdp_get_counts");
        int64_t N = 0;
        for (int64_t d = 0; d < D; d++) N += (int64_t)data[d].Assignments[m].tokens.size();
        std::vector<int32_t> z((size_t)std::max<int64_t>(N, 1));
        check(dev_.getAssignments(m, z.data()), "mvhdp_get_assignments");
        int64_t o = 0;
        for (int64_t d = 0; d < D; d++) {                             // back into the very arrays getFeatures() returns
            TopicAssignment& ta = data[d].Assignments[m];
            std::copy(z.begin() + o, z.begin() + o + (int64_t)ta.topics.size(), ta.topics.begin());
            o += (int64_t)ta.topics.size();
        }
        if (histograms) {
            topicDocCounts[m].assign((size_t)K * (histogramSize[m] + 1), 0);
            check(dev_.docTopicHist(m, topicDocCounts[m].data(), histogramSize[m] + 1,
                                    docLengthCounts[m].data(), histogramSize[m] + 1), "mvhdp_get_doc_topic_hist");
        }
    }
    // a topic activation (UPD:263-270) may have changed alpha / inActiveTopicIndex
    std::vector<double> a((size_t)M * (K + 1));
    std::vector<uint8_t> ina((size_t)K);
    check(dev_.getAlpha(a.data(), ina.data()), "mvhdp_get_alpha");
    for (int m = 0; m < M; m++) std::copy(a.begin() + (size_t)m * (K + 1), a.begin() + (size_t)(m + 1) * (K + 1), alpha[m].begin());
    inActiveTopicIndex.clear();
    for (int k = 0; k < K; k++) if (ina[k]) inActiveTopicIndex.insert(k);
}

// ---- MALLET 2.0.8 Dirichlet.digamma as compiled: the series coefficients are integer quotients (= 0) ----
double FastQMVWVParallelTopicModel::digamma(double z)
{
    double psi = 0;
    if (z < 1e-06) { psi = -0.5772156649015329 - 1 / z; return psi; }
    while (z < 9.5) { psi = psi - 1 / z; z = z + 1; }
    double invZ = 1 / z;
    double invZSquared = invZ * invZ;
    psi = psi + (std::log(z) - 0.5 * invZ
          - invZSquared * (0.0 - invZSquared * (0.0 - invZSquared * (0.0 - invZSquared * (0.0 - invZSquared * (0.0 - invZSquared * (0.0 - invZSquared * 0.0)))))));
    return psi;
}

// ---- MALLET 2.0.8 Dirichlet.learnSymmetricConcentration (quirks kept: previousLength never advances) ----
double FastQMVWVParallelTopicModel::learnSymmetricConcentration(const std::vector<int32_t>& countHistogram,
                                                                const std::vector<int32_t>& observationLengths,
                                                                int numDimensions, double currentValue)
{
    double currentDigamma;
    int largestNonZeroCount = 0;
    std::vector<int> nonZeroLengthIndex(observationLengths.size());
    for (size_t index = 0; index < countHistogram.size(); index++) if (countHistogram[index] > 0) largestNonZeroCount = (int)index;
    int denseIndex = 0;
    for (size_t index = 0; index < observationLengths.size(); index++)
        if (observationLengths[index] > 0) { nonZeroLengthIndex[denseIndex] = (int)index; denseIndex++; }
    int denseIndexSize = denseIndex;
    for (int iteration = 1; iteration <= 200; iteration++) {
        double currentParameter = currentValue / numDimensions;
        currentDigamma = 0;
        double numerator = 0;
        for (int index = 1; index <= largestNonZeroCount; index++) {
            currentDigamma += 1.0 / (currentParameter + index - 1);
            numerator += countHistogram[index] * currentDigamma;
        }
        currentDigamma = 0;
        double denominator = 0;
        int previousLength = 0;
        double cachedDigamma = digamma(currentValue);
        for (denseIndex = 0; denseIndex < denseIndexSize; denseIndex++) {
            int length = nonZeroLengthIndex[denseIndex];
            if (length - previousLength > 20) currentDigamma = digamma(currentValue + length) - cachedDigamma;
            else for (int index = previousLength; index < length; index++) currentDigamma += 1.0 / (currentValue + index);
            denominator += currentDigamma * observationLengths[length];
        }
        currentValue = currentParameter * numerator / denominator;
    }
    return currentValue;
}

void FastQMVWVParallelTopicModel::optimizeP(bool appendMetadata)
{
    (void)appendMetadata;
    const int M = numModalities;
    std::vector<double> sums((size_t)M * M, 0.0);
    check(dev_.viewOverlapSums(sums.data()), "mvhdp_view_overlap_sums");      // PTM:2706-2782 on the device
    pMean.assign(M, std::vector<double>(M, 0.0));
    for (int m = 0; m < M; m++) {                                                    // PTM:2784-2812
        pMean[m][m] = 1;
        for (int i = m + 1; i < M; i++) {
            double sum = sums[(size_t)m * M + i];
            pMean[m][i] = sum / (std::min(totalDocsPerModality[m], totalDocsPerModality[i]));
            pMean[i][m] = pMean[m][i];
            double a = pMean[m][i] == 1 ? 5000 : -1.0 / std::log(pMean[m][i]);
            double b = 1;
            p_a[m][i] = std::min(a, 100.0);
            p_a[i][m] = std::min(a, 100.0);
            p_b[m][i] = b;
            p_b[i][m] = b;
        }
    }
}

void FastQMVWVParallelTopicModel::optimizeBeta()
{
    const int M = numModalities, K = numTopics;
    for (int m = 0; m < M; m++) {                                                    // PTM:2293
        double prevBetaSum = betaSum[m];
        std::vector<int32_t> countHistogram((size_t)maxTypeCount[m] + 1, 0);
        check(dev_.countHistogram(m, countHistogram.data(), maxTypeCount[m] + 1), "mvhdp_get_count_histogram");  // PTM:2299-2309
        std::vector<int32_t> nk((size_t)K);
        check(dev_.getCounts(m, nullptr, nk.data()), "mvhdp_get_counts");
        int maxTopicSize = 0;
        for (int topic = 0; topic < K; topic++) maxTopicSize = std::max(maxTopicSize, nk[topic]);
        std::vector<int32_t> topicSizeHistogram((size_t)maxTopicSize + 1, 0);
        for (int topic = 0; topic < K; topic++) topicSizeHistogram[nk[topic]]++;
        betaSum[m] = learnSymmetricConcentration(countHistogram, topicSizeHistogram, numTypes[m], betaSum[m]);   // PTM:2327
        if (betaSum[m] < numTypes[m] * 0.0001) {                                     // PTM:2332-2335
            beta[m] = 0.0001;
            betaSum[m] = beta[m] * numTypes[m];
        } else if (std::isnan(betaSum[m])) {                                         // PTM:2337-2349
            if (beta[m] == 0.01) { beta[m] = 0.0001; betaSum[m] = beta[m] * numTypes[m]; }
            else { betaSum[m] = prevBetaSum; beta[m] = betaSum[m] / numTypes[m]; }
        } else {
            beta[m] = betaSum[m] / numTypes[m];                                      // PTM:2351
        }
    }
}

void FastQMVWVParallelTopicModel::seedHostSamplers(int64_t sampSeed, int64_t randomSeed64)
{
    sampRand_.setSeed(sampSeed);
    random_ = Randoms(randomSeed64);
    hostSamplersSeeded_ = true;
}

void FastQMVWVParallelTopicModel::ensureHostSamplers()
{
    if (hostSamplersSeeded_) return;
    if (randomSeed == -1) { std::random_device rd; seedHostSamplers((int64_t)rd(), (int64_t)rd()); }
    else seedHostSamplers((int64_t)randomSeed + 1, (int64_t)randomSeed);
}

std::vector<double> FastQMVWVParallelTopicModel::sampleDirichlet(const std::vector<double>& p)
{
    double magnitude = 0;                                                            // PTM:2593-2634
    std::vector<double> partition(p.size());
    for (size_t i = 0; i < p.size(); i++) magnitude += p[i];
    for (size_t i = 0; i < p.size(); i++) partition[i] = p[i] / magnitude;
    std::vector<double> distribution(partition.size());
    double sum = 0;
    for (size_t i = 0; i < distribution.size(); i++) {
        if (partition[i] * magnitude > 0) {
            distribution[i] = random_.nextGamma(partition[i] * magnitude, 1);
            if (distribution[i] <= 0) distribution[i] = 0.0001;
        } else {
            distribution[i] = 0.0001;
        }
        sum += distribution[i];
    }
    for (size_t i = 0; i < distribution.size(); i++) distribution[i] /= sum;
    return distribution;
}

void FastQMVWVParallelTopicModel::optimizeDP()
{
    const int M = numModalities, K = numTopics;
    ensureHostSamplers();
    std::vector<std::vector<double>> mk(M, std::vector<double>((size_t)K + 1, 0.0));
    std::vector<double> mk_root((size_t)K + 1, 0.0);
    std::fill(tablesCnt.begin(), tablesCnt.end(), 0.0);
    for (int t = 0; t < K; t++) inActiveTopicIndex.insert(t);                        // PTM:2449-2451

    // view tables simulation PTM:2454-2488; topicDocCounts[m][t][i] = entities with i tokens of topic t in view m
    for (int m = 0; m < M; m++) {
        const int len = histogramSize[m] + 1;
        topicDocCounts[m].assign((size_t)K * len, 0);
        check(dev_.docTopicHist(m, topicDocCounts[m].data(), len, docLengthCounts[m].data(), len), "mvhdp_get_doc_topic_hist");
        if (deviceTableStatistics_) {                                                    // (opt-in: the same statistic from the library, see include/mvhdp.h)
            std::vector<double> conc((size_t)K); std::vector<uint8_t> act((size_t)K);
            for (int t = 0; t < K; t++) conc[t] = gamma[m] * alpha[m][t];
            check(dev_.dpTableStatistics(m, topicDocCounts[m].data(), len, conc.data(), (randomSeed == -1) ? 0x9E3779B97F4A7C15ull : (uint64_t)(int64_t)randomSeed, (uint32_t)(gammaCalls_ * 16 + m), mk[m].data(), act.data()), "mvhdp_dp_table_statistics");
            for (int t = 0; t < K; t++) if (act[t]) inActiveTopicIndex.erase(t);
            continue;
        }
        for (int t = 0; t < K; t++) {
            const int32_t* tdc = topicDocCounts[m].data() + (size_t)t * len;
            for (int i = 0; i < len; i++) {
                if (tdc[i] > 0 && i > 1) {
                    inActiveTopicIndex.erase(t);
                    int curTbls = 0;
                    try { curTbls = Samplers.randAntoniak(gamma[m] * alpha[m][t], i); }
                    catch (const std::exception&) { curTbls = 1; }
                    mk[m][t] += (tdc[i] * curTbls);
                } else if (tdc[i] > 0 && i == 1) {
                    inActiveTopicIndex.erase(t);
                    mk[m][t] += tdc[i];
                }
            }
        }
    }
    // root tables simulation PTM:2491-2517
    if (deviceTableStatistics_) {                                                        // (opt-in: the K * M draws from the library)
        std::vector<int32_t> items((size_t)K * M), tabs((size_t)K * M);
        std::vector<double> conc((size_t)K * M, gammaRoot);
        for (int t = 0; t < K; t++)
            for (int m = 0; m < M; m++) {
                const double c = std::ceil(mk[m][t]);
                items[(size_t)t * M + m] = mk[m][t] > 1 ? (c >= 2147483647.0 ? 2147483647 : (int)c) : (mk[m][t] == 1 ? 1 : 0);
            }
        check(dev_.antoniakDraws(K * M, items.data(), conc.data(), (randomSeed == -1) ? 0x9E3779B97F4A7C15ull : (uint64_t)(int64_t)randomSeed, (uint32_t)(gammaCalls_ * 16 + 15), tabs.data()), "mvhdp_antoniak_draws");
        for (int t = 0; t < K; t++) for (int m = 0; m < M; m++) mk_root[t] += tabs[(size_t)t * M + m];
    } else
    for (int t = 0; t < K; t++)
        for (int m = 0; m < M; m++) {
            if (mk[m][t] > 1) {
                int curTbls = 0;
                try {
                    const double c = std::ceil(mk[m][t]);
                    // Java (int) of a double saturates
                    const int n = c >= 2147483647.0 ? 2147483647 : (int)c;
                    curTbls = Samplers.randAntoniak(gammaRoot, n);
                } catch (const std::exception&) { curTbls = 1; }
                mk_root[t] += curTbls;
            } else if (mk[m][t] == 1) {
                mk_root[t] += 1;
            }
        }

    std::vector<double> v((size_t)K + 1, 0.0);
    mk_root[K] = gammaRoot;
    rootTablesCnt = 0;
    for (double x : mk_root) rootTablesCnt += x;                                     // Vectors.sum
    int numSamples = 10;
    for (int i = 0; i < numSamples; i++) {                                           // PTM:2525-2535
        std::vector<double> tt = sampleDirichlet(mk_root);
        for (int kk = 0; kk <= K; kk++) v[kk] += tt[kk] / (double)numSamples;
    }
    for (int m = 0; m < M; m++) {                                                    // PTM:2549-2580
        for (int t = 0; t < K; t++) mk[m][t] += v[t] * gammaRoot;
        std::fill(alpha[m].begin(), alpha[m].end(), 0.0);
        alphaSum[m] = 0;
        mk[m][K] = gammaView[m] + v[K] * gammaRoot;
        tablesCnt[m] = 0;
        for (double x : mk[m]) tablesCnt[m] += x;
        for (int i = 0; i < numSamples; i++) {
            std::vector<double> tt = sampleDirichlet(mk[m]);
            for (int kk = 0; kk <= K; kk++) {
                double sampleAlpha = tt[kk] / (double)numSamples;
                alpha[m][kk] += sampleAlpha;
                alphaSum[m] += sampleAlpha;
            }
        }
    }
}

void FastQMVWVParallelTopicModel::optimizeGamma()
{
    const int M = numModalities, K = numTopics;
    ensureHostSamplers();
    gammaCalls_++;
    RandomSamplers<JavaRandom> samp(&sampRand_);
    const double aalpha = 5, balpha = 0.1, agamma = 5, bgamma = 0.1;                 // PTM:2373-2379
    const int R = 10;
    for (int r = 0; r < R; r++) {                                                    // root level PTM:2384-2395
        double eta = samp.randBeta(gammaRoot + 1, rootTablesCnt);
        double bloge = bgamma - std::log(eta);
        double pie = 1. / (1. + (rootTablesCnt * bloge / (agamma + K - 1)));
        int u = samp.randBernoulli(pie);
        gammaRoot = samp.randGamma(agamma + K - 1 + u, 1. / bloge);
    }
    for (int m = 0; m < M; m++)                                                      // per view PTM:2398-2437
        for (int r = 0; r < R; r++) {
            double prevGamma = gamma[m];
            double eta = samp.randBeta(gammaView[m] + 1, tablesCnt[m]);
            double bloge = bgamma - std::log(eta);
            double pie = 1. / (1. + (tablesCnt[m] * bloge / (agamma + K - 1)));
            int u = samp.randBernoulli(pie);
            gammaView[m] = samp.randGamma(agamma + K - 1 + u, 1. / bloge);
            double qs = 0, qw = 0;                                                   // document level (Teh+06)
            if (deviceGammaStatistics_) {
                // the same two sums, every entity drawing from its own counter-based stream on the device (the
                // reference's stream for them, `samp` over ThreadLocalRandom, cannot be seeded or replayed anyway)
                const uint64_t seed = (randomSeed == -1) ? 0x9E3779B97F4A7C15ull : (uint64_t)(int64_t)randomSeed;
                check(dev_.gammaDocStatistics(m, gamma[m], seed, (uint32_t)(gammaCalls_ * 16 + r), &qs, &qw), "mvhdp_gamma_doc_statistics");
            } else
            for (size_t j = 0; j < docLengthCounts[m].size(); j++)
                for (int i = 0; i < docLengthCounts[m][j]; i++) {
                    qs += samp.randBernoulli((double)j / ((double)j + gamma[m]));
                    qw += std::log(samp.randBeta(gamma[m] + 1, (double)j));
                }
            gamma[m] = samp.randGamma(aalpha + tablesCnt[m] - qs, 1. / (balpha - qw));
            if (gamma[m] == 0) gamma[m] = prevGamma;
        }
}

// Java Double.toString layout: decimal for 1e-3 <= |v| < 1e7, otherwise d.dddE[-]n; always at least one
// digit after the point.  Digits: shortest round-trip (JDK >= 19; older JDKs print a few values longer).
std::string FastQMVWVParallelTopicModel::javaDoubleToString(double v)
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v > 0 ? "Infinity" : "-Infinity";
    if (v == 0) return std::signbit(v) ? "-0.0" : "0.0";
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, std::fabs(v), std::chars_format::scientific);
    std::string sci(buf, r.ptr);                       // d[.ddd]e[+-]xx
    size_t epos = sci.find('e');
    std::string mant = sci.substr(0, epos);
    int exp10 = std::stoi(sci.substr(epos + 1));
    std::string digits;
    for (char c : mant) if (c != '.') digits.push_back(c);
    std::string out = v < 0 ? "-" : "";
    if (exp10 >= -3 && exp10 < 7) {
        if (exp10 >= 0) {
            std::string ip = digits.substr(0, std::min<size_t>(digits.size(), (size_t)exp10 + 1));
            while ((int)ip.size() < exp10 + 1) ip.push_back('0');
            std::string fp = digits.size() > (size_t)exp10 + 1 ? digits.substr((size_t)exp10 + 1) : "0";
            out += ip + "." + fp;
        } else {
            out += "0." + std::string((size_t)(-exp10 - 1), '0') + digits;
        }
    } else {
        out += digits.substr(0, 1) + "." + (digits.size() > 1 ? digits.substr(1) : "0") + "E" + std::to_string(exp10);
    }
    return out;
}

std::string FastQMVWVParallelTopicModel::printStateToString()
{
    syncFromDevice(false);
    const int M = numModalities;
    std::ostringstream out;
    out << "#doc source pos typeindex type topic\n";                                  // PTM:3278
    out << "#alpha : ";
    for (int m = 0; m < M; m++) {
        out << "modality:" << m << "\n";
        for (int topic = 0; topic < numTopics; topic++) out << javaDoubleToString(gamma[m] * alpha[m][topic]) << " ";   // PTM:3283
    }
    out << "\n";
    out << "#beta[0] : " << javaDoubleToString(beta[0]) << "\n";
    for (size_t doc = 0; doc < data.size(); doc++) {
        for (int m = 0; m < M; m++) {
            const TopicAssignment& ta = data[doc].Assignments[m];
            if (!ta.present)                                                           // PTM:3291 dereferences null here
                throw std::runtime_error("printState: entity " + data[doc].EntityId + " has no view " + std::to_string(m) +
                                         " (the reference throws NullPointerException, PTM:3291)");
            const std::string source = ta.source.empty() ? "NA" : ta.source;
            for (size_t pi = 0; pi < ta.tokens.size(); pi++) {
                int type = ta.tokens[pi];
                const std::string word = (m < (int)alphabet.size() && type >= 0 && type < (int)alphabet[m].size())
                                             ? alphabet[m][type] : std::to_string(type);
                out << doc << " " << source << " " << pi << " " << type << " " << word << " " << ta.topics[pi] << "\n";   // PTM:3303
            }
        }
    }
    return out.str();
}

// java.text.NumberFormat.getInstance() (en-US DecimalFormat "#,##0.###") with setMaximumFractionDigits(5): HALF_EVEN on
// the exact binary value, trailing zeros dropped, grouping commas in the integer part, "-0" for a negative that rounds to 0
std::string FastQMVWVParallelTopicModel::numberFormat5(double v)
{
    if (std::isnan(v)) return "\xEF\xBF\xBD";                      // DecimalFormatSymbols NaN = U+FFFD
    if (std::isinf(v)) return v < 0 ? "-\xE2\x88\x9E" : "\xE2\x88\x9E";
    char buf[400];
    snprintf(buf, sizeof buf, "%.5f", std::fabs(v));               // glibc rounds the exact value, ties to even
    std::string t(buf);
    std::string ip = t.substr(0, t.find('.')), fp = t.substr(t.find('.') + 1);
    while (!fp.empty() && fp.back() == '0') fp.pop_back();
    std::string g;
    for (size_t i = 0; i < ip.size(); i++) {
        if (i && (ip.size() - i) % 3 == 0) g += ',';
        g += ip[i];
    }
    std::string out = (std::signbit(v) ? "-" : "") + g;
    if (!fp.empty()) out += "." + fp;
    return out;
}

std::vector<std::vector<std::pair<int32_t, int32_t>>> FastQMVWVParallelTopicModel::getSortedWords(int modality)
{
    const int K = numTopics, Vm = numTypes[modality];
    std::vector<std::vector<std::pair<int32_t, int32_t>>> topicSortedWords((size_t)K);
    const std::vector<int32_t>& ttc = typeTopicCounts[modality];
    for (int type = 0; type < Vm; type++)
        for (int topic = 0; topic < K; topic++) {
            const int cnt = ttc[(size_t)type * K + topic];
            if (cnt > 0) topicSortedWords[topic].emplace_back(type, cnt);              // PTM:1803-1806
        }
    for (auto& v : topicSortedWords)                                                    // TreeSet<IDSorter>: IDSorter.compareTo
        std::sort(v.begin(), v.end(), [](const std::pair<int32_t, int32_t>& a, const std::pair<int32_t, int32_t>& b) {
            return a.second > b.second || (a.second == b.second && a.first > b.first);
        });
    return topicSortedWords;
}

std::string FastQMVWVParallelTopicModel::displayTopWords(int numWords, int numLabels, bool usingNewLines)
{
    (void)numLabels;
    const int M = numModalities, K = numTopics;
    typeTopicCounts.resize(M);
    for (int m = 0; m < M; m++) {                                                      // the counts only (not z)
        typeTopicCounts[m].resize((size_t)numTypes[m] * K);
        check(dev_.getCounts(m, typeTopicCounts[m].data(), tokensPerTopic[m].data()), "mvhdp_get_counts");
    }
    std::vector<std::vector<std::vector<std::pair<int32_t, int32_t>>>> topicSortedWords;
    for (int m = 0; m < M; m++) topicSortedWords.push_back(getSortedWords(m));
    auto word_of = [&](int m, int type) {
        return (m < (int)alphabet.size() && type < (int)alphabet[m].size()) ? alphabet[m][type] : std::to_string(type);
    };
    std::string out;
    for (int topic = 0; topic < K; topic++) {
        for (int m = 0; m < M; m++) {
            const auto& sortedWords = topicSortedWords[m][topic];
            int word = 1;
            size_t it = 0;
            if (usingNewLines) {
                out += std::to_string(topic) + "\t" + numberFormat5(alpha[m][topic]) + "\n";
                while (it < sortedWords.size() && word < numWords) {                   // PTM:1869: numWords - 1 of them
                    out += word_of(m, sortedWords[it].first) + "\t" + numberFormat5((double)sortedWords[it].second) + "\n";
                    it++; word++;
                }
            } else {
                out += std::to_string(topic) + "\t" + numberFormat5(alpha[m][topic]) + "\t";
                while (it < sortedWords.size() && word < numWords) {
                    out += word_of(m, sortedWords[it].first) + "; ";
                    it++; word++;
                }
            }
        }
        out += "\n";
    }
    return out;
}

std::string FastQMVWVParallelTopicModel::printDocumentTopicsToString(double threshold, int max)
{
    const int M = numModalities, K = numTopics;
    if (!h_) throw std::runtime_error("printDocumentTopics() before addInstances()");
    if (dev_.numShards() > 1) throw std::runtime_error("printDocumentTopics: the carry-over of PTM:2873-2886 runs over ONE handle's entities (setNumShards(1))");
    if (pMean.empty()) throw std::runtime_error("printDocumentTopics: pMean is not set (optimizeP has not run; PTM:134)");
    if (discrWeightPerModality.empty()) discrWeightPerModality.assign(M, 1.0);
    pushHyper();
    std::vector<double> w((size_t)M);
    for (int m = 0; m < M; m++) w[m] = (m == 0 ? 1 : discrWeightPerModality[m]) * pMean[0][m];   // PTM:2895
    std::vector<std::string> names;
    names.reserve(data.size());
    for (auto& e : data) names.push_back(e.EntityId);
    return formatDocumentTopics(h_, names, K, w, threshold, max);
}

std::unique_ptr<FastQMVWVTopicInferencer> FastQMVWVParallelTopicModel::getInferencer()
{
    if (!h_) throw std::runtime_error("getInferencer() before addInstances()");
    syncFromDevice(false);                                       // typeTopicCounts / tokensPerTopic as trained so far
    return std::unique_ptr<FastQMVWVTopicInferencer>(new FastQMVWVTopicInferencer(
        numTypes, alpha, alphaSum, typeTopicCounts, tokensPerTopic, beta, betaSum, gamma, numTopics, numModalities,
        p_a, p_b, discrWeightPerModality, pMean));
}

void FastQMVWVParallelTopicModel::printDocumentTopics(const std::string& filename, double threshold, int max)
{
    const std::string text = printDocumentTopicsToString(threshold, max);
    std::ofstream f(filename, std::ios::binary);
    if (!f) throw std::runtime_error("printDocumentTopics: cannot open " + filename);
    f << text;
}

void FastQMVWVParallelTopicModel::printState(const std::string& filename)
{
    const std::string text = printStateToString();
    if (filename.size() > 3 && filename.compare(filename.size() - 3, 3, ".gz") == 0) {   // PTM:3269-3274 GZIPOutputStream
        gzFile f = gzopen(filename.c_str(), "wb");
        if (!f) throw std::runtime_error("printState: cannot open " + filename);
        gzwrite(f, text.data(), (unsigned)text.size());
        gzclose(f);
    } else {
        std::ofstream f(filename, std::ios::binary);
        if (!f) throw std::runtime_error("printState: cannot open " + filename);
        f << text;
    }
}

std::vector<double> FastQMVWVParallelTopicModel::modelLogLikelihood()
{
    std::vector<double> ll((size_t)numModalities, 0.0);
    check(dev_.logLikelihood(ll.data()), "mvhdp_model_log_likelihood");
    return ll;
}

void FastQMVWVParallelTopicModel::estimate()
{
    if (!h_) throw std::runtime_error("estimate() before addInstances()");
    const int M = numModalities;
    // PTM:1036-1037: nst = 3T/4 sampler threads and nut = T/4 updater threads become one kernel launch.
    for (int i = 0; i < M; i++) {                                     // PTM:1055-1058
        std::fill(p_a[i].begin(), p_a[i].end(), 0.2);
        std::fill(p_b[i].begin(), p_b[i].end(), 1.0);
    }
    uint64_t seed = (randomSeed == -1) ? (uint64_t)std::random_device{}() : (uint64_t)(int64_t)randomSeed;
    iterationLog.clear();
    for (int iteration = 1; iteration <= numIterations; iteration++) {   // PTM:1146
        auto t0 = std::chrono::steady_clock::now();
        if (showTopicsInterval != 0 && iteration != 0 && iteration % showTopicsInterval == 0)   // PTM:1150-1152
            topWordsLog.emplace_back(iteration, "\n" + displayTopWords(wordsPerTopic, 5, false));
        if (iteration < burninPeriod && M > 1) {                      // PTM:1166-1171
            double v = std::min((double)iteration / 100 + 0.3, 1.1);
            for (int i = 0; i < M; i++) std::fill(p_a[i].begin(), p_a[i].end(), v);
        } else if (iteration > burninPeriod && optimizeInterval != 0 && iteration % optimizeInterval == 0) {
            // PTM:1173-1210: statistics from device kernels, the closed forms and samplers are the reference's
            optimizeP(iteration + optimizeInterval > numIterations);                 // PTM:1176
            optimizeDP();                                                            // PTM:1184
            optimizeGamma();                                                         // PTM:1185
            optimizeBeta();                                                          // PTM:1186
            // buildFTrees(false) PTM:1209: mvhdp_sweep rebuilds the trees from the counts and the new hyper-parameters
        }
        pushHyper();
        mvhdp_sweep_stats st;
        const uint32_t sweepFlags = liveUpdates_ ? (MVHDP_SWEEP_LIVE | MVHDP_SWEEP_LIVE_SEGMENTS(liveSegments_))
                                  : segmentedUpdates_ ? (MVHDP_SWEEP_SEGMENT_APPLY | MVHDP_SWEEP_LIVE_SEGMENTS(liveSegments_)) : 0u;
        check(dev_.sweep((uint32_t)iteration, seed, sweepFlags, &st), "mvhdp_sweep");  // PTM:1213-1239
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        iterationLog.push_back({iteration, ms, st});
        if (iteration % 10 == 0 && printLogLikelihood) {               // PTM:1296-1304
            std::vector<double> ll = modelLogLikelihood();
            if (perplexities.size() != (size_t)M) perplexities.assign(M, std::vector<double>());
            for (int i = 0; i < M; i++) {
                if ((int)perplexities[i].size() <= iteration / 10) perplexities[i].resize(iteration / 10 + 1, 0.0);
                perplexities[i][iteration / 10] = ll[i] / totalTokens[i];
            }
        }
        if (st.activated_topic >= 0) {                                // keep the host copy of alpha / inActiveTopicIndex current
            std::vector<double> a((size_t)M * (numTopics + 1));
            std::vector<uint8_t> ina((size_t)numTopics);
            check(dev_.getAlpha(a.data(), ina.data()), "mvhdp_get_alpha");
            for (int m = 0; m < M; m++) std::copy(a.begin() + (size_t)m * (numTopics + 1), a.begin() + (size_t)(m + 1) * (numTopics + 1), alpha[m].begin());
            inActiveTopicIndex.clear();
            for (int k = 0; k < numTopics; k++) if (ina[k]) inActiveTopicIndex.insert(k);
        }
    }
    syncFromDevice(true);
}

}  // namespace mvtm

// ---------------------------------------------------------------------------
// extern "C" hooks so that the Python test/bench harness can drive the C++
// host class through ctypes (entity names are passed as int64 ids).
// ---------------------------------------------------------------------------
using mvtm::FastQMVWVParallelTopicModel;

static thread_local std::string g_host_err;

extern "C" {

const char* mvtm_last_error(void) { return g_host_err.c_str(); }
void mvtm_set_last_error(const char* msg) { g_host_err = msg ? msg : ""; }

int mvtm_model_set_device_gamma_statistics(void* p, int on)
{
    ((FastQMVWVParallelTopicModel*)p)->setDeviceGammaStatistics((on & 1) != 0);
    ((FastQMVWVParallelTopicModel*)p)->setDeviceTableStatistics((on & 2) != 0);          // (bit 1: optimizeDP's table simulation on the device)
    return 0;
}

int mvtm_model_set_shards(void* p, int n)
{
    ((FastQMVWVParallelTopicModel*)p)->setNumShards(n);
    return 0;
}

int mvtm_model_set_live_updates(void* p, int live, int tree_rebuilds_per_sweep)
{
    auto* m = (FastQMVWVParallelTopicModel*)p;
    if (live == 2) m->setSegmentedUpdates(true, tree_rebuilds_per_sweep);        // 2 = MVHDP_SWEEP_SEGMENT_APPLY
    else { m->setSegmentedUpdates(false); m->setLiveUpdates(live != 0, tree_rebuilds_per_sweep); }
    return 0;
}

void* mvtm_model_new(int K, int M, double alpha, double beta)
{
    try { return new FastQMVWVParallelTopicModel(K, (int8_t)M, alpha, beta); }
    catch (const std::exception& e) { g_host_err = e.what(); return nullptr; }
}

void mvtm_model_delete(void* p) { delete (FastQMVWVParallelTopicModel*)p; }

int mvtm_model_configure(void* p, int numIterations, int burninPeriod, int optimizeInterval, int randomSeed, int device, int64_t docIdBase)
{
    auto* m = (FastQMVWVParallelTopicModel*)p;
    m->setNumIterations(numIterations); m->setBurninPeriod(burninPeriod);
    m->setOptimizeInterval(optimizeInterval); m->setRandomSeed(randomSeed);
    m->setDevice(device); m->setDocIdBase(docIdBase);
    return 0;
}

// per view v: n_inst[v] instances with names name_ids[v][i], features tokens[v][off[v][i]..off[v][i+1])
int mvtm_model_add_instances(void* p, int M, const int64_t* n_inst, const int64_t* const* name_ids,
                             const int64_t* const* off, const int32_t* const* tokens, const int32_t* alphabet)
{
    auto* model = (FastQMVWVParallelTopicModel*)p;
    try {
        std::vector<mvtm::InstanceList> training(M);
        for (int v = 0; v < M; v++) {
            training[v].alphabetSize = alphabet[v];
            training[v].instances.resize((size_t)n_inst[v]);
            for (int64_t i = 0; i < n_inst[v]; i++) {
                training[v].instances[i].name = std::to_string(name_ids[v][i]);
                training[v].instances[i].features.assign(tokens[v] + off[v][i], tokens[v] + off[v][i + 1]);
            }
        }
        model->addInstances(training);
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return -1; }
}

int mvtm_model_estimate(void* p)
{
    try { ((FastQMVWVParallelTopicModel*)p)->estimate(); return 0; }
    catch (const std::exception& e) { g_host_err = e.what(); return -1; }
}

int64_t mvtm_model_num_entities(void* p) { return (int64_t)((FastQMVWVParallelTopicModel*)p)->data.size(); }

int64_t mvtm_model_view_tokens(void* p, int m)
{
    auto* model = (FastQMVWVParallelTopicModel*)p;
    int64_t n = 0;
    for (auto& e : model->data) n += (int64_t)e.Assignments[m].tokens.size();
    return n;
}

// entity names (as ids), CSR offsets, tokens and topics of view m in `data` order
int mvtm_model_get_view(void* p, int m, int64_t* entity_ids, int64_t* off, int32_t* tokens, int32_t* topics)
{
    auto* model = (FastQMVWVParallelTopicModel*)p;
    int64_t o = 0, d = 0;
    if (off) off[0] = 0;
    for (auto& e : model->data) {
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

int mvtm_model_get_counts(void* p, int m, int32_t* typeTopicCounts, int32_t* tokensPerTopic)
{
    auto* model = (FastQMVWVParallelTopicModel*)p;
    if (typeTopicCounts) std::copy(model->typeTopicCounts[m].begin(), model->typeTopicCounts[m].end(), typeTopicCounts);
    if (tokensPerTopic) std::copy(model->tokensPerTopic[m].begin(), model->tokensPerTopic[m].end(), tokensPerTopic);
    return 0;
}

int mvtm_model_get_log(void* p, int i, double* ms, mvhdp_sweep_stats* st)
{
    auto* model = (FastQMVWVParallelTopicModel*)p;
    if (i < 0 || i >= (int)model->iterationLog.size()) return -1;
    if (ms) *ms = model->iterationLog[i].ms;
    if (st) *st = model->iterationLog[i].stats;
    return 0;
}

int mvtm_model_optimize_p(void* p, double* p_a_out, double* pMean_out)
{
    auto* model = (FastQMVWVParallelTopicModel*)p;
    try {
        model->optimizeP(false);
        const int M = model->numModalities;
        for (int m = 0; m < M; m++) for (int i = 0; i < M; i++) {
            if (p_a_out) p_a_out[m * M + i] = model->p_a[m][i];
            if (pMean_out) pMean_out[m * M + i] = model->pMean[m][i];
        }
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return -1; }
}

int mvtm_model_optimize_beta(void* p, double* beta_out, double* betaSum_out)
{
    auto* model = (FastQMVWVParallelTopicModel*)p;
    try {
        model->optimizeBeta();
        for (int m = 0; m < model->numModalities; m++) { beta_out[m] = model->beta[m]; betaSum_out[m] = model->betaSum[m]; }
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return -1; }
}

int mvtm_model_seed_host_samplers(void* p, int64_t samp_seed, int64_t random_seed)
{
    ((FastQMVWVParallelTopicModel*)p)->seedHostSamplers(samp_seed, random_seed);
    return 0;
}

int mvtm_model_optimize_dp(void* p, double* alpha_out, double* alphaSum_out, uint8_t* inactive_out, double* tables_out)
{
    auto* mdl = (FastQMVWVParallelTopicModel*)p;
    try {
        mdl->optimizeDP();
        const int M = mdl->numModalities, K = mdl->numTopics;
        for (int m = 0; m < M; m++) {
            if (alpha_out) std::copy(mdl->alpha[m].begin(), mdl->alpha[m].end(), alpha_out + (size_t)m * (K + 1));
            if (alphaSum_out) alphaSum_out[m] = mdl->alphaSum[m];
            if (tables_out) tables_out[m] = mdl->tablesCnt[m];
        }
        if (tables_out) tables_out[M] = mdl->rootTablesCnt;
        if (inactive_out) for (int k = 0; k < K; k++) inactive_out[k] = mdl->inActiveTopicIndex.count(k) ? 1 : 0;
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return -1; }
}

int mvtm_model_optimize_gamma(void* p, double* gamma_out, double* gammaView_out, double* gammaRoot_out)
{
    auto* mdl = (FastQMVWVParallelTopicModel*)p;
    try {
        mdl->optimizeGamma();
        for (int m = 0; m < mdl->numModalities; m++) {
            if (gamma_out) gamma_out[m] = mdl->gamma[m];
            if (gammaView_out) gammaView_out[m] = mdl->gammaView[m];
        }
        if (gammaRoot_out) *gammaRoot_out = mdl->gammaRoot;
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return -1; }
}

// Known-answer hooks for the host-side samplers (fresh generator state per call)
int mvtm_cokus_stream(int n, uint32_t* out)
{
    mvtm::Cokus c;
    for (int i = 0; i < n; i++) out[i] = c.rand();
    return 0;
}

int mvtm_rand_antoniak_seq(int ncalls, const double* alpha, const int32_t* n, int32_t* out)
{
    mvtm::StaticSamplers s;
    for (int i = 0; i < ncalls; i++) {
        try { out[i] = s.randAntoniak(alpha[i], n[i]); }
        catch (const std::exception&) { out[i] = -1; }
    }
    return 0;
}

int mvtm_random_samplers_stream(int64_t seed, int kind, double a, double b, int n, double* out)
{
    mvtm::JavaRandom r(seed);
    mvtm::RandomSamplers<mvtm::JavaRandom> samp(&r);
    for (int i = 0; i < n; i++) {
        switch (kind) {
        case 0: out[i] = samp.randGamma(a); break;
        case 1: out[i] = samp.randBeta(a, b); break;
        case 2: out[i] = samp.randBernoulli(a); break;
        case 3: out[i] = samp.randGamma(a, b); break;
        default: return -1;
        }
    }
    return 0;
}

int mvtm_mallet_next_gamma_stream(int64_t seed, double alpha, double beta, int n, double* out)
{
    mvtm::Randoms r(seed);
    try { for (int i = 0; i < n; i++) out[i] = r.nextGamma(alpha, beta); }
    catch (const std::exception& e) { g_host_err = e.what(); return -1; }
    return 0;
}

int mvtm_model_log_likelihood(void* p, double* ll_out)
{
    auto* model = (FastQMVWVParallelTopicModel*)p;
    try {
        std::vector<double> ll = model->modelLogLikelihood();
        for (size_t m = 0; m < ll.size(); m++) ll_out[m] = ll[m];
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return -1; }
}

// perplexities[m][iteration/10] (LL/token, PTM:1302-1303); returns the number of entries written per view
int mvtm_model_get_perplexities(void* p, int m, double* out, int cap)
{
    auto* model = (FastQMVWVParallelTopicModel*)p;
    if (m < 0 || m >= (int)model->perplexities.size()) return 0;
    int n = std::min<int>(cap, (int)model->perplexities[m].size());
    for (int i = 0; i < n; i++) out[i] = model->perplexities[m][i];
    return n;
}

int mvtm_model_print_document_topics(void* p, const char* filename, double threshold, int max, const double* discr_weight, const double* p_mean)
{
    auto* mdl = (FastQMVWVParallelTopicModel*)p;
    try {
        const int M = mdl->numModalities;
        if (discr_weight) mdl->discrWeightPerModality.assign(discr_weight, discr_weight + M);
        if (p_mean) {
            mdl->pMean.assign(M, std::vector<double>(M, 0.0));
            for (int a = 0; a < M; a++) for (int b = 0; b < M; b++) mdl->pMean[a][b] = p_mean[a * M + b];
        }
        mdl->printDocumentTopics(filename, threshold, max);
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return -1; }
}

int mvtm_model_display_top_words(void* p, int numWords, int usingNewLines, char* out, int cap)
{
    try {
        const std::string t = ((FastQMVWVParallelTopicModel*)p)->displayTopWords(numWords, 5, usingNewLines != 0);
        if (out && cap > 0) { const int n = std::min<int>((int)t.size(), cap - 1); memcpy(out, t.data(), (size_t)n); out[n] = 0; }
        return (int)t.size();
    } catch (const std::exception& e) { g_host_err = e.what(); return -1; }
}

int mvtm_number_format5(double v, char* out, int cap)
{
    const std::string t = FastQMVWVParallelTopicModel::numberFormat5(v);
    if (cap <= (int)t.size()) return -1;
    memcpy(out, t.c_str(), t.size() + 1);
    return (int)t.size();
}

int mvtm_model_print_state(void* p, const char* filename)
{
    try { ((FastQMVWVParallelTopicModel*)p)->printState(filename); return 0; }
    catch (const std::exception& e) { g_host_err = e.what(); return -1; }
}

int mvtm_java_double_to_string(double v, char* out, int cap)
{
    std::string s = FastQMVWVParallelTopicModel::javaDoubleToString(v);
    if ((int)s.size() + 1 > cap) return -1;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
}

void* mvtm_model_native_handle(void* p) { return ((FastQMVWVParallelTopicModel*)p)->nativeHandle(); }

// PTM:465-515 on a flattened corpus: the same draw order as addInstances (for harnesses that
// already hold CSR arrays, e.g. bench.py on a 1M-entity synthetic corpus).
int mvtm_init_assignments(int K, int M, int64_t D, const int64_t* const* doc_off, int64_t seed, int32_t* const* z_out)
{
    mvtm::Randoms random(seed);
    std::vector<int32_t> activeTopics;
    for (int64_t d = 0; d < D; d++) {
        for (int m = 0; m < M; m++) {
            if (m == 0) activeTopics.clear();
            for (int64_t i = doc_off[m][d]; i < doc_off[m][d + 1]; i++) {
                int topic;
                if (m == 0) { topic = random.nextInt(K); activeTopics.push_back(topic); }
                else if (!activeTopics.empty()) topic = activeTopics[random.nextInt((int32_t)activeTopics.size())];
                else topic = random.nextInt(K);
                z_out[m][i] = topic;
            }
        }
    }
    return 0;
}

}  // extern "C"
