// FastQMVWVTopicInferencer.h — C++ host-side mirror of org.madgik.MVTopicModel.FastQMVWVTopicInferencer (INF), the
// second caller of the sweep: fold-in inference on new documents with the trained model frozen.
//   ctor / getInferencer()                INF:75-101, PTM:3457-3463   (the model's counts and hyper-parameters)
//   initInferencer()                      INF:557-586   trees with leaves p_wt alone -> mvhdp_build_inference_trees
//   inferTopicDistributionsOnNewDocs()    INF:114-330   align views by name, topics drawn from the trees
//                                                       (mvhdp_init_assignments_from_trees), numIterations sweeps with
//                                                       nst = 1, nut = 0 (MVHDP_SWEEP_FROZEN), printDocumentTopics
//   printDocumentTopics()                 INF:332-490   text half (no JDBC); proportions from the device kernel
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "FastQMVWVParallelTopicModel.h"

namespace mvtm {

class FastQMVWVTopicInferencer {
public:
    // INF:75.  pipes / docSmoothingOnlyMass / docSmoothingOnlyCumValues carry nothing the path reads and are dropped.
    FastQMVWVTopicInferencer(const std::vector<int>& numTypes, const std::vector<std::vector<double>>& alpha,
                             const std::vector<double>& alphaSum, const std::vector<std::vector<int32_t>>& typeTopicCounts,
                             const std::vector<std::vector<int32_t>>& tokensPerTopic, const std::vector<double>& beta,
                             const std::vector<double>& betaSum, const std::vector<double>& gamma, int numTopics, int8_t numModalities,
                             const std::vector<std::vector<double>>& p_a, const std::vector<std::vector<double>>& p_b,
                             const std::vector<double>& discrWeightPerModality, const std::vector<std::vector<double>>& pMean);
    ~FastQMVWVTopicInferencer();
    FastQMVWVTopicInferencer(const FastQMVWVTopicInferencer&) = delete;
    FastQMVWVTopicInferencer& operator=(const FastQMVWVTopicInferencer&) = delete;

    void setRandomSeed(int seed) { randomSeed = seed; }          // INF:103-105
    void setNumIterations(int n) { numIterations = n; }
    void setDevice(int device) { device_ = device; }

    // INF:114.  Returns the text printDocumentTopics(out, 0.03, -1, ...) writes (INF:326-329).
    std::string inferTopicDistributionsOnNewDocs(const std::vector<InstanceList>& training);
    std::string printDocumentTopicsToString(double threshold, int max);          // INF:332-490, text half
    std::vector<double> docTopicProportions();                                    // [D][K], INF:383-411

    std::vector<MixTopicModelTopicAssignment> data;              // INF:70
    int numTopics;
    int8_t numModalities;
    std::vector<int> numTypes;
    std::vector<std::vector<double>> alpha;
    std::vector<double> alphaSum, beta, betaSum, gamma;
    std::vector<std::vector<int32_t>> typeTopicCounts, tokensPerTopic;
    std::vector<std::vector<double>> p_a, p_b, pMean;
    std::vector<double> discrWeightPerModality;
    int numIterations = 10;                                      // INF:73, INF:561
    int randomSeed = -1;
    std::vector<mvhdp_sweep_stats> iterationStats;

private:
    void check(int rc, const char* what);
    mvhdp_handle h_ = nullptr;
    int device_ = 0;
};

// PTM:2862-2909 / INF:352-420: the text of printDocumentTopics for the entities of `h` (names[d] = EntityId).
std::string formatDocumentTopics(mvhdp_handle h, const std::vector<std::string>& names, int K, const std::vector<double>& w,
                                 double threshold, int max);

}  // namespace mvtm
