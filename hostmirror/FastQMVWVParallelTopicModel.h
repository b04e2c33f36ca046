// FastQMVWVParallelTopicModel.h — C++ host-side mirror of the hot subset of
// org.madgik.MVTopicModel.FastQMVWVParallelTopicModel (PTM).  The reference's
// host language (Java) has no toolchain in the build image, so the host side
// above the C ABI is written in C++ with the reference's names, argument
// meaning and call order: ctor PTM:183, setters PTM:273-335, addInstances
// PTM:396, estimate PTM:1033.  The iteration body of estimate()
// ("submit updaters + submit workers + barrier.await()", PTM:1213-1239) is one
// mvhdp_sweep() call on the GPU.  The Java/JNI form of the same class is in
// INTEGRATION.md.
#pragma once
#include <cstdint>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "../include/mvhdp.h"
#include "device_model.h"
#include "java_random.h"
#include "knowceans_samplers.h"

namespace mvtm {

class FastQMVWVTopicInferencer;

// MALLET Instance restricted to what the path reads: getName() (PTM:437) and the
// FeatureSequence indices of getData() (PTM:427).
struct Instance {
    std::string name;
    std::vector<int32_t> features;
    std::string source;                   // getSource(), "NA" when empty (PTM:3293-3296)
};

// MALLET InstanceList: instances + getDataAlphabet().size() (PTM:412-413).
struct InstanceList {
    std::vector<Instance> instances;
    int32_t alphabetSize = 0;
    std::vector<std::string> alphabet;    // optional: getDataAlphabet().lookupObject(type) for printState (PTM:3303)
};

// cc.mallet.topics.TopicAssignment: instance + topicSequence (LabelSequence features).
struct TopicAssignment {
    std::string source;                   // instance.getSource()
    bool present = false;                 // false == null (MTA:19)
    std::vector<int32_t> tokens;          // instance.getData()
    std::vector<int32_t> topics;          // topicSequence.getFeatures(), sized by getLength()
};

// org.madgik.utils.MixTopicModelTopicAssignment (MTA:11-43)
struct MixTopicModelTopicAssignment {
    std::string EntityId;
    std::vector<TopicAssignment> Assignments;   // [numModalities]
};

struct IterationLog {
    int iteration;
    double ms;                                   // PTM:1272-1277
    mvhdp_sweep_stats stats;
};

class FastQMVWVParallelTopicModel {
public:
    static constexpr int UNASSIGNED_TOPIC = -1;  // PTM:63

    // PTM:183.  useCycleProposals / SQLConnectionString / useTypeVectors / vectorsLambda /
    // trainTypeVectors belong to subsystems outside the hot path; they are accepted for
    // signature compatibility and must be false/empty/0 (embedding mix off, FLOW:68-69).
    FastQMVWVParallelTopicModel(int numberOfTopics, int8_t numModalities, double alpha, double beta,
                                bool useCycleProposals = false, const std::string& SQLConnectionString = "",
                                bool useTypeVectors = false, double vectorsLambda = 0, bool trainTypeVectors = false);
    ~FastQMVWVParallelTopicModel();
    FastQMVWVParallelTopicModel(const FastQMVWVParallelTopicModel&) = delete;
    FastQMVWVParallelTopicModel& operator=(const FastQMVWVParallelTopicModel&) = delete;

    // PTM:273-335
    void setNumIterations(int n) { numIterations = n; }
    void setBurninPeriod(int n) { burninPeriod = n; }
    void setRandomSeed(int seed) { randomSeed = seed; }
    void setOptimizeInterval(int interval) { optimizeInterval = interval; }
    void setNumThreads(int threads) { numThreads = threads; }   // a hint only: parallelism is the GPU's
    void setDevice(int device) { device_ = device; }
    void setDocIdBase(int64_t base) { docIdBase_ = base; }      // document shards: global id of entity 0
    // the model as n document shards behind an mvhdp_group (include/mvhdp.h), all on the chosen device: the same chain, the same
    // statistics -- every call of this class is routed to its mvhdp_group_* counterpart (device_model.h).  Before addInstances().
    void setNumShards(int n) { numShards_ = n < 1 ? 1 : n; }
    int numShards() const { return numShards_; }
    // Update discipline of the sweeps estimate() runs: false = deferred (snapshot sweep, bit-reproducible: the parity
    // contract), true = MVHDP_SWEEP_LIVE (the updater threads' own discipline, UPD:197-218; profiles/r02_ll_curves.md)
    void setLiveUpdates(bool live, int treeRebuildsPerSweep = 0) { liveUpdates_ = live; liveSegments_ = treeRebuildsPerSweep; if (live) segmentedUpdates_ = false; }
    // the deterministic middle ground: MVHDP_SWEEP_SEGMENT_APPLY with this many segments (0 = library default)
    void setSegmentedUpdates(bool on, int segments = 0) { segmentedUpdates_ = on; liveSegments_ = segments; if (on) liveUpdates_ = false; }
    // optimizeGamma's document-level sums (PTM:2415-2433): false = the reference's sequential host loop over every entity,
    // ten rounds per view (reproducible against the Python oracle under an injected stream; 1.9 s per call at C4),
    // true = mvhdp_gamma_doc_statistics on the device (the same random variables in distribution; milliseconds)
    void setDeviceGammaStatistics(bool on) { deviceGammaStatistics_ = on; }
    // true = optimizeDP's view-table simulation through mvhdp_dp_table_statistics (the Antoniak draws on the device; the default is the reference's loop)
    void setDeviceTableStatistics(bool on) { deviceTableStatistics_ = on; }

    // PTM:396.  batchId / vectorSize / previousModel are outside the hot path (previousModel must be null).
    void addInstances(const std::vector<InstanceList>& training, const std::string& batchId = "", int vectorSize = 0);

    // PTM:1033.  Blocks until numIterations sweeps are done.  Throws std::runtime_error on a device error.
    void estimate();

    // state the reference exposes as public fields
    std::vector<MixTopicModelTopicAssignment> data;             // PTM:67
    int numTopics;                                              // PTM:72
    int8_t numModalities;                                       // PTM:71
    std::vector<int> numTypes;                                  // PTM:78
    std::vector<std::vector<double>> alpha;                     // PTM:79  [M][K+1]
    std::vector<double> alphaSum, beta, betaSum, gamma;         // PTM:80-83
    std::vector<std::vector<int32_t>> typeTopicCounts;          // PTM:86  per view, [V_m*K] row-major
    std::vector<std::vector<int32_t>> tokensPerTopic;           // PTM:87  [M][K]
    std::vector<int> totalTokens;                               // PTM:89
    std::vector<int> totalDocsPerModality;                      // PTM:90
    std::vector<std::vector<int32_t>> docLengthCounts;          // PTM:107
    std::vector<std::vector<int32_t>> topicDocCounts;           // PTM:108 per view, [K*(histogramSize+1)]
    std::vector<int> histogramSize;                             // PTM:109
    std::set<int> inActiveTopicIndex;                           // PTM:95
    std::vector<std::vector<double>> p_a, p_b;                  // PTM:130-131
    std::vector<std::vector<int>> typeTotals;                   // PTM:169
    std::vector<int> maxTypeCount;                              // PTM:171
    int numIterations = 1000;                                   // PTM:111
    int burninPeriod = 200;                                     // PTM:112
    int optimizeInterval = 50;                                  // PTM:114
    int randomSeed = -1;                                        // PTM:126
    int numThreads = 1;                                         // PTM:173

    // what the log lines of PTM:1272-1310 would have shown
    std::vector<IterationLog> iterationLog;
    mvhdp_handle nativeHandle() const { return h_; }

    // refresh typeTopicCounts / tokensPerTopic / topicDocCounts / data[].topics from the device
    void syncFromDevice(bool histograms = true);

    // SURVEY §8f #1/#2 (the steps either side of the sweep).  The statistics come from device kernels;
    // the closed-form updates are the reference's.
    void optimizeP(bool appendMetadata = false);               // PTM:2698-2819
    void optimizeBeta();                                       // PTM:2288-2367
    std::vector<double> modelLogLikelihood();                  // PTM:3322-3452
    // The two randomised steps.  Their statistics (topicDocCounts) come from a device kernel; the samplers are
    // the reference's own (knowceans_samplers.h, java_random.h).  RNG streams: Samplers.randAntoniak draws from
    // the JVM-wide Cokus generator (self-seeded with 4357, reproduced as is); `samp` runs over
    // ThreadLocalRandom and `random` is an unseeded Randoms in the reference (PTM:236,241-246) -- neither can
    // be seeded there, so this build feeds both from java.util.Random streams derived from randomSeed.
    void optimizeDP();                                         // PTM:2440-2591
    void optimizeGamma();                                      // PTM:2369-2438
    std::vector<double> sampleDirichlet(const std::vector<double>& p);   // PTM:2593-2634
    std::vector<double> tablesCnt;                              // PTM:136
    double rootTablesCnt = 0;                                   // PTM:137
    std::vector<double> gammaView;                              // PTM:138
    double gammaRoot = 10;                                      // PTM:139
    StaticSamplers Samplers;                                    // the static half of org.knowceans.util.Samplers
    void seedHostSamplers(int64_t sampSeed, int64_t randomSeed64);
    std::vector<std::vector<double>> pMean;                     // PTM:134
    std::vector<std::vector<double>> perplexities;              // PTM:144  [M][iteration/10] = LL/token
    bool printLogLikelihood = true;                             // PTM:128
    std::vector<std::string> notes;

    // printDocumentTopics PTM:2820-2960 without its JDBC half: per entity the topic proportions
    //   sum_m (m == 0 ? 1 : discrWeightPerModality[m]) * pMean[0][m] * (n_dk + gamma*alpha) / (len + gamma*alphaSum) / sum_m (...)
    // (device kernel), sorted by IDSorter.compareTo (descending weight, ties by descending topic id), cut at `threshold` and `max`.  The text is the
    // reference's, including its habit of printing the growing line once per retained topic (PTM:2905-2909).
    // discrWeightPerModality comes from the diagnostics (PTM:1370), which are outside this build: all 1 unless set.
    std::vector<double> discrWeightPerModality;                 // PTM:164
    std::string printDocumentTopicsToString(double threshold, int max);
    void printDocumentTopics(const std::string& filename, double threshold, int max);
    // PTM:3457-3463: a tool for estimating topic distributions of new documents with this model frozen
    std::unique_ptr<FastQMVWVTopicInferencer> getInferencer();

    // getSortedWords PTM:1792-1811 / displayTopWords PTM:1852-1890, what estimate() logs every showTopicsInterval
    // iterations (PTM:1150-1152).  Order: cc.mallet.types.IDSorter.compareTo of MALLET 2.0.8 (count descending, equal
    // counts by DESCENDING type id); alpha through java.text.NumberFormat.getInstance() with at most 5 fraction digits
    // (PTM:221-222; en-US: grouping commas, HALF_EVEN).  Words print as alphabet strings, or the type index when none.
    std::vector<std::vector<std::pair<int32_t, int32_t>>> getSortedWords(int modality);   // per topic: (type, count)
    std::string displayTopWords(int numWords, int numLabels, bool usingNewLines);
    static std::string numberFormat5(double v);
    int showTopicsInterval = 50;                                // PTM:117
    int wordsPerTopic = 15;                                     // PTM:118
    std::vector<std::pair<int, std::string>> topWordsLog;       // (iteration, text) of the PTM:1151 log lines

    // the text state format of printState (PTM:3269-3320); gz when the name ends in ".gz".
    // Java's Double.toString is approximated by the shortest round-trip decimal in Java's layout.
    void printState(const std::string& filename);
    std::string printStateToString();
    static std::string javaDoubleToString(double v);
    std::vector<std::vector<std::string>> alphabet;             // PTM:68 (strings), empty = print the type index

    // MALLET 2.0.8 arithmetic used by optimizeBeta (restated from the jar's bytecode, see tools/javap_lite.py)
    static double digamma(double z);
    static double learnSymmetricConcentration(const std::vector<int32_t>& countHistogram, const std::vector<int32_t>& observationLengths,
                                              int numDimensions, double currentValue);

private:
    void initializeHistograms();          // PTM:849-897
    void pushHyper();
    void check(int rc, const char* what);
    void ensureHostSamplers();
    JavaRandom sampRand_{0};                                    // stands in for ThreadLocalRandom.current()
    Randoms random_{0};                                         // the ctor's `random` field (PTM:241-246)
    bool hostSamplersSeeded_ = false;
    DeviceModel dev_;                   // one handle, or document shards + their group
    mvhdp_handle h_ = nullptr;          // = dev_.first(): the replicated model (and the whole model when there is one shard)
    int numShards_ = 1;
    int device_ = 0;
    int64_t docIdBase_ = 0;
    bool liveUpdates_ = false, segmentedUpdates_ = false;
    int liveSegments_ = 0;
    bool deviceGammaStatistics_ = false, deviceTableStatistics_ = false;
    uint32_t gammaCalls_ = 0;
};

}  // namespace mvtm
