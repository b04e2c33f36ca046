/*
 * mvtm_host.h — extern "C" hooks around the C++ host-side mirror of
 * FastQMVWVParallelTopicModel (hostmirror/).  These exist so
 * that a harness without a C++ compiler (the Python tests, bench.py) can drive
 * the host class the way a MALLET client drives the reference:
 *   new FastQMVWVParallelTopicModel(K, M, alpha, beta)   PTM:183
 *   setNumIterations/…/setRandomSeed                      PTM:273-335
 *   addInstances(InstanceList[] training, …)              PTM:396
 *   estimate()                                            PTM:1033
 * Entity names (Instance.getName(), PTM:437) cross as int64 ids.
 * The drop-in boundary itself is include/mvhdp.h (libmvhdp.so: kernels + C ABI only).  This host mirror is a
 * SEPARATE library, libmvtm_host.so, built from hostmirror/ and linked against libmvhdp.so: it
 * stands in for the Java host that cannot be compiled here, it is not part of the product library.
 */
#ifndef MVTM_HOST_H
#define MVTM_HOST_H
#include <stdint.h>
#include "mvhdp.h"
#ifdef __cplusplus
extern "C" {
#endif

const char* mvtm_last_error(void);
void* mvtm_model_new(int K, int M, double alpha, double beta);
void  mvtm_model_delete(void* model);
int   mvtm_model_configure(void* model, int numIterations, int burninPeriod, int optimizeInterval,
                           int randomSeed, int device, int64_t docIdBase);
int   mvtm_model_add_instances(void* model, int M, const int64_t* n_inst, const int64_t* const* name_ids,
                               const int64_t* const* off, const int32_t* const* tokens, const int32_t* alphabet);
int   mvtm_model_estimate(void* model);
int64_t mvtm_model_num_entities(void* model);
int64_t mvtm_model_view_tokens(void* model, int m);
int   mvtm_model_get_view(void* model, int m, int64_t* entity_ids, int64_t* off, int32_t* tokens, int32_t* topics);
int   mvtm_model_get_counts(void* model, int m, int32_t* typeTopicCounts, int32_t* tokensPerTopic);
int   mvtm_model_get_log(void* model, int i, double* ms, mvhdp_sweep_stats* st);
/* optimizeP PTM:2698-2819, optimizeBeta PTM:2288-2367, modelLogLikelihood PTM:3322-3452 */
int   mvtm_model_optimize_p(void* model, double* p_a_out /*[M][M]*/, double* pMean_out /*[M][M]*/);
int   mvtm_model_optimize_beta(void* model, double* beta_out /*[M]*/, double* betaSum_out /*[M]*/);
int   mvtm_model_log_likelihood(void* model, double* ll_out /*[M]*/);
/* optimizeDP PTM:2440-2591 and optimizeGamma PTM:2369-2438.  tables_out = tablesCnt[0..M-1], rootTablesCnt.
 * mvtm_model_seed_host_samplers seeds the two java.util.Random streams that stand in for the reference's
 * unseedable ones (`samp` over ThreadLocalRandom PTM:236; the ctor's unseeded Randoms PTM:241-246). */
int   mvtm_model_seed_host_samplers(void* model, int64_t samp_seed, int64_t random_seed);
int   mvtm_model_optimize_dp(void* model, double* alpha_out /*[M][K+1]*/, double* alphaSum_out /*[M]*/,
                             uint8_t* inactive_out /*[K]*/, double* tables_out /*[M+1]*/);
int   mvtm_model_optimize_gamma(void* model, double* gamma_out /*[M]*/, double* gammaView_out /*[M]*/, double* gammaRoot_out);
/* known-answer hooks for the host-side samplers those two steps use (fresh generator state per call):
 * Cokus.java (MT19937, self-seeded 4357), Samplers.randAntoniak (static Stirling cache, -1 = the call threw),
 * RandomSamplers.randGamma(a) / randBeta(a,b) / randBernoulli(a) / randGamma(a,b) = kind 0..3 over
 * java.util.Random(seed), MALLET Randoms(seed).nextGamma(alpha, beta). */
int   mvtm_cokus_stream(int n, uint32_t* out);
int   mvtm_rand_antoniak_seq(int ncalls, const double* alpha, const int32_t* n, int32_t* out);
int   mvtm_random_samplers_stream(int64_t seed, int kind, double a, double b, int n, double* out);
int   mvtm_mallet_next_gamma_stream(int64_t seed, double alpha, double beta, int n, double* out);
int   mvtm_model_get_perplexities(void* model, int m, double* out, int cap);
/* printState PTM:3269-3320 (text; gzip when the name ends in .gz) */
/* displayTopWords PTM:1852-1890 (returns the text length; copies at most cap-1 bytes) and the NumberFormat it uses */
int   mvtm_model_display_top_words(void* model, int numWords, int usingNewLines, char* out, int cap);
int   mvtm_number_format5(double v, char* out, int cap);
int   mvtm_model_print_state(void* model, const char* filename);
/* printDocumentTopics PTM:2820-2960 (text half; no JDBC): discr_weight[M] / p_mean[M][M] replace the model's when non-NULL */
int   mvtm_model_print_document_topics(void* model, const char* filename, double threshold, int max,
                                       const double* discr_weight, const double* p_mean);
int   mvtm_java_double_to_string(double v, char* out, int cap);
void* mvtm_model_native_handle(void* model);
/* update discipline of estimate()'s sweeps: 0 = deferred (parity contract), 1 = MVHDP_SWEEP_LIVE (UPD:197-218),
 * 2 = MVHDP_SWEEP_SEGMENT_APPLY (deterministic, segments applied in between); second argument = segments (0 = default) */
int   mvtm_model_set_live_updates(void* model, int live, int tree_rebuilds_per_sweep);
/* before add_instances: keep the model as n document shards behind an mvhdp_group (all on the model's device); estimate(), the
 * optimize* steps and the log-likelihood then run over the group (mvhdp_group_*), same integers as one handle */
int   mvtm_model_set_shards(void* model, int n);
/* optimizeGamma's per-entity Bernoulli / Beta sums (PTM:2415-2433): 0 = sequential host loop (the reference's), 1 = device kernel */
int   mvtm_model_set_device_gamma_statistics(void* model, int on);
/* FastQMVWVTopicInferencer (INF:114-330) as one call chain: getInferencer() PTM:3457, then
 * inferTopicDistributionsOnNewDocs = align views by name, trees without gamma*alpha, tree-sampled initial topics,
 * numIterations (10) frozen sweeps, printDocumentTopics(out, 0.03, -1) text.  Returns the text length (-1 on error). */
void* mvtm_model_get_inferencer(void* model, const double* discr_weight /*[M] or NULL*/, const double* p_mean /*[M][M] or NULL*/);
void  mvtm_inferencer_delete(void* inferencer);
int   mvtm_inferencer_configure(void* inferencer, int numIterations, int randomSeed, int device);
int64_t mvtm_inferencer_infer(void* inferencer, int M, const int64_t* n_inst, const int64_t* const* name_ids,
                              const int64_t* const* off, const int32_t* const* tokens, char* text_out, int64_t cap);
int64_t mvtm_inferencer_num_entities(void* inferencer);
int64_t mvtm_inferencer_view_tokens(void* inferencer, int m);
int   mvtm_inferencer_get_view(void* inferencer, int m, int64_t* entity_ids, int64_t* off, int32_t* tokens, int32_t* topics);
int   mvtm_inferencer_doc_topics(void* inferencer, double* out /*[D][K]*/);
int64_t mvtm_inferencer_print_document_topics(void* inferencer, double threshold, int max, char* text_out, int64_t cap);
int   mvtm_inferencer_get_stats(void* inferencer, int i, mvhdp_sweep_stats* st);
/* PTM:465-515 on CSR arrays: initial topic draw order of addInstances with java.util.Random(seed) */
int   mvtm_init_assignments(int K, int M, int64_t D, const int64_t* const* doc_off, int64_t seed, int32_t* const* z_out);

#ifdef __cplusplus
}
#endif
#endif
