/*
 * mvtm_host.h — extern "C" hooks around the C++ host-side mirror of
 * FastQMVWVParallelTopicModel (mvtopicmodel_amd/csrc/host/).  These exist so
 * that a harness without a C++ compiler (the Python tests, bench.py) can drive
 * the host class the way a MALLET client drives the reference:
 *   new FastQMVWVParallelTopicModel(K, M, alpha, beta)   PTM:183
 *   setNumIterations/…/setRandomSeed                      PTM:273-335
 *   addInstances(InstanceList[] training, …)              PTM:396
 *   estimate()                                            PTM:1033
 * Entity names (Instance.getName(), PTM:437) cross as int64 ids.
 * The drop-in boundary itself is include/mvhdp.h.
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
int   mvtm_model_get_perplexities(void* model, int m, double* out, int cap);
/* printState PTM:3269-3320 (text; gzip when the name ends in .gz) */
int   mvtm_model_print_state(void* model, const char* filename);
int   mvtm_java_double_to_string(double v, char* out, int cap);
void* mvtm_model_native_handle(void* model);
/* PTM:465-515 on CSR arrays: initial topic draw order of addInstances with java.util.Random(seed) */
int   mvtm_init_assignments(int K, int M, int64_t D, const int64_t* const* doc_off, int64_t seed, int32_t* const* z_out);

#ifdef __cplusplus
}
#endif
#endif
