/*
 * mvhdp_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the multi-view HDP collapsed-Gibbs sweep of
 * hmetaxa/MVTopicModel (FastQMVWVWorkerRunnable / FastQMVWVUpdaterRunnable /
 * FTree behind FastQMVWVParallelTopicModel.estimate()).  Every function cites
 * the reference file:line it follows (aliases as in SURVEY.md:
 * WRK / UPD / FT / QD / PTM).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product (libmvhdp.so) never
 * links, loads or calls it.
 *
 * PARITY PINNING: the reference ships no tests, fixtures or golden vectors and
 * cannot be run here (Java, no JVM in the image).  The oracle is pinned by the
 * hand-derived known-answer tests of SURVEY.md §8c (KAT-1..6), the documented
 * java.util.Random contract, the Random123 Philox4x32-10 known answers and
 * count-conservation invariants — see tests/test_oracle_kats.py.  With no
 * reference-produced vectors to check against: "parity unpinned" beyond those.
 */
#ifndef MVHDP_ORACLE_H
#define MVHDP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_M 8

/* ---------------- primitives (KAT surface) ---------------- */

/* Random123 Philox4x32-10. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* The sweep's token stream (contract shared with the HIP kernel): one Philox
 * call per token, ctr = (pos, view, doc_lo, sweep) , key = (seed_lo, seed_hi ^ doc_hi);
 * u1 from words 0,1 and u2 from words 2,3, each ((hi<<32|lo)>>11) * 2^-53. */
void orc_token_uniforms(uint64_t seed, uint32_t sweep, int64_t doc, int view, uint32_t pos,
                        double* u1, double* u2);

/* java.util.Random (documented 48-bit LCG), as used through MALLET Randoms
 * (PTM:404-408, PTM:1067-1072). */
typedef struct { uint64_t s; int have_gauss; double next_gauss; } orc_jrand;
void    orc_jrand_seed(orc_jrand* r, int64_t seed);
int32_t orc_jrand_next(orc_jrand* r, int bits);
int32_t orc_jrand_next_int(orc_jrand* r, int32_t bound);
double  orc_mallet_next_uniform(orc_jrand* r);           /* Randoms.nextUniform */
double  orc_mallet_next_gaussian(orc_jrand* r);          /* Randoms.nextGaussian */
double  orc_mallet_next_beta(orc_jrand* r, double a, double b); /* Randoms.nextBeta, WRK:333 */

/* FTree (FT:57-147).  tree has 2*size doubles. */
void orc_ftree_construct(double* tree, int size, const double* weights); /* FT:96-109 */
int  orc_ftree_sample(const double* tree, int size, double u);           /* FT:111-136; -2 if u>1 */
void orc_ftree_update(double* tree, int size, int topic, double v);      /* FT:138-147 */

/* WRK:257-277 */
int orc_lower_bound(const double* arr, double key, int len);

/* Java Math.round(double) */
int64_t orc_java_round(double x);

/* ---------------- model ---------------- */

typedef struct orc_model orc_model;

typedef struct {
    int64_t tokens;               /* tokens visited (OOV skipped ones excluded) */
    int64_t changed;              /* new != old (deltas emitted, WRK:587) */
    int64_t new_mass_cnt;         /* WRK:523 */
    int64_t topic_doc_mass_cnt;   /* WRK:530 */
    int64_t word_ftree_mass_cnt;  /* WRK:533 */
    int64_t oov_skipped;          /* WRK:427-428 */
    int64_t aborted_docs;         /* Q11: exception inside a doc */
    int32_t activated_topic;      /* UPD:263-270, -1 if none */
    int32_t activated_modality;
    int64_t activation_key;       /* (global doc<<34 | view<<31 | pos<<11 | topic) of that first delta, INT64_MAX if none */
} orc_stats;

orc_model* orc_create(int K, int M, const int32_t* V);
void       orc_destroy(orc_model* o);

/* corpus: CSR per view; a missing view is an empty span.  Copies. */
int  orc_set_corpus(orc_model* o, int m, int64_t D, const int64_t* doc_off, const int32_t* tokens);
int  orc_set_assignments(orc_model* o, int m, const int32_t* z);
int  orc_get_assignments(const orc_model* o, int m, int32_t* z);

/* hyper-parameters: alpha[M][K+1]; per-view scalars [M]; p_a/p_b [M][M];
 * inactive[K] (1 = member of inActiveTopicIndex) or NULL. */
void orc_set_hyper(orc_model* o, const double* alpha, const double* alpha_sum,
                   const double* beta, const double* beta_sum, const double* gamma,
                   const double* p_a, const double* p_b, const uint8_t* inactive);
void orc_get_alpha(const orc_model* o, double* alpha);
void orc_get_inactive(const orc_model* o, uint8_t* inactive);

/* PTM:465-515 — random init of z with java.util.Random(seed). */
void orc_init_assignments(orc_model* o, int64_t seed);
/* PTM:600-652 */
void orc_build_counts(orc_model* o);
/* PTM:2660-2696 */
void orc_build_trees(orc_model* o);
void orc_build_inference_trees(orc_model* o);                                            /* INF:557-586 */
void orc_init_assignments_from_trees(orc_model* o, uint64_t seed, int64_t doc_id_base);  /* INF:169-199 */
void orc_get_counts(const orc_model* o, int m, int32_t* nwk, int32_t* nk);
void orc_set_counts(orc_model* o, int m, const int32_t* nwk, const int32_t* nk);
void orc_get_tree(const orc_model* o, int m, int w, double* tree2K);
/* topicDocCounts[m][k][c] for c<hist_len (PTM:647-649 semantics, recomputed
 * from z) and docLengthCounts[m][len] (PTM:626). */
void orc_get_doc_topic_hist(const orc_model* o, int m, int32_t* hist, int32_t hist_len,
                            int32_t* doc_len_counts, int32_t len_len);

/* WRK:327-337 with the worker's own MALLET Randoms (seeded LCG): fills
 * p[D][M][M] for docs [0,D) in order, continuing the stream in *r. */
void orc_draw_p_mallet(const orc_model* o, orc_jrand* r, double* p);
/* The device contract: the same nextBeta algorithm fed by a per-(doc,pair)
 * Philox uniform stream (no state carried between docs). */
void orc_draw_p_philox(const orc_model* o, uint64_t seed, uint32_t sweep, int64_t doc_id_base, double* p);

#define ORC_SWEEP_REUSE_TREES 1u  /* do not rebuild trees from the snapshot first */
#define ORC_SWEEP_NO_APPLY    2u  /* leave n_wk/n_k untouched; deltas returned */
#define ORC_SWEEP_FROZEN      16u /* the inferencer's mode (INF:211-212 nst=1, nut=0): stored trees, no deltas at all */

/* One deferred-update sweep (SURVEY §7 "hard parts": every token sampled
 * against the sweep-start n_wk, n_k, trees; deltas applied afterwards in
 * (doc, view, position) order as a single updater would, UPD:181-272).
 * p: D*M*M view weights, or NULL to draw them with orc_draw_p_philox.
 * delta_nwk/delta_nk: optional outputs (sumV*K, M*K).
 * tok_dbg[m]: optional, 4 doubles per token {newTopicMass, topicDocWordMass,
 * tree root, sample}.
 * trace: optional, n_trace tokens (doc,view,pos) whose full K+1 conditional
 * (normalised) is written to trace_out[n_trace][K+1]; slot K = new-topic mass. */
int orc_sweep(orc_model* o, uint32_t sweep_idx, uint64_t seed, int64_t doc_id_base,
              const double* p, uint32_t flags, orc_stats* st,
              int32_t* delta_nwk, int32_t* delta_nk,
              double* const* tok_dbg,
              int n_trace, const int64_t* trace_doc, const int32_t* trace_view,
              const int32_t* trace_pos, double* trace_out);

/* The same deferred sweep over a list of entities only (local indices, list order): one segment of a segmented sweep. */
/* The live sweep (MVHDP_SWEEP_LIVE) as the sequential algorithm it is with one resident wavefront: see mvhdp_oracle.c.
 * order: entities in the product's longest-first order; rows / cell16: which form of the tree branch. */
int orc_sweep_live_seq(orc_model* o, uint32_t sweep_idx, uint64_t seed, int64_t doc_id_base, const double* p_in,
                       const int64_t* order, int64_t n_order, int nseg, int rows, int cell16, orc_stats* st);
int orc_row_sample_live(const int32_t* row, const float* cf, const float* smp, int K, int cell16, float u2f, float rootf, float mass0);
int orc_sweep_list(orc_model* o, uint32_t sweep_idx, uint64_t seed, int64_t doc_id_base,
                   const double* p, uint32_t flags, orc_stats* st, int32_t* delta_nwk, int32_t* delta_nk,
                   const int64_t* doc_list, int64_t n_list);

/* Apply externally reduced deltas (multi-rank tests): n_wk += d, n_k += d, then the topic
 * activation (UPD:263-270) decided by the caller (act_topic < 0: none). */
void orc_apply_delta(orc_model* o, const int32_t* delta_nwk, const int32_t* delta_nk,
                     int32_t act_topic, int32_t act_modality);

/* ---------------- SURVEY §8f next rows ---------------- */
double orc_log_gamma_stirling(double z);                      /* MALLET Dirichlet.logGammaStirling, used by PTM:3343-3441 */
double orc_mallet_digamma(double z);                          /* MALLET Dirichlet.digamma as compiled in 2.0.8 */
double orc_learn_symmetric_concentration(const int32_t* countHistogram, int n_count, const int32_t* observationLengths,
                                         int n_len, int numDimensions, double currentValue);   /* PTM:2327 */
void   orc_count_histogram(const orc_model* o, int m, int32_t* hist, int32_t len);             /* PTM:2295-2309 */
double orc_optimize_beta(orc_model* o, int m, int maxTypeCount, double* betaSum_out);          /* PTM:2293-2366 */
void   orc_optimize_p_sums(const orc_model* o, double* sums /*[M][M]*/);                        /* PTM:2706-2792 */
void   orc_model_log_likelihood(const orc_model* o, double* logLikelihood /*[M]*/);            /* PTM:3322-3452 */

/* ---------------- CPU baseline: the reference's thread topology ---------------- */
/* T threads -> nst = 3T/4 samplers over contiguous doc slices, nut = T/4
 * updaters over type%nut stripes, nst*nut unbounded queues, live (racy) reads,
 * incremental FTree.update (PTM:1036-1101, WRK:186-233, UPD:164-297).
 * Nondeterministic by design, like the reference.  Returns seconds of wall
 * time for `iters` iterations, or <0 on error. */
double orc_threaded_estimate(orc_model* o, int num_threads, int iters, uint64_t seed, orc_stats* st);

#ifdef __cplusplus
}
#endif
#endif
