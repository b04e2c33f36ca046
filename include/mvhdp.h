/*
 * mvhdp.h — C ABI of libmvhdp.so: the MI355X (gfx950) multi-view HDP
 * collapsed-Gibbs sweep that replaces the iteration body of
 * FastQMVWVParallelTopicModel.estimate() in hmetaxa/MVTopicModel.
 *
 * Reference aliases (under src/main/java/org/madgik/ of the reference):
 *   PTM = MVTopicModel/FastQMVWVParallelTopicModel.java
 *   WRK = MVTopicModel/FastQMVWVWorkerRunnable.java
 *   UPD = MVTopicModel/FastQMVWVUpdaterRunnable.java
 *   FT  = utils/FTree.java   QD = utils/FastQDelta.java
 *   MTA = utils/MixTopicModelTopicAssignment.java
 *
 * Conventions: extern "C"; plain pointers and sizes; every function returns
 * 0 (MVHDP_OK) or a negative mvhdp_status; no exceptions cross the boundary;
 * the caller owns every host buffer, the library copies.  A handle is bound to
 * one HIP device and is not thread-safe (estimate() is single-threaded,
 * PTM:1033).  The JNI stub that binds these entry points from Java is shown
 * in INTEGRATION.md.
 */
#ifndef MVHDP_H
#define MVHDP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVHDP_MAX_MODALITIES 8
#define MVHDP_MAX_TOPICS     2048
#define MVHDP_UNASSIGNED_TOPIC (-1)          /* PTM:63 */

typedef enum {
    MVHDP_OK = 0,
    MVHDP_ERR_INVALID_ARG   = -1,
    MVHDP_ERR_STATE         = -2,  /* call order violated (e.g. sweep before set_corpus) */
    MVHDP_ERR_HIP           = -3,  /* HIP runtime failure, see mvhdp_last_error */
    MVHDP_ERR_NO_DEVICE     = -4,  /* no gfx950 device: there is NO CPU fallback */
    MVHDP_ERR_NEGATIVE_COUNT= -5,  /* a count went below zero (the reference only logs it, UPD:202-215) */
    MVHDP_ERR_UNSUPPORTED   = -6
} mvhdp_status;

typedef struct mvhdp_ctx* mvhdp_handle;

/* Shape of the model: replaces the ctor arguments PTM:183 + numTypes PTM:413. */
typedef struct {
    int32_t num_topics;                          /* K, PTM:193 */
    int32_t num_modalities;                      /* M, PTM:189 */
    int32_t num_types[MVHDP_MAX_MODALITIES];     /* V_m = alphabet[m].size(), PTM:413; below 2^29 (two bits of a type id carry the row's class inside the kernels) */
    int32_t device;                              /* HIP device ordinal */
    int64_t doc_id_base;                         /* global id of local entity 0 (document shards, one per GPU) */
    uint32_t flags;                              /* reserved, 0 */
} mvhdp_config;

/* Hyper-parameters the sampler reads: PTM:79-83,130-131,95. */
typedef struct {
    const double* alpha;                         /* [M][K+1], index K = new-topic weight PTM:196 */
    double alpha_sum[MVHDP_MAX_MODALITIES];      /* PTM:80 */
    double beta[MVHDP_MAX_MODALITIES];           /* PTM:81 */
    double beta_sum[MVHDP_MAX_MODALITIES];       /* PTM:82, = beta*V_m PTM:420 */
    double gamma[MVHDP_MAX_MODALITIES];          /* PTM:83 */
    double p_a[MVHDP_MAX_MODALITIES][MVHDP_MAX_MODALITIES]; /* PTM:130 */
    double p_b[MVHDP_MAX_MODALITIES][MVHDP_MAX_MODALITIES]; /* PTM:131 */
    const uint8_t* inactive;                     /* [K] 1 = member of inActiveTopicIndex (PTM:95); NULL = none */
} mvhdp_hyper;

/* What one sweep reports: the reference's branch counters WRK:33-35 plus
 * bookkeeping. */
typedef struct {
    int64_t tokens;               /* tokens sampled */
    int64_t changed;              /* tokens whose topic changed = FastQDelta records, WRK:587 */
    int64_t new_mass_cnt;         /* WRK:523 */
    int64_t topic_doc_mass_cnt;   /* WRK:530 */
    int64_t word_ftree_mass_cnt;  /* WRK:533 */
    int64_t oov_skipped;          /* WRK:427-428 */
    int64_t aborted_docs;         /* Q11 (WRK:599-601): always 0 unless a mass is NaN */
    int64_t exact_fallbacks;      /* tokens that left the certified scan for the sequential sum */
    int32_t activated_topic;      /* UPD:263-270: topic leaving inActiveTopicIndex, -1 if none */
    int32_t activated_modality;   /*   and the view whose alpha[m][k] took alpha[m][K] */
    int64_t activation_key;       /* ordering key of that first delta, MVHDP_ACT_KEY(doc, view, pos, topic); INT64_MAX if none */
    double  sweep_kernel_ms;      /* device time of the sweep kernel alone (hipEvents on the handle's stream) */
    double  total_ms;             /* device time of the whole call: trees + view weights + sweep + apply */
    int32_t activations;          /* topics that left inActiveTopicIndex during this call: 0 or 1 for a deferred sweep; a SEGMENT_APPLY sweep or a
                                     LIVE sweep on stored trees activates one per segment border (up to its segment count); a LIVE sweep in its
                                     live-rows form gives birth chunk by chunk (any number: see MVHDP_SWEEP_LIVE); activated_topic is the FIRST of them
                                     -- a host pulls alpha / inActiveTopicIndex with mvhdp_get_alpha whenever activated_topic >= 0;
                                     0 with MVHDP_SWEEP_NO_APPLY / FROZEN (the caller activates) */
    int32_t reserved;
} mvhdp_sweep_stats;

/* Activation key: the FastQDelta that activates a topic first in (global entity, view, position) order wins (UPD:263-270
 * with a single updater).  One definition for the kernels, the host and any binding (multi-GPU: MIN-all-reduce the key). */
#define MVHDP_ACT_DOC_SHIFT   34
#define MVHDP_ACT_VIEW_SHIFT  31
#define MVHDP_ACT_POS_SHIFT   11
#define MVHDP_ACT_TOPIC_MASK  0x7ffLL
#define MVHDP_ACT_VIEW_MASK   0x7LL
#define MVHDP_ACT_KEY(doc, view, pos, topic) \
    (((int64_t)(doc) << MVHDP_ACT_DOC_SHIFT) | ((int64_t)(view) << MVHDP_ACT_VIEW_SHIFT) | ((int64_t)(pos) << MVHDP_ACT_POS_SHIFT) | (int64_t)(topic))
#define MVHDP_ACT_KEY_TOPIC(key) ((int32_t)((key) & MVHDP_ACT_TOPIC_MASK))
#define MVHDP_ACT_KEY_VIEW(key)  ((int32_t)(((key) >> MVHDP_ACT_VIEW_SHIFT) & MVHDP_ACT_VIEW_MASK))
#define MVHDP_ACT_KEY_NONE INT64_MAX

/* Optional debug outputs of a sweep (parity tests only; slows the kernel). */
typedef struct {
    /* per view: 4 doubles per token {newTopicMass, topicDocWordMass, tree root, sample}
     * (WRK:515-519); NULL entries are skipped. Host pointers. */
    double* tok_dbg[MVHDP_MAX_MODALITIES];
    /* full conditionals of selected tokens: (doc, view, pos) -> K+1 doubles,
     * normalised; slot K is the new-topic mass (SURVEY §8a "full conditional"). */
    int32_t n_trace;
    const int64_t* trace_doc;     /* local entity index */
    const int32_t* trace_view;
    const int32_t* trace_pos;
    double* trace_out;            /* [n_trace][K+1] host */
} mvhdp_debug;

/* sweep flags */
#define MVHDP_SWEEP_REUSE_TREES 0x1u  /* do not rebuild the F+trees from the counts first (PTM:1209 cadence is the host's) */
#define MVHDP_SWEEP_NO_APPLY    0x2u  /* leave the deltas unapplied (multi-GPU: all-reduce MVHDP_BUF_DELTA, then mvhdp_apply_delta).
                                       * The delta buffer may be written by the caller only between such a sweep and mvhdp_apply_delta. */
#define MVHDP_SWEEP_EXACT_CHAIN 0x4u  /* always use the sequential WRK:501-513 sum (test mode for the certified scan) */
#define MVHDP_SWEEP_GENERIC_KERNEL 0x8u /* force the LDS-resident kernel even when the register-resident one applies (test mode) */
#define MVHDP_SWEEP_FROZEN      0x10u /* the inferencer's call of the same worker (INF:211-294: nst=1, nut=0): sample against the
                                         stored trees and counts, queue no deltas (WRK:587), leave the model untouched */

/* The reference's own update discipline (UPD:197-218 applied WHILE the workers sample; PTM:84-87: racy reads by design):
 * the sweep's n_wk atomics go straight to the shared count array and every later token of the sweep reads them.  Not
 * reproducible run to run (like the reference); counts stay consistent with z.  The entities are cut into
 * MVHDP_SWEEP_LIVE_SEGMENTS(n) interleaved segments (n = 1..255, 0 = library default); at each segment boundary the
 * tokensPerTopic updates of the segment (privatised per workgroup: M*K hot words) land.
 * The tree branch (WRK:533-535) -- two forms, mvhdp_tuning.live_rows:
 *   live rows (default wherever every kernel of the sweep is register-resident; default 1 segment): no stored tree is sampled for a
 *     word of at most 65534 tokens; the branch draws from leaf_k = coef_k * (n_wk + beta) with n_wk the word's LIVE row, read when the
 *     token's turn comes, and coef_k = gamma alpha_k / (n_k + beta Sigma) of the segment start -- what the reference's updater achieves
 *     by refreshing the two touched leaves with every delta (UPD:242-260 -> FT:138-147).  Heavier words keep stored trees, rebuilt from
 *     the live counts over and over by a kernel that runs beside the samplers;
 *   stored trees (live_rows = 0, and wherever the generic kernel serves: default 4 segments): the F+trees are rebuilt from the live
 *     counts at every segment boundary (with REUSE_TREES: no rebuild at all, the host's PTM:1209 cadence).
 * With NO_APPLY the delta buffer receives (counts after - counts before) of this shard and counts are restored, so the
 * multi-GPU sequence all-reduce + mvhdp_apply_delta is the same as for a deferred sweep.  Not combinable with FROZEN.
 * (LIVE_SEGMENTS without LIVE cuts a deferred sweep into the same segments: same integers as one segment.)
 * Topic activation (UPD:263-270): the first delta of a SEGMENT that lands on an inactive topic activates it at the segment's
 * end -- alpha[m][k] takes alpha[m][K], the topic leaves inActiveTopicIndex, the next segment's new-topic draws go to the
 * next inactive index (WRK:523-526) -- so one sweep can give birth to up to n topics (mvhdp_sweep_stats.activations); the
 * reference's updater does it delta by delta (on C5 its 100 inactive topics are all born within the first sweep).  A live sweep in
 * its live-rows form does it chunk by chunk: a new-topic draw takes the first inactive topic no delta has reached yet, a chunk whose
 * deltas reach it moves the samplers on to the next, and the segment's end activates every topic that was reached, in index order
 * (mvhdp_sweep_stats.activations can exceed the segment count).  With NO_APPLY nothing is activated here (the caller reduces the
 * key first). */
#define MVHDP_SWEEP_LIVE        0x20u
#define MVHDP_SWEEP_LIVE_SEGMENTS(n) (((uint32_t)(n) & 0xffu) << 16)

/* A deferred sweep cut into MVHDP_SWEEP_LIVE_SEGMENTS(n) segments WITH the updater catching up in between: every
 * segment is a snapshot sweep over its entities (trees rebuilt from the current counts, deltas collected), then its deltas
 * are applied before the next segment starts.  Bit-reproducible like the plain deferred sweep (the oracle follows it
 * segment by segment), and statistically between the deferred and the live sweep: a token sees counts that are at most
 * one segment old.  Single handle only (not combinable with NO_APPLY, LIVE or FROZEN); a topic is activated at the end of
 * the segment whose delta reached it first (as for LIVE above).  The segments are the interleaved ones of the live sweep: positions s, s+n, s+2n, ... of the
 * entities ordered by decreasing token count (ties by entity index). */
#define MVHDP_SWEEP_SEGMENT_APPLY 0x40u

/* With MVHDP_SWEEP_SEGMENT_APPLY: the updater runs BESIDE the samplers, as UPD:164-297 runs beside WRK:186-233 -- the deltas of
 * segment s are applied to the counts while segment s+1 is being sampled, so segment s+2 is the first to see them: every token of
 * segment s samples against the counts after segment s-2 (segments 0 and 1: the sweep-start counts).  The F+trees are built once, at
 * the start of the sweep, and serve every segment: a DEVIATION from the reference, whose updater refreshes the two touched leaves with
 * every delta (UPD:242-260) so that its trees follow the counts -- here a token's tree-branch mass and its count-based branch come from
 * different model states (plain SEGMENT_APPLY rebuilds the trees at every segment border).  Still a deterministic chain,
 * followed by the oracle segment by segment (tests/test_gpu_segmented.py); and without the per-segment stall of plain SEGMENT_APPLY:
 * two segments are in flight, no kernel boundary idles the chip (the counts and their mirror are kept twice).  Not with inactive
 * topics (the activation of UPD:263-270 needs the host between segments): MVHDP_ERR_UNSUPPORTED. */
#define MVHDP_SWEEP_SEGMENT_OVERLAP 0x80u

/* Only segment s (0-based) of the MVHDP_SWEEP_LIVE_SEGMENTS(n) interleaved segments is swept: the entities at positions s, s+n,
 * s+2n, ... of the longest-first order; the statistics cover those entities.  With it a HOST drives a segmented sweep and can put
 * anything between two segments -- an all-reduce over document shards, a look at the counts (tests/test_gpu_full_size.py checks
 * every segment of a full-size segmented sweep against the oracle this way).  n calls with s = 0..n-1 of a deferred sweep (each
 * applying its deltas) give the integers of one MVHDP_SWEEP_SEGMENT_APPLY call with n segments.  Not with LIVE or SEGMENT_APPLY. */
#define MVHDP_SWEEP_ONLY_SEGMENT(s) ((((uint32_t)(s) + 1u) & 0xffu) << 24)

/* device buffers a host may hand to a collective (RCCL through torch.distributed or directly) */
typedef enum {
    MVHDP_BUF_COUNTS = 0,  /* int32 [sumV*K + M*K]: n_wk rows of every view, then n_k */
    MVHDP_BUF_DELTA  = 1   /* int32 [sumV*K + M*K]: the last sweep's deltas, same layout */
} mvhdp_buffer;

/* ---- lifetime ---- */
int mvhdp_create(const mvhdp_config* cfg, mvhdp_handle* out);   /* replaces new FastQMVWV…TopicModel PTM:183 + initSpace PTM:575 */
int mvhdp_destroy(mvhdp_handle h);
const char* mvhdp_last_error(mvhdp_handle h);                    /* h may be NULL: last create error */
const char* mvhdp_version(void);

/* ---- corpus and assignments: MTA / MALLET FeatureSequence + LabelSequence flattened to CSR ---- */
/* One call per view m; D entities in `data` order (PTM:443-455); an entity
 * without view m (Assignments[m]==null, MTA:19) is an empty span. */
int mvhdp_set_corpus(mvhdp_handle h, int32_t m, int64_t num_docs,
                     const int64_t* doc_off /*[D+1]*/, const int32_t* tokens /*[N_m]*/);
int mvhdp_set_assignments(mvhdp_handle h, int32_t m, const int32_t* z /*[N_m]*/);  /* topicSequence.getFeatures() PTM:481 */
/* Which entities HAVE view m (Assignments[m] != null, MTA:19): present[d] = 1 also for an instance with an empty FeatureSequence,
 * which the CSR alone cannot tell from a missing one.  NULL (the default) = present iff the span is non-empty.  The sweep does not
 * care (WRK:341,403 treat null and length 0 alike); the statistics do: modelLogLikelihood's two phantom tokens of topic 0 and its
 * modalityCnt (PTM:3348-3373: the backing array of an empty LabelSequence has length 2), totalDocsPerModality and
 * docLengthCounts[0] (PTM:620-651), and the carry-over of printDocumentTopics (PTM:2873-2886: a present empty view scores with
 * zeros, a missing one with the previous holder's counts).  Call after mvhdp_set_corpus of that view. */
int mvhdp_set_view_presence(mvhdp_handle h, int32_t m, const uint8_t* present /*[D] or NULL*/);
int mvhdp_get_assignments(mvhdp_handle h, int32_t m, int32_t* z /*[N_m]*/);

/* ---- model state ---- */
int mvhdp_set_hyper(mvhdp_handle h, const mvhdp_hyper* hy);
int mvhdp_get_alpha(mvhdp_handle h, double* alpha /*[M][K+1]*/, uint8_t* inactive /*[K]*/); /* after a topic activation UPD:263-270 */
int mvhdp_build_counts(mvhdp_handle h);                          /* buildInitialTypeTopicCounts PTM:600-652 */
int mvhdp_build_trees(mvhdp_handle h);                           /* buildFTrees PTM:2660-2696 */
int mvhdp_build_inference_trees(mvhdp_handle h);                 /* FastQMVWVTopicInferencer.initInferencer INF:557-586: leaves p_wt, no gamma*alpha */
/* INF:169-199: z = trees[m][type].sample(u) for in-vocabulary tokens, 0 otherwise; u from the token stream with sweep 0xFFFFFFFF */
int mvhdp_init_assignments_from_trees(mvhdp_handle h, uint64_t seed);
int mvhdp_get_counts(mvhdp_handle h, int32_t m, int32_t* n_wk /*[V_m][K] or NULL*/, int32_t* n_k /*[K] or NULL*/);
int mvhdp_set_counts(mvhdp_handle h, int32_t m, const int32_t* n_wk, const int32_t* n_k);
int mvhdp_get_tree(mvhdp_handle h, int32_t m, int32_t type, double* tree /*[2K], FTree.tree FT:21*/);
/* topicDocCounts[m][k][c] (c < hist_len) and docLengthCounts[m][len] (PTM:107-108,
 * 620-651), recomputed from z instead of the updater's incremental bookkeeping
 * UPD:220-232. Either output may be NULL. */
int mvhdp_get_doc_topic_hist(mvhdp_handle h, int32_t m, int32_t* hist /*[K][hist_len]*/, int32_t hist_len,
                             int32_t* doc_len_counts /*[len_len]*/, int32_t len_len);

/* ---- the steps either side of the sweep: statistics the host's optimize* and logging need ---- */
/* countHistogram of optimizeBeta PTM:2295-2309: hist[c] = number of (type, topic) pairs of view m holding count c (c >= 1, c < len). */
int mvhdp_get_count_histogram(mvhdp_handle h, int32_t m, int32_t* hist, int32_t len);
/* optimizeP PTM:2706-2792: sums[m][i] = sum over entities, in entity order, of pDistr_Mean[m][i][doc]. */
int mvhdp_view_overlap_sums(mvhdp_handle h, double* sums /*[M][M]*/);
/* modelLogLikelihood PTM:3322-3452, one value per view. */
int mvhdp_model_log_likelihood(mvhdp_handle h, double* log_likelihood /*[M]*/);
/* optimizeGamma PTM:2415-2433, the document level (Teh et al. 2006): over the entities that have view m, of length j,
 * qs = sum Bernoulli(j/(j+gamma_m)) and qw = sum log Beta(gamma_m+1, j).  The reference draws them sequentially from a
 * stream that cannot be seeded (RandomSamplers over ThreadLocalRandom, PTM:236), ten rounds per view; here every entity
 * draws from its own counter-based stream (seed, global entity id, view, round): the same random variables in distribution,
 * reproducible, shard-independent.  A host keeps its closed-form updates and calls this for the two sums. */
int mvhdp_gamma_doc_statistics(mvhdp_handle h, int32_t m, double gamma_m, uint64_t seed, uint32_t round, double* qs, double* qw);
/* optimizeDP PTM:2454-2488, the view-table simulation over topicDocCounts[m] (hist [K][hist_len] as mvhdp_get_doc_topic_hist or
 * mvhdp_group_doc_topic_hist returns it: host memory).  For every cell (topic t, count i) that holds entities: i == 1 adds them; i > 1
 * adds them times ONE draw of the number of tables a CRP(conc[t]) makes of i items (conc[t] = gamma[m] * alpha[m][t], PTM:2471) -- the
 * Antoniak distribution, drawn as the sum of the i - 1 Bernoulli(conc / (conc + l)) table openings from a counter-based stream
 * (seed, round, view, topic, count).  The reference draws it through a table of Stirling numbers from a stream that cannot be seeded
 * and scales the CACHED table row in place on every call (Samplers.java:1086-1110); a host that wants that sequence keeps its own loop
 * (the default of the host classes), one that wants the statistic calls this.  mk[t]: the sum over the cells; active[t]: 1 iff a cell
 * with i >= 1 holds an entity (PTM:2461,2480: the topic leaves inActiveTopicIndex).  The root level (PTM:2491-2517: K * M draws) stays
 * with the host. */
int mvhdp_dp_table_statistics(mvhdp_handle h, int32_t m, const int32_t* hist /*[K][hist_len]*/, int32_t hist_len, const double* conc /*[K]*/,
                              uint64_t seed, uint32_t round, double* mk /*[K]*/, uint8_t* active /*[K]*/);
/* n independent draws of the same kind -- optimizeDP's root level PTM:2491-2517 asks for K * M of them, randAntoniak(gammaRoot,
 * ceil(mk[m][t])): tables[j] = the number of tables a CRP(conc[j]) makes of items[j] items; items <= 0: 0, 1: 1, more than 20000
 * (the reference's MAXSTIRLING, Samplers.java:1024: its call throws there and PTM:2507-2509 falls back to one table): 1. */
int mvhdp_antoniak_draws(mvhdp_handle h, int32_t n, const int32_t* items /*[n]*/, const double* conc /*[n]*/, uint64_t seed, uint32_t round, int32_t* tables /*[n]*/);
/* printDocumentTopics PTM:2871-2899 (and the inferencer's INF:383-411): topic proportions of entities [d0, d1),
 * out[(d-d0)*K + k] = sum_m w[m]*(n_dk[m][k] + gamma[m]*alpha[m][k])/(len[m] + gamma[m]*alphaSum[m]) / sum_m w[m],
 * w[m] = (m == 0 ? 1 : discrWeightPerModality[m]) * pMean[0][m].  As in the reference, whose topicCounts[m] / docLen[m]
 * are refreshed only when the entity has view m (PTM:2873-2886), an entity WITHOUT view m is scored with n_dk[m] and
 * len[m] of the last earlier entity of this handle that had it (zeros before the first). */
int mvhdp_doc_topic_proportions(mvhdp_handle h, const double* view_weights /*[M]*/, int64_t d0, int64_t d1, double* out /*[d1-d0][K]*/);

/* ---- the hot path ---- */
/* One Gibbs sweep over every entity: replaces "submit updaters + submit
 * workers + barrier.await()" PTM:1213-1239, i.e. WRK:186-233 x nst threads and
 * UPD:164-297 x nut threads.  sweep_idx and seed select the counter-based RNG
 * stream that stands in for ThreadLocalRandom (WRK:517,534).  p_override:
 * host [D][M][M] view weights drawn as WRK:327-337, or NULL to draw them on
 * the device (same nextBeta algorithm over a Philox stream).  dbg may be NULL.
 * Synchronous. */
int mvhdp_sweep(mvhdp_handle h, uint32_t sweep_idx, uint64_t seed, uint32_t flags,
                const double* p_override, const mvhdp_debug* dbg, mvhdp_sweep_stats* stats);
/* n sweeps, indices first_idx .. first_idx+n-1, enqueued back to back: the iteration loop PTM:1146-1239 without a host
 * round trip per iteration (one plan for the batch, statistics collected on the device, one synchronisation at the end).
 * Same integers as n calls of mvhdp_sweep with p_override = dbg = NULL.  stats: [n] or NULL (total_ms is the batch's time / n).
 * Falls back to n single calls where every sweep needs the host: a model with inactive topics (activation UPD:263-270),
 * MVHDP_SWEEP_NO_APPLY. */
int mvhdp_sweep_many(mvhdp_handle h, uint32_t first_idx, int32_t n, uint64_t seed, uint32_t flags, mvhdp_sweep_stats* stats /*[n]*/);
/* n_wk += delta, n_k += delta, delta = 0; also performs the topic activation
 * recorded by the sweep when (topic,modality) >= 0 (multi-GPU: the host passes
 * the winner of the min-reduction over activation_key). */
int mvhdp_apply_delta(mvhdp_handle h, int32_t activated_topic, int32_t activated_modality);
/* The same update as a stream-ordered pipeline, for document shards on several GPUs: the all-reduce of the delta buffer
 * can be issued in row-range chunks and each chunk's rows applied AND their F+trees rebuilt (buildFTrees PTM:2660-2696
 * from the updated counts) while the next chunk is still on the wire.  n_wk rows are numbered over all views (view m
 * starts at num_types[0] + .. + num_types[m-1]); the tokensPerTopic part sits behind the last row in both buffers.
 *   mvhdp_apply_delta_begin    applies the tokensPerTopic part (all-reduce it first: every tree needs all of it)
 *   mvhdp_apply_delta_rows     rows [row_begin, row_end): counts += delta, delta = 0, trees of those rows rebuilt; no wait
 *   mvhdp_apply_delta_end      every row must have been applied exactly once; topic activation as mvhdp_apply_delta;
 *                              one synchronisation; negative counts reported here
 * Afterwards mvhdp_trees_current() is 1 (unless a topic was activated) and the next sweep may pass REUSE_TREES. */
int mvhdp_apply_delta_begin(mvhdp_handle h);
int mvhdp_apply_delta_rows(mvhdp_handle h, int64_t row_begin, int64_t row_end);
int mvhdp_apply_delta_end(mvhdp_handle h, int32_t activated_topic, int32_t activated_modality);
int mvhdp_trees_current(mvhdp_handle h);    /* 1: the F+trees match the counts and hyper-parameters, 0: not, < 0: error */
/* the per-document view weights used by the last sweep */
int mvhdp_get_view_weights(mvhdp_handle h, double* p /*[D][M][M]*/);

/* ---- tuning: the sweep's own choices, pinned or carried over ----
 * None of these changes a result: they decide which kernel variant visits an entity and when a word tree is walked, never
 * what is sampled.  The library reads the environment ONCE, in mvhdp_create (MVHDP_FORCE_RMAX, MVHDP_NARROW, MVHDP_LIVE16,
 * MVHDP_WALK_THETA, MVHDP_SINGLE_STREAM, MVHDP_PRIMARY_MIN_SHARE, MVHDP_LIVE_ROWS, MVHDP_DEBUG: diagnostics); a host uses this block.
 * learnt_walk_step / tree_branch_share are what the walk-threshold search has found: read them from one handle
 * (mvhdp_get_tuning) and hand them to another -- a document shard, a resumed chain -- and it does not search again. */
typedef struct {
    int32_t force_primary;                       /* 0: the library chooses; 1,2,4,8,16: primary register variant; 32: generic kernel only */
    int32_t narrow;                              /* -1: 16-bit mirror of n_wk wherever legal (default); 0: never; 1: for the 1-round kernel variant only */
    int32_t walk_fixed;                          /* 1: walk_theta[] as given, no search */
    int32_t single_stream;                       /* 1: all class kernels on the handle's stream (diagnostics) */
    int32_t live16;                              /* MVHDP_SWEEP_LIVE keeps the light n_wk rows current in the 16-bit mirror (half-width gathers): -1 where K >= 256 (default), 0 never, 1 always */
    int32_t single_wave;                         /* diagnostics, 0 by default.  1: every sweep kernel runs as ONE wavefront (one block of 64 threads), the class kernels one
                                                    after another, and a live sweep waits for its chunk-end atomics and drops its L1 before it goes on: a live sweep then IS
                                                    the sequential algorithm (every token sees every earlier update of the same segment), so the 32-bit form and the 16-bit
                                                    mirror form must give the same integers (tests/test_gpu_live.py) */
    double  walk_theta[MVHDP_MAX_MODALITIES];    /* with walk_fixed: walk a token's word tree up front iff u1 >= walk_theta[view] */
    double  primary_min_share;                   /* narrowest kernel class holding this share of the tokens gets its own kernel (0 = default 0.10) */
    int32_t learnt_walk_step[4];                 /* searched threshold in 1/20 steps per kernel flavour: [0] 1-round variant on the 16-bit mirror, [1] 1-round
                                                    variant on 32-bit rows, [2] the wider variants; -1: none yet (the library's default); [3] reserved */
    double  tree_branch_share[MVHDP_MAX_MODALITIES]; /* tree-branch (WRK:533) share of each view's tokens in the last measured sweep; < 0: not measured */
    int32_t live_overlap;                        /* MVHDP_SWEEP_LIVE with several segments: -1 (default) / 1: the next segment's trees are rebuilt (from the live counts)
                                                    and its kernels launched when the current segment is nearly through, so that no segment border idles the chip;
                                                    0: one segment after the other (the round-3 form) */
    int32_t live_rows;                           /* MVHDP_SWEEP_LIVE: -1 (default) / 1: the tree branch of a token (WRK:533-535) samples from the word's LIVE count row
                                                    (see MVHDP_SWEEP_LIVE) wherever every kernel of the sweep is register-resident; one segment per sweep by default;
                                                    0: stored trees rebuilt at every segment border, four segments by default (the round-4 form) */
} mvhdp_tuning;
int mvhdp_get_tuning(mvhdp_handle h, mvhdp_tuning* t);
int mvhdp_set_tuning(mvhdp_handle h, const mvhdp_tuning* t);

/* The sweep's planner and its walk-threshold search are pure functions (mvtopicmodel_amd/csrc/mvhdp_plan.h); these two run
 * them WITHOUT a device or a handle, on recorded inputs (tests/test_plan.py). */
typedef struct {
    int32_t num_topics, num_modalities;
    int64_t num_entities;
    int64_t max_entity_tokens;                   /* longest entity, all views together */
    int64_t entities_longer_than[5];             /* entities with more than 64, 128, 256, 512, 1024 tokens */
    uint64_t tokens_by_list_rounds[17];          /* tokens of the entities whose topic list needs 1..16, >16 rounds of 64 slots */
    uint64_t entities_by_class[8];               /* entities per kernel class 0..5 (64 << c slots; 5: generic kernel); [6]: list size not known */
    uint32_t flags;                              /* MVHDP_SWEEP_* */
    int32_t debug, batch, trees_current;
    int32_t num_cus;                             /* 0 = 256 */
    int32_t kernel_registers[6][3];              /* VGPRs of each kernel class: plain, walk flavour, debug build */
    int32_t inactive_topics;                     /* 1: inActiveTopicIndex is not empty (a live sweep then takes more segments: a topic is born per border) */
} mvhdp_plan_input;
typedef struct {
    int32_t status;                              /* what mvhdp_sweep would return for these flags (MVHDP_OK or an error) */
    int32_t segments, primary_class, register_resident, need_full_trees, dominant_class;
    int64_t routed_prefix;                       /* entities of the longest-first order that go through the route pass */
    int32_t class_used[6], class_map[6], class_stream[6], class_grid[6], class_walk[6], class_narrow[6], class_register_resident[6];
    int64_t class_lds_bytes[6];
    double  class_theta0[6];                     /* walk threshold of view 0 for that class's kernel */
    int32_t delta16, live_rows;                  /* delta16 1: the sweep keeps the n_wk deltas of rows with at most 32767 tokens in 16-bit cells (half the table the
                                                    chunk-end atomics land in), plain deferred sweeps only; live_rows 1: a live sweep in its live-rows form
                                                    (mvhdp_tuning.live_rows) */
} mvhdp_plan_output;
int mvhdp_plan_probe(const mvhdp_plan_input* in, const mvhdp_tuning* tuning /* or NULL */, mvhdp_plan_output* out);
/* The search alone: a kernel whose time per token at threshold step i is ns_by_step[i] (i = 0..20); steps_out[k] = the
 * threshold step proposed for sweep k. */
int mvhdp_tuner_probe(int32_t num_modalities, const double* tree_branch_share /*[M]*/, const double* u1_hist /*[20] or NULL*/,
                      const double* ns_by_step /*[21]*/, int32_t n_sweeps, int32_t group, int32_t* steps_out /*[n_sweeps]*/);

/* ---- document shards on several GPUs: the exchange step inside the library ----
 * What the reference keeps inside its own process -- the nst x nut queue mesh between sampler and updater threads and the
 * barrier that ends an iteration (PTM:1042-1049, PTM:1232) -- for samplers that are GPUs: every member handle holds a
 * contiguous range of entities (mvhdp_config.doc_id_base = global id of its first entity) and a full replica of the model.
 * One mvhdp_group_sweep = every member samples its entities against the same snapshot; the int32 deltas are summed over all
 * members (on the device where members share a GPU, by RCCL all-reduce over xGMI between GPUs: the only collective of the
 * path), pipelined in row ranges with the update and F+tree rebuild of the rows that have arrived; with inactive topics the
 * activation key (MVHDP_ACT_KEY) is MIN-reduced so that every replica activates the same topic (UPD:263-270).  Results are
 * bit-identical to one handle holding every entity.  RCCL is opened at run time (a copy already mapped into the process, else the
 * file MVHDP_RCCL_LIB names, else librccl.so.1; MVHDP_RCCL_LIB_FIRST=1 tries the named file before a mapped copy -- tests): a single-GPU host never loads it.  A handle belongs to at most one group; destroy
 * the group before its members (a group call on a group whose member is gone returns MVHDP_ERR_STATE).  Group calls leave the
 * caller's current HIP device as they found it. */
typedef struct mvhdp_group_ctx* mvhdp_group;
#define MVHDP_UNIQUE_ID_BYTES 128
typedef struct {
    int32_t local_members;        /* handles of this process in the group */
    int32_t local_devices;        /* distinct GPUs among them = RCCL ranks of this process */
    int32_t ranks, first_rank;    /* RCCL ranks over all processes; rank of this process's first device */
    int32_t rccl;                 /* 1: the collective is RCCL; 0: one device and no RCCL installed (device-side sum only) */
    int32_t rccl_version;
    int32_t exchange_chunks;      /* row ranges per exchange (default 4) */
    int32_t exchange_packed;      /* 1: the n_wk deltas of the rows whose type holds at most 32767 tokens travel two to a 32-bit word (a delta of one sweep,
                                     summed over all ranks, cannot leave 16 bits there): agreed on by every rank behind the first completed sweep after a
                                     (re)count, 0 until then */
    double  last_exchange_ms;     /* device time of the last sweep's exchange on this process's first device: collectives + updates + tree rebuilds */
    int64_t last_exchange_bytes;  /* bytes this process handed to the collective in the last sweep, per device (96 MB at C4 at full width, 49 MB packed) */
} mvhdp_group_info;
/* one process drives n GPUs (the Java host of INTEGRATION.md): ncclCommInitAll over the members' devices */
int mvhdp_group_create(int32_t n, const mvhdp_handle* members, mvhdp_group* out);
/* one process per GPU: rank 0 obtains an id, the host's launcher hands it to every rank, every rank calls create_rank (collective) */
int mvhdp_group_unique_id(uint8_t* id /*[MVHDP_UNIQUE_ID_BYTES]*/);
int mvhdp_group_create_rank(mvhdp_handle member, const uint8_t* id /*[MVHDP_UNIQUE_ID_BYTES]*/, int32_t rank, int32_t nranks, mvhdp_group* out);
int mvhdp_group_destroy(mvhdp_group g);
const char* mvhdp_group_last_error(mvhdp_group g);           /* g may be NULL: last create error */
int mvhdp_group_get_info(mvhdp_group g, mvhdp_group_info* info);
int mvhdp_group_set_exchange_chunks(mvhdp_group g, int32_t chunks /* 1..64 */);
/* buildInitialTypeTopicCounts PTM:600-652 over all shards: every member counts its entities, the counts are summed over the group */
int mvhdp_group_build_counts(mvhdp_group g);
/* flags: MVHDP_SWEEP_LIVE (+ LIVE_SEGMENTS), SEGMENT_APPLY (+ LIVE_SEGMENTS), EXACT_CHAIN, GENERIC_KERNEL; stats: one per local member, or NULL.
 * Failure (one process per GPU): every rank enters the same collectives whatever happens locally; a rank whose sweep failed contributes
 * zero deltas and raises a status word that is reduced with the tokensPerTopic part, so ALL ranks return an error from the same call
 * (the failing rank its own, the others MVHDP_ERR_STATE) and none waits inside a collective for a rank that has given up.  After such
 * an error call mvhdp_group_build_counts on every rank (a recount from the assignments) before the next sweep. */
int mvhdp_group_sweep(mvhdp_group g, uint32_t sweep_idx, uint64_t seed, uint32_t flags, mvhdp_sweep_stats* stats);
/* The collective off the critical path, for LIVE sweeps of a group (opt-in).  A live sweep across shards is AD-LDA already: a replica
 * is live for its own entities and stale for the others'.  With this flag sweep t keeps its own changes in place, puts its deltas on
 * the wire at once -- the all-reduce runs on a stream of its own BESIDE sweep t+1 -- and the other shards' share of them is added when
 * sweep t+1 has been sampled: the step costs max(sampling, collective) instead of their sum, the chain sees the other shards' tokens
 * one to two sweeps late instead of zero to one.  Between such sweeps the replicas differ (each lacks the others' last sweep):
 * mvhdp_group_drain lands what is in flight and makes every replica the global model (the group's statistics, build_counts and any
 * sweep without the flag drain by themselves).  A failure shows one sweep late, on every rank together.  Not with inactive topics. */
#define MVHDP_SWEEP_ASYNC_EXCHANGE 0x100u
int mvhdp_group_drain(mvhdp_group g);
/* A host whose rank cannot go on calls this before its next mvhdp_group_sweep: that sweep samples nothing here and fails on every rank
 * together (see above) instead of leaving the others inside an all-reduce. */
int mvhdp_group_abort(mvhdp_group g);

/* ---- the steps either side of the sweep for a sharded model: what estimate() does every optimizeInterval and every tenth iteration
 * (PTM:1173-1210 -> optimizeP PTM:2698-2819, optimizeDP PTM:2440-2591, optimizeGamma PTM:2369-2438, optimizeBeta PTM:2288-2367;
 * PTM:1296-1320 -> modelLogLikelihood PTM:3322-3452).  Statistics over the replicated counts are any member's; statistics over the
 * entities are put together from the members -- in one process in ascending doc_id_base with the running sums carried from member to
 * member (the single handle's additions in the single handle's order: bit-identical), across processes by adding every rank's partial
 * result in rank order (equal to rounding; collective: every rank calls).  Arguments as for the single-handle functions. ---- */
int mvhdp_group_set_hyper(mvhdp_group g, const mvhdp_hyper* hy);          /* every local member (replicated: every rank passes the same values) */
int mvhdp_group_log_likelihood(mvhdp_group g, double* log_likelihood /*[M]*/);
int mvhdp_group_doc_topic_hist(mvhdp_group g, int32_t m, int32_t* hist /*[K][hist_len]*/, int32_t hist_len,
                               int32_t* doc_len_counts /*[len_len]*/, int32_t len_len);
int mvhdp_group_count_histogram(mvhdp_group g, int32_t m, int32_t* hist, int32_t len);
int mvhdp_group_view_overlap_sums(mvhdp_group g, double* sums /*[M][M]*/);
int mvhdp_group_gamma_doc_statistics(mvhdp_group g, int32_t m, double gamma_m, uint64_t seed, uint32_t round, double* qs, double* qw);

/* ---- interop for collectives and stream sharing ---- */
int mvhdp_device_buffer(mvhdp_handle h, mvhdp_buffer which, void** dev_ptr, size_t* bytes);
/* The caller has written MVHDP_BUF_COUNTS through the device pointer (e.g. the all-reduce of the shards' initial
 * counts): the counts are now valid, any F+trees built from the old values are not. */
int mvhdp_counts_written(mvhdp_handle h);
int mvhdp_set_stream(mvhdp_handle h, void* hip_stream /* hipStream_t, NULL = library's own */);
int mvhdp_synchronize(mvhdp_handle h);

#ifdef __cplusplus
}
#endif
#endif /* MVHDP_H */
