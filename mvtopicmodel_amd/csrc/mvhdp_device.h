// mvhdp_device.h — kernel argument block and launch prototypes shared by the
// HIP kernels (mvhdp_kernels.hip) and the C-ABI host side (mvhdp_api.hip).
// gfx950 only: 64-lane wavefronts are assumed everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MVHDP_MAXM 8
#define MVHDP_WALK_BINS 20                      /* walk thresholds are multiples of 1/MVHDP_WALK_BINS */
#define MVHDP_VIEW_STATS (2 + MVHDP_WALK_BINS)

// Everything a kernel needs, passed by value (lands in SGPRs / constant memory).
struct MvModel {
    int32_t K, M;
    int32_t V[MVHDP_MAXM];
    int64_t rowbase[MVHDP_MAXM + 1];   // cumulative V: n_wk row of (m,w) = rowbase[m]+w
    int64_t D;                         // local entities
    int64_t doc_id_base;               // global id of entity 0
    const int64_t* doc_off[MVHDP_MAXM];
    const int32_t* tok[MVHDP_MAXM];
    int32_t* z[MVHDP_MAXM];
    // [D] per view, or nullptr: 1 = the entity HAS this view (Assignments[m] != null, MTA:19) even if it holds no token.  nullptr (the
    // default) = a view is present iff its span is non-empty.  Only the statistics either side of the sweep can tell the difference
    // (the worker treats null and length 0 alike, WRK:341,403): modelLogLikelihood's phantom tokens and modalityCnt (PTM:3348-3373),
    // totalDocsPerModality / docLengthCounts[0] (PTM:620-651), the carry-over of printDocumentTopics (PTM:2873-2886).
    const uint8_t* present[MVHDP_MAXM];
    // [D] distinct topics of each entity's current assignments over all views = the size of its topic list (WRK:376-391) at
    // the NEXT visit.  Every sweep kernel writes it when it leaves an entity (slot_count_kernel after assignments came from
    // the host), so the next sweep routes every entity to the narrowest kernel variant that holds it without measuring
    // anything (route_kernel); 0xFFFF = not known (an abandoned entity, Q11): the host recounts before the next sweep.
    uint16_t* nslots;
    // model: counts = [sumV*K n_wk | M*K n_k], delta same layout
    int32_t* counts;
    uint16_t* delta16;                 // [sumV*K] the sweep's n_wk deltas of the rows that cannot overflow 16 bits (MVHDP_ROW_BIG clear), biased by 0x8000,
                                       //   two cells a word: the chunk-end atomics of a deferred sweep land in HALF the table (SweepLaunch::delta16)
    uint16_t* counts16;                // [sumV*K] min(n_wk, 65535): written row by row whenever the row's tree is built (build_trees_kernel),
                                       //   so it is the sweep-start n_wk of every sweep that starts with the trees; read by the NARROW kernel flavour
    uint8_t* heavy;                    // [sumV] MVHDP_ROW_HEAVY (1): the row's type holds more than 65534 tokens in all: its mirror cells are all 65535 and its
                                       //   counts live in the 32-bit table only.  A LIGHT row's cells can never reach 65535, so a live sweep may keep
                                       //   them current IN the mirror (two cells per 32-bit word, +-1 / +-65536 atomics, no carry) -- SweepLaunch::live16
    int32_t* delta;
    double* trees;                     // [sumV][2K]  FTree.tree (FT:21)
    double* root;                      // [sumV]      tree[1]
    // MVHDP_SWEEP_LIVE in its "live rows" form (SweepLaunch::live_rows): the tree branch of a token (WRK:533-535) samples from the word's
    // LIVE count row instead of a stored tree -- leaf_k = coef[m][k] * (n_wk + beta_m), coef[m][k] = (float)(gamma_m alpha_mk / (n_k + betaSum_m)),
    // 0 for an inactive topic (PTM:2670-2678 with the segment-start tokensPerTopic) -- and `root` above holds sum_k leaf_k per type, built
    // at the segment start and kept current by one fp64 atomic per FastQDelta (coef[new] - coef[old]): what UPD:242-260 does to the
    // two touched leaves and the root path, without a tree.
    float* coef;                       // [M][Kp] (Kp = K rounded up to a multiple of 8, the pad zero), then [M][K] the running sums over k of
                                       //   coef[m][k] * beta_m (the smoothing part of every leaf of the view)
    float* mass0;                      // [sumV] live-rows form, rows of TWO register batches (K above 512 cells of the mirror / 256 of the 32-bit table):
                                       //   sum over the FIRST batch's topics of coef_k * n_wk at the segment start (live_rows_prepare_kernel) -- a target
                                       //   beyond S_m + mass0 starts in the second batch, so ONE speculative row load serves either half (row_sample_live)
    // Descent table: what FTree.sample (FT:118-132) reads -- tree[1] and the left-child sums tree[2i] of the
    // internal nodes i -- regrouped so that three consecutive levels of one path share a 64-byte block
    // (8 doubles: L[b]; L[2b], L[2b+1]; L[4b..4b+3]; spare, = tree[1] in block 0; L[i] = tree[2i]).
    // A descent reads ceil(levels/3) sectors instead of one per level.  The first block holds dt_f = 1..3
    // levels so that every later block is full; blocks rooted at depth dt_depth[j] start at block dt_base[j].
    double* dtab;                      // [sumV][dt_nblk][8]
    int32_t dt_nblk, dt_nbd, dt_f;
    int32_t dt_base[6], dt_depth[6];
    const double* alpha;               // [M][K+1]
    const uint8_t* inactive;           // [K]
    double alpha_sum[MVHDP_MAXM], beta[MVHDP_MAXM], beta_sum[MVHDP_MAXM], gamma[MVHDP_MAXM];
    double p_a[MVHDP_MAXM][MVHDP_MAXM], p_b[MVHDP_MAXM][MVHDP_MAXM];
    double* p;                         // [D][M][M] view weights (nullptr when M==1)
    int32_t first_inactive;            // inActiveTopicIndex.first() or -1
};

// SweepLaunch::flags, internal: a live sweep waits for its chunk-end atomics and invalidates the CU's L1 before it goes on (with one
// resident wave the sweep is then the sequential algorithm: mvhdp_tuning.single_wave)
#define MVHDP_SL_STRICT_LIVE 0x8000u
// MvModel::heavy values: 0 = light and small, MVHDP_ROW_HEAVY = more than 65534 tokens (not in the mirror), MVHDP_ROW_BIG = light, but
// more than 32767 tokens (mirror yes, 16-bit deltas no).  Written by build_trees_kernel from the row's sum.
#define MVHDP_ROW_HEAVY 1
#define MVHDP_ROW_BIG 2

struct SweepLaunch {
    uint32_t sweep_idx;
    uint32_t seed_lo, seed_hi;
    uint32_t flags;                    // MVHDP_SWEEP_* (low 15 bits) | MVHDP_SL_STRICT_LIVE
    int32_t  S_cap;                    // dense-slot capacity per wave (multiple of 64)
    int32_t  waves_per_block;
    uint32_t block_shared_bytes;       // n_k delta table (unless nk_global) + the topic-list histogram
    int32_t  nk_global;                // 1: the n_k deltas go straight to the delta buffer (M*K too large to privatise in LDS)
    uint32_t wave_bytes;               // per-wave LDS region
    unsigned long long* stats;         // [16] device counters
    long long* act_key;                // activation key (atomicMin)
    // births of a live sweep (UPD:263-270 chunk by chunk; null: one activation per segment, by act_key): births[0] = how many of the topics
    // that were inactive at the segment start have been reached by a delta, [1] = their number n, [2 .. 2+n) = those topics in index order
    // (a new-topic draw, WRK:523-526, goes to births[2 + births[0]]), [2+K .. 2+2K) = a topic's position in that list or -1;
    // birth_keys[r] = the first delta (MVHDP_ACT_KEY) that reached topic births[2 + r] (its view takes alpha[m][K], UPD:268)
    int32_t* births;
    long long* birth_keys;
    // Work queue: waves pull entities in batches from one head.  Queue position q maps to an entity through two
    // segments: first the entities the route pass listed for this kernel (q_list[0 .. *q_list_count)), then
    // q_order_count entities of a static order (q_order[q_order_start + i*q_order_stride], or the identity when q_order
    // is null).  The stride cuts the longest-first order into interleaved segments (MVHDP_SWEEP_LIVE): every segment
    // sees the same length distribution, longest first.
    unsigned long long* doc_counter;
    const int32_t* q_list;
    const unsigned int* q_list_count;  // device memory (written by classify_kernel earlier in stream order), or nullptr = 0
    const int32_t* q_order;
    int64_t q_order_start, q_order_count, q_order_stride;
    // Speculative tree walk of the chunk head: a token's word tree is walked up front iff its first uniform u1 >= walk_theta[m]
    // (only a large u1 can reach the tree branch, WRK:529-535); a token that reaches it unwalked walks on demand, same
    // arithmetic, same result.  0 = walk every token.
    double walk_theta[MVHDP_MAXM];
    int32_t walk;                      // 1: launch the kernel flavour that knows about thresholds (and counts the per-view statistics)
    int32_t narrow;                    // 1: the flavour that gathers n_wk from the 16-bit mirror
    int32_t delta16;                   // 1 (deferred sweep, narrow flavour): the n_wk deltas of rows without MVHDP_ROW_BIG go to MvModel::delta16
    int32_t live16;                    // 1 (MVHDP_SWEEP_LIVE with narrow): the sweep's n_wk atomics of LIGHT rows go to the mirror itself, which is then the
                                       //   authoritative copy of those rows until the next tree build / widen pass; heavy rows: the 32-bit table as ever
    int32_t live_rows;                 // 1 (MVHDP_SWEEP_LIVE): the tree branch samples from the live n_wk row (MvModel::coef); tree[1] = MvModel::root of the
                                       //   segment start; only heavy words (their rows are not in the mirror) walk a stored tree; walk_theta[m] is also
                                       //   the u1 from which a token's row is loaded ahead of its turn
    int32_t coef_lds;                  // 1: the block keeps MvModel::coef in LDS (block_shared_bytes holds room for it)
    unsigned long long* slot_hist;     // [MVHDP_HIST_BINS] tokens of the entities whose NEW topic list has ceil(size/64) = 1..16, >16, then
                                       // [MVHDP_ENT_BINS] entities per kernel class 0..5 of that new list ([6]: not known, [7]: spare) --
                                       // what the next sweep's plan is made from
    // debug
    double* tok_dbg[MVHDP_MAXM];
    int32_t n_trace;
    const int64_t* trace_doc; const int32_t* trace_view; const int32_t* trace_pos;
    double* trace_out;
};

enum {
    ST_TOKENS = 0, ST_CHANGED, ST_NEW, ST_DOC, ST_TREE, ST_OOV, ST_ABORT, ST_FALLBACK, ST_NEGATIVE, ST_MISCLASS, ST_ONDEMAND,
    // per view, MVHDP_VIEW_STATS words each: tokens sampled, tokens that took the tree branch, and the histogram of the latter's u1
    // in MVHDP_WALK_BINS bins -- what a walk threshold would cost in walks on demand (mvhdp_sweep's threshold search)
    ST_VIEW_BASE, ST_VIEW_LAST = ST_VIEW_BASE + MVHDP_VIEW_STATS * MVHDP_MAXM - 1,
    // wave cycles by segment, summed over waves; filled only by a -DMVHDP_TIMING build (diagnostics)
    ST_T_QUEUE, ST_T_PROLOGUE, ST_T_VIEW, ST_T_CHUNK_HEAD, ST_T_TOKENS, ST_T_CHUNK_END, ST_T_TOTAL,
    ST_T_ENT0, ST_T_ENT1, ST_T_ENT2, ST_N_ENT0, ST_N_ENT1, ST_N_ENT2, ST_T_INIT, ST_T_FLUSH, ST_N_WAVES, ST_T_ROWS, ST_T_ROWS_WAIT,   // (timing build: a wave's first, second, later entities; block init and flush)
    ST_COUNT
};

size_t mvhdp_sweep_wave_bytes(int M, int S_cap);

hipError_t mvhdp_launch_build_counts(const MvModel& mm, const int64_t* n_tokens_per_view, hipStream_t s);
// write_full = false: only the descent table (what the register-resident kernels read), not the FTree.tree arrays
hipError_t mvhdp_launch_build_trees(const MvModel& mm, bool inference_leaves, bool write_full, hipStream_t s, bool beside_samplers = false);
// the same for the rows [row_begin, row_end) only; apply_first: counts += delta, delta = 0 for those rows before the build
hipError_t mvhdp_launch_build_trees_rows(const MvModel& mm, bool inference_leaves, bool write_full, int64_t row_begin, int64_t row_end,
                                         bool apply_first, unsigned long long* negatives, hipStream_t s);
// a segment border of a live16 sweep: the light rows are read from the 16-bit mirror (and written through to the 32-bit table), the heavy ones from the table
hipError_t mvhdp_launch_build_trees_from_mirror(const MvModel& mm, bool write_full, hipStream_t s, bool beside_samplers = false);
// end of a live16 sweep: counts <- mirror for the light rows
hipError_t mvhdp_launch_widen_mirror(const MvModel& mm, hipStream_t s);
// a segment start of a live sweep in its live-rows form: MvModel::coef from the current tokensPerTopic, then per row the weight class and
// 16-bit mirror (sweep start: from the 32-bit table; from_mirror: a later segment of a live16 sweep, light rows read from the mirror) and
// MvModel::root = sum_k coef_k (n_wk + beta) -- no tree, no descent table: one pass over the counts; then the stored trees of the HEAVY
// rows alone (with_heavy_trees: a sweep on the mirror, whose heavy words walk them)
hipError_t mvhdp_launch_live_rows_prepare(const MvModel& mm, bool from_mirror, bool with_heavy_trees, int32_t* heavy_list, unsigned int* heavy_ctl, int heavy_cap, int batch_cells, hipStream_t s);
// the stored trees of the listed heavy rows rebuilt from the live counts again and again until ctl[1] (stop) is set or 2 s have passed
hipError_t mvhdp_launch_heavy_refresh(const MvModel& mm, const int32_t* heavy_list, const unsigned int* ctl, int heavy_cap, int blocks, hipStream_t s);
hipError_t mvhdp_launch_set_u32(unsigned int* p, unsigned int v, hipStream_t s);
hipError_t mvhdp_launch_apply_nk(const MvModel& mm, unsigned long long* negatives, hipStream_t s);
// overlapped segments: dst (counts / mirror / descent tables of the copy segment s + 2 reads) += dA (+ dB, zeroed), trees rebuilt; see the kernel
hipError_t mvhdp_launch_apply2_counts(const MvModel& dst, const int32_t* dA, int32_t* dB, bool use_mirror, unsigned long long* negatives, hipStream_t s);
hipError_t mvhdp_launch_apply_sparse(const MvModel& dst, int32_t* d, bool use_mirror, hipStream_t s);
hipError_t mvhdp_launch_gate(const unsigned long long* qhead, unsigned long long threshold, hipStream_t s);
hipError_t mvhdp_launch_init_from_trees(const MvModel& mm, uint32_t seed_lo, uint32_t seed_hi, hipStream_t s);
hipError_t mvhdp_launch_draw_p(const MvModel& mm, uint32_t sweep_idx, uint32_t seed_lo, uint32_t seed_hi, hipStream_t s);
hipError_t mvhdp_launch_sweep(const MvModel& mm, const SweepLaunch& sl, int grid_blocks, bool debug, hipStream_t s);
hipError_t mvhdp_launch_apply_delta(const MvModel& mm, unsigned long long* stats, hipStream_t s, bool with_delta16 = false);
hipError_t mvhdp_launch_delay(int microseconds, hipStream_t s);
// zeroes the given counter arrays (any may be null) and sets *act_key to "none", in one launch
hipError_t mvhdp_launch_ctl_reset(unsigned long long* stats, int n_stats, long long* act_key, unsigned long long* meta, int n_meta,
                                  unsigned int* class_counts, unsigned long long* qheads, hipStream_t s);
// MVHDP_SWEEP_LIVE helpers: 0 = delta <- -counts, 1 = delta <- counts + delta (after - before), counts <- snapshot, 2 = count negatives
hipError_t mvhdp_launch_live_helper(const MvModel& mm, int which, unsigned long long* stats, hipStream_t s);
hipError_t mvhdp_launch_doc_topic_hist(const MvModel& mm, int m, int32_t* hist, int32_t hist_len,
                                       int32_t* doc_len_counts, int32_t len_len, hipStream_t s);
hipError_t mvhdp_sweep_set_max_lds(size_t bytes);
hipError_t mvhdp_launch_count_hist(const MvModel& mm, int m, int32_t* hist, int32_t len, hipStream_t s);
hipError_t mvhdp_launch_view_overlap(const MvModel& mm, double* out, hipStream_t s);
hipError_t mvhdp_launch_loglik_doc(const MvModel& mm, int m, double* doc_out, hipStream_t s);
hipError_t mvhdp_launch_loglik_topic(const MvModel& mm, int m, double* partial, int n_partial, unsigned long long* nonzero, hipStream_t s);
hipError_t mvhdp_launch_gamma_doc_stats(const MvModel& mm, int m, double gamma_m, uint32_t seed_lo, uint32_t seed_hi, uint32_t round,
                                        double* partial, int n_blocks, hipStream_t s);
hipError_t mvhdp_launch_dp_tables(const int32_t* hist, int hist_len, int K, int m, const double* conc, uint32_t seed_lo, uint32_t seed_hi, uint32_t round,
                                  double* mk, uint8_t* active, hipStream_t s);
hipError_t mvhdp_launch_antoniak_draws(int n, const int32_t* items, const double* conc, uint32_t seed_lo, uint32_t seed_hi, uint32_t round, int32_t* tables, hipStream_t s);
// counts every entity's topic list from z: writes MvModel::nslots and the histograms of SweepLaunch::slot_hist
hipError_t mvhdp_launch_slot_hist(const MvModel& mm, unsigned long long* hist, hipStream_t s);
struct DocTopicCarry { const int64_t* src[MVHDP_MAXM]; };   // per view [D]: the entity whose counts score entity d (PTM:2873-2886)
hipError_t mvhdp_launch_doc_topic_prop(const MvModel& mm, const DocTopicCarry& carry, const double* w_dev, int64_t d0, int64_t d1, double* out_dev, hipStream_t s);
// Classes of the sweep kernels by topic-list size: 0..4 = the register-resident variants with 64 << c slots, 5 = the generic LDS kernel
#define MVHDP_N_CLASSES 6
struct ClassifyArgs {
    const int32_t* order;              // entities to route: order[start + i*stride], i in [0, n) (nullptr = identity)
    int64_t n, start, stride;
    int32_t class_map[MVHDP_N_CLASSES]; // class of a topic list -> the launched kernel class that takes it (the primary for narrower lists, the
                                       // next wider launched class where a class has no kernel of its own); -1: nobody (an error, counted)
    int32_t check_views;               // 1: some entity has a view beyond 65535 tokens (then the views' lengths are looked at)
    int32_t* lists[MVHDP_N_CLASSES];   // per class, capacity n
    unsigned int* counts;              // [MVHDP_N_CLASSES]
    unsigned long long* misrouted;     // counts entities whose list size is not known (the host should have recounted)
};
hipError_t mvhdp_launch_classify(const MvModel& mm, const ClassifyArgs& ca, hipStream_t s);
size_t mvhdp_sweep_fast_wave_bytes(int M, int S_cap, int rmax);
hipError_t mvhdp_launch_sweep_fast(const MvModel& mm, const SweepLaunch& sl, int rmax, int grid_blocks, bool debug, hipStream_t s);
int mvhdp_sweep_fast_occupancy(int rmax, bool debug, bool walk, int block_threads, size_t lds_bytes);
int mvhdp_sweep_generic_occupancy(bool debug, int block_threads, size_t lds_bytes);
// VGPRs of the compiled sweep kernel of class c (0..4 register-resident, 5 generic); flavour 0: plain, 1: walk, 2: debug
int mvhdp_sweep_kernel_regs(int cls, int flavour);
int mvhdp_sweep_generic_regs(bool debug);

#define MVHDP_DOC_BATCH 2
#define MVHDP_HIST_BINS 17
#define MVHDP_ENT_BINS 8
#define MVHDP_NSLOTS_UNKNOWN 0xFFFFu

// kernel class of a topic list of n slots (0..4: the register-resident variants with 64 << c slots, 5: the generic LDS kernel);
// the 8- and 16-round variants count tokens per slot in 16 bits, so an entity with a view beyond 65535 tokens is generic too
__host__ __device__ __forceinline__ int mvhdp_class_of(int n, bool view_beyond_16_bits)
{
    int c = (n <= 64) ? 0 : (n <= 128) ? 1 : (n <= 256) ? 2 : (n <= 512) ? 3 : (n <= 1024) ? 4 : 5;
    if (c >= 3 && view_beyond_16_bits) c = 5;
    return c;
}

// FTree.sample (FT:111-136) through the descent table (MvModel::dtab), one lane per token: the same reads, comparisons and
// subtractions as the literal descent over FTree.tree, three levels per 64-byte block (see the chunk head of the
// register-resident sweep kernel, which walks the same table the same way).
__device__ __forceinline__ int dtab_sample(const MvModel& mm, int64_t row, double u01)
{
    const int K = mm.K;
    const double* __restrict__ dt = mm.dtab + row * (int64_t)mm.dt_nblk * 8;
    double u = 0.0;
    int i = 1;
    for (int bd = 0; bd < mm.dt_nbd; bd++) {
        if (bd == 0 || i < K) {
            const double2* __restrict__ blk = (const double2*)(dt + (int64_t)(mm.dt_base[bd] + (i - (1 << mm.dt_depth[bd]))) * 8);
            const double2 q0 = blk[0], q1 = blk[1], q2 = blk[2], q3 = blk[3];
            const int levels = (bd == 0) ? mm.dt_f : 3;
            if (bd == 0) u = u01 * q3.y;                                              // FT:120  u *= tree[1]
            int path = 0;
            if (i < K && levels > 0) {                                                // FT:122-130
                const double l = q0.x;
                if (u < l) { i = 2 * i; } else { u = u - l; i = 2 * i + 1; path = 1; }
            }
            if (i < K && levels > 1) {
                const double l = path ? q1.x : q0.y;
                if (u < l) { i = 2 * i; path = 2 * path; } else { u = u - l; i = 2 * i + 1; path = 2 * path + 1; }
                if (i < K && levels > 2) {
                    const double l3 = (path == 0) ? q1.y : (path == 1) ? q2.x : (path == 2) ? q2.y : q3.x;
                    if (u < l3) { i = 2 * i; } else { u = u - l3; i = 2 * i + 1; }
                }
            }
        }
    }
    return i - K;                                                                     // FT:132
}

// The same walk for ONE token on behalf of the whole wave (row and u01 wave-uniform): the blocks come through the scalar
// data cache (s_load, counted by lgkmcnt), so the walk never waits behind the vector-memory operations the wave has in
// flight -- the n_wk gather of the next token, the previous chunk's atomics.  The descent table is read-only for the
// duration of a kernel (the scalar cache is invalidated at every kernel start), the arithmetic is that of dtab_sample.
__device__ __forceinline__ int dtab_sample_uniform(const MvModel& mm, int64_t row, double u01)
{
    typedef const __attribute__((address_space(4))) double* cdptr;
    const int K = mm.K;
    const cdptr dt = (cdptr)(mm.dtab + row * (int64_t)mm.dt_nblk * 8);
    double u = 0.0;
    int i = 1;
    for (int bd = 0; bd < mm.dt_nbd; bd++) {
        if (bd == 0 || i < K) {
            const cdptr blk = dt + (int64_t)(mm.dt_base[bd] + (i - (1 << mm.dt_depth[bd]))) * 8;
            const double q0x = blk[0], q0y = blk[1], q1x = blk[2], q1y = blk[3], q2x = blk[4], q2y = blk[5], q3x = blk[6], q3y = blk[7];
            const int levels = (bd == 0) ? mm.dt_f : 3;
            if (bd == 0) u = u01 * q3y;                                               // FT:120  u *= tree[1]
            int path = 0;
            if (i < K && levels > 0) {                                                // FT:122-130
                const double l = q0x;
                if (u < l) { i = 2 * i; } else { u = u - l; i = 2 * i + 1; path = 1; }
            }
            if (i < K && levels > 1) {
                const double l = path ? q1x : q0y;
                if (u < l) { i = 2 * i; path = 2 * path; } else { u = u - l; i = 2 * i + 1; path = 2 * path + 1; }
                if (i < K && levels > 2) {
                    const double l3 = (path == 0) ? q1y : (path == 1) ? q2x : (path == 2) ? q2y : q3x;
                    if (u < l3) { i = 2 * i; } else { u = u - l3; i = 2 * i + 1; }
                }
            }
            i = __builtin_amdgcn_readfirstlane(i);                                    // the next block's address stays scalar
        }
    }
    return i - K;                                                                     // FT:132
}
