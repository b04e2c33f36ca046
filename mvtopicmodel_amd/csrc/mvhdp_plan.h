// mvhdp_plan.h — the host-side decisions of one Gibbs sweep as PURE functions: no HIP call, no clock, no environment.
//
//   plan_sweep()   what the last sweep left behind (tokens by topic-list size, entities by kernel class), the corpus
//                  geometry, the sweep flags and the tuning block  ->  which kernel variants run, over which entities, on
//                  which stream, with which walk thresholds, grids and LDS sizes
//   WalkTuner      the clock-driven search of the chunk head's walk threshold: propose() before a sweep, observe() after
//
// mvhdp_sweep() = plan_sweep() -> enqueue (mvhdp_api.hip: launches only) -> finish (read-back, WalkTuner::observe).
// Both are reachable without a GPU through mvhdp_plan_probe / mvhdp_tuner_probe (include/mvhdp.h) so that the CPU tests
// can feed them recorded inputs (tests/test_plan.py).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <algorithm>
#include <cmath>
#include <cstring>

#include "mvhdp_device.h"
#include "../../include/mvhdp.h"

// Streams of one sweep: the primary variant on the handle's stream, every wider class that carries weight on a side stream of its own
// (the 4-round class behind the 2-round one where it is a handful of entities: see stream_of in plan_sweep).  HIP maps the
// streams of one priority onto at most four hardware queues (GPU_MAX_HW_QUEUES), the null stream's included, and two streams that share
// a hardware queue run one after the other: the C5 trace of round 4 (profiles/r04_timeline_c5_*.txt) showed the 8-, 4- and 2-round
// kernels in ONE queue -- 5, 9 and 20 ms in series, the first two at two waves per SIMD with nothing beside them.  So the streams are
// spread over the runtime's two queue pools:
//   normal priority   the null stream, the handle's stream, side streams C and D (the 2- and 4-round classes) -- four; a fifth, the second
//                     stream of overlapped segments, shares a queue with one of them
//   high priority     side streams A and B (the 16- and 8-round classes: the long entities are the sweep's critical path anyway)
// Measured (gpurun_out/r4y, ms over sweeps 5-24): C5 deferred 44.0 -> 42.2, C5 in 8 segments 64.1 -> 58.1; C4 deferred / segmented / live
// unchanged.  D in the high pool as well (so that no normal stream shares) costs the overlapped segments of C4 1.5 ms a sweep: a busy
// high-priority queue delays the launches of the small kernels between the segments.
enum { PLAN_STREAM_MAIN = 0, PLAN_STREAM_A = 1, PLAN_STREAM_B = 2, PLAN_STREAM_C = 3, PLAN_STREAM_D = 4, PLAN_N_STREAMS = 5 };

// Register counts of the compiled kernels (hipFuncGetAttributes at mvhdp_create; typical values in the CPU tests):
// [class 0..4][0: plain, 1: walk flavour, 2: debug], [5][0/2]: the generic kernel
struct PlanRegs { int regs[MVHDP_N_CLASSES][3]; };

struct PlanIn {
    int K = 0, M = 0;
    int64_t D = 0;                              // entities of this handle
    int64_t mdt = 0;                            // tokens of the longest entity (all views)
    int64_t n_longer[5] = {0, 0, 0, 0, 0};      // entities with more than 64 << c tokens, c = 0..4 (a prefix of the longest-first order)
    bool have_order = false;                    // the longest-first order is on the device (else: natural order, everything is routed)
    unsigned long long tok_hist[MVHDP_HIST_BINS] = {0};   // tokens by ceil(topic-list size / 64) of their entity, for THIS sweep's lists
    unsigned long long ent_hist[MVHDP_ENT_BINS] = {0};    // entities by kernel class; [MVHDP_N_CLASSES]: size not known
    uint32_t flags = 0;                         // MVHDP_SWEEP_*
    bool debug = false;
    bool batch = false;                         // mvhdp_sweep_many: the plan must hold for several sweeps (lists move between them)
    bool trees_current = false;                 // the F+trees (and the 16-bit mirror) match the counts before this call
    bool unassigned = false;                    // some token may still be unassigned (z = -1): row totals can grow during the sweep
    int first_inactive = -1;
    int num_cus = 256;
    size_t max_lds = 160 * 1024;
    PlanRegs regs{};
};

struct ClassLaunch {
    bool used = false;                          // a kernel of this class is launched
    bool fast = false;                          // register-resident variant (else the generic LDS kernel)
    int r = 0;                                  // slot rounds (1 << class)
    int S_cap = 0;
    uint32_t wave_bytes = 0;
    int wpb = 0;                                // waves per block
    size_t lds = 0;
    int grid = 0;                               // resident blocks (the launch takes min(grid, what its entities need))
    int stream = PLAN_STREAM_MAIN;
    int walk = 0, narrow = 0;                   // kernel flavour
    double theta[MVHDP_MAXM] = {0};             // walk thresholds per view
};

struct SweepPlan {
    int err = 0;                                // MVHDP_OK or an mvhdp_status
    const char* msg = "";
    uint32_t flags = 0;
    bool live = false, seg_apply = false, frozen = false, debug = false;
    bool overlap = false;                       // two segments in flight (SEGMENT_OVERLAP, or a live sweep with live_overlap): the grids leave one block per CU free
    int nseg = 1;
    int only_seg = -1;                          // >= 0: only this segment is swept (MVHDP_SWEEP_ONLY_SEGMENT)
    int S_cap = 0;
    bool fast = false;                          // register-resident variants in use
    int pc = 0;                                 // class of the primary variant
    bool route = false;                         // a route pass lists the entities of the prefix [0, H) by class
    int64_t H = 0;
    ClassLaunch cls[MVHDP_N_CLASSES];
    int class_map[MVHDP_N_CLASSES] = {0, 1, 2, 3, 4, 5};   // class of a topic list -> the launched class that takes it
    bool need_full = false;                     // FTree.tree itself is read (generic kernel, debug trace)
    int nk_global = 0;
    uint32_t block_shared_bytes = 0;
    bool delta16 = false;                       // a deferred sweep whose narrow kernels keep the deltas of the small rows in 16-bit cells (SweepLaunch::delta16)
    bool live16 = false;                        // a live sweep whose light rows are kept current in the 16-bit mirror (every kernel the NARROW flavour)
    bool live_rows = false;                     // a live sweep whose tree branch samples from the live count rows (SweepLaunch::live_rows): no trees, no segment overlap
    bool coef_lds = false;                      //   ... with the coefficient table of the segment in every block's LDS (block_shared_bytes holds room for it)
    int dominant = 0;                           // class holding most tokens (its group's walk threshold is the one being searched)
    int walk_cfg = 0;                           // key of the configuration the walk search compares sweeps within
};

// tokens of the histogram that fall into kernel class c (bins are rounds of 64 slots: class c = rounds (2^(c-1), 2^c])
static inline double plan_class_tokens(const unsigned long long* tok_hist, int c)
{
    double t = 0;
    for (int b = 0; b < MVHDP_HIST_BINS; b++) {
        const int rounds = b + 1;
        int cb = 0;
        while (cb < 5 && rounds > (1 << cb)) cb++;
        if (b == MVHDP_HIST_BINS - 1) cb = 5;
        if (cb == c) t += (double)tok_hist[b];
    }
    return t;
}

static inline int plan_blocks_per_cu(int regs, int threads, size_t lds)
{
    int r = (regs + 7) / 8 * 8;
    int waves_simd = r > 0 ? 512 / r : 8;
    waves_simd = std::max(1, std::min(8, waves_simd));
    const int wpb = threads / 64;
    int b = waves_simd * 4 / wpb;
    b = std::min(b, (int)((160 * 1024) / (lds > 0 ? lds : 1)));
    b = std::min(b, 32 / wpb);
    return std::max(1, b);
}

size_t mvhdp_sweep_wave_bytes(int M, int S_cap);
size_t mvhdp_sweep_fast_wave_bytes(int M, int S_cap, int rmax);

// The tuning block (include/mvhdp.h mvhdp_tuning): what a host may pin, and what the library has learnt
struct PlanTuning {
    int force_primary = 0;                      // 0: choose; 1, 2, 4, 8, 16: the primary variant; 32: the generic kernel for everything
    int narrow = -1;                            // -1: whenever legal; 0: never; 1: the 1-round variant only
    int walk_fixed = 0;                         // 1: walk_theta below as it stands, no search
    double walk_theta[MVHDP_MAXM] = {0};
    double primary_min_share = 0.10;            // the narrowest class holding at least this share of the tokens gets its own (primary) kernel
    int single_stream = 0;                      // 1: every class kernel on the handle's stream, one after another (diagnostics)
    int four_round_on_c = -1;                   // the 4-round class behind the 2-round one on stream C: -1 where it holds under 3 % of the tokens, 0 never, 1 always
    int delta16 = 1;                            // 1: 16-bit delta cells for the rows that cannot overflow them (plain deferred sweeps); 0: never
    int narrow_wide = 1;                        // 1: deferred sweeps gather from the mirror in the wider variants too (0: the 1-round variant only)
    int fork_delay_us = 0;                      // microseconds the handle's stream is held between the fork event and the primary kernel (0: none)
    int widest_on_main = 0;                     // 1: the widest class on the handle's stream, the primary on a side stream (diagnostics)
    int live_overlap = -1;                      // live sweeps: segments overlapped (two in flight); 0: one after the other
    int single_wave = 0;                        // diagnostics: every sweep kernel as one wavefront, class kernels one after another, live sweeps strictly ordered
    int live16 = -1;                            // live sweeps keep the light rows current in the 16-bit mirror: -1 where a row has at least 1 KiB (K >= 256), 0 never, 1 always
    int live_rows = -1;                         // live sweeps sample their tree branch from the live count rows (no stored trees): -1 / 1 wherever every kernel is
                                                //   register-resident (default), 0: stored trees rebuilt at every segment border (the round-4 form)
    int live_rows_segments = 1;                 // segments of such a sweep when the flags name none (a border refreshes tokensPerTopic and the roots: one pass over the counts)
    int coef_lds_max_bytes = 12 * 1024;         // live-rows form: the coefficient table [M][Kp] sits in every block's LDS up to this size (C4: 4.8 KB; C5's 20 KB stay in global memory by default)
    double live_rows_theta = 0.3;               // steered views: a token's row is loaded ahead of its turn iff its u1 reaches this (a token below it that reaches the tree
                                                //   branch after all loads it then)
};

// Walk-threshold search ("The thresholded walk").  The threshold changes WHEN a word tree is walked, never what
// is sampled, so it is steered by the clock: sweeps at the current threshold (A) alternate with sweeps a step away (B); B replaces A
// when its kernel time per token beats the mean of the A sweeps on either side (the chain's own drift cancels).  An upward step is as
// long as the last sweep's histogram of the tree-branch tokens' u1 says is nearly free (<= walk_cap of the tokens more to walk on
// demand); a step that does not pay turns the search around, two in a row let it rest for a growing number of sweeps, and the first
// B sweep after a rest tries half the threshold.  Views where most tokens take the tree branch anyway (share >= 0.35) are always
// walked up front.  What pays depends on what the kernel is bound by -- a walk skipped saves three lines of traffic, a walk on demand
// stalls its wave -- so every kernel flavour keeps its own threshold and starts from what was measured on C4:
//   group 0  1-round variant gathering from the 16-bit mirror   0.30  (flat from 0.1 to 0.45, 10 % slower at 0.6)
//   group 1  1-round variant gathering 32-bit rows              1.00  (bandwidth-bound: every walk on demand)
//   group 2  the wider variants                                 0.45  (flat from 0.3 to 0.65)
// and 0 for all of them where a row is short (K < 256: latency-bound, the best threshold is 0).
enum { WALK_GROUPS = 3 };
struct WalkTuner {
    int walk_i = 0, walk_probe_i = 0, walk_b_i = 0, walk_phase = 0, walk_dir = 1, walk_fails = 0, walk_wait = 4, walk_cfg = -1, walk_maxj = 6;
    bool walk_far = false;
    double walk_ns_a1 = 0.0, walk_ns_b = 0.0;
    int walk_cls = 2, walk_i_by[WALK_GROUPS] = {-1, -1, -1};
    int walk_default[WALK_GROUPS] = {0, 0, 0};

    void init_defaults(int K)
    {
        if (K >= 256) { walk_default[0] = 6; walk_default[1] = MVHDP_WALK_BINS; walk_default[2] = 9; }
        walk_i = walk_default[walk_cls];
    }
    int stored(int g) const { return walk_i_by[g] >= 0 ? walk_i_by[g] : walk_default[g]; }
    double walk_cap = 0.004;
    long long walk_idle_until = 0, walk_refresh_at = 0, sweeps_done = 0;
    double walk_f[MVHDP_MAXM] = {-1, -1, -1, -1, -1, -1, -1, -1};   // tree-branch share per view in the last measured sweep (< 0: not known)
    double walk_hist[MVHDP_WALK_BINS] = {0};                          // tree-branch tokens of the steered views by u1 bin, as a share of all tokens

    bool controlled(int m) const { return walk_f[m] >= 0.0 && walk_f[m] < 0.35; }
    bool any_controlled(int M) const { for (int m = 0; m < M; m++) if (controlled(m)) return true; return false; }
    bool any_unknown(int M) const { for (int m = 0; m < M; m++) if (walk_f[m] < 0.0) return true; return false; }

    // Thresholds of the coming sweep: group `grp` is the one being searched (its kernel holds most tokens), the other groups run at
    // their own stored thresholds.  theta[g][m]; *measure: the walk flavour must run although every threshold is 0 (its statistics
    // are due).
    void propose(int grp, int M, double theta[WALK_GROUPS][MVHDP_MAXM], bool* measure)
    {
        if (grp != walk_cls) {
            walk_i_by[walk_cls] = walk_i;
            walk_i = stored(grp);
            walk_cls = grp;
            walk_phase = 0; walk_far = false;
        }
        const bool any = any_controlled(M);
        const int top = MVHDP_WALK_BINS;                      // thresholds up to 1 (= no token of the view walked up front)
        walk_probe_i = walk_i;
        if (walk_phase == 1 && any) {
            if (walk_far && walk_i < 4) walk_far = false;
            if (walk_far) walk_dir = -1;
            if (walk_dir > 0 && walk_i >= top) walk_dir = -1;
            if (walk_dir < 0 && walk_i <= 0) walk_dir = 1;
            if (walk_dir > 0) {
                int j = 1;
                double extra = walk_hist[walk_i];
                while (walk_i + j < top && j < walk_maxj && extra + walk_hist[walk_i + j] <= walk_cap) { extra += walk_hist[walk_i + j]; j++; }
                walk_probe_i = walk_i + j;
            } else walk_probe_i = walk_far ? walk_i / 2 : walk_i - 1;
        }
        for (int m = 0; m < MVHDP_MAXM; m++) {
            const bool c = m < M && controlled(m);
            for (int g = 0; g < WALK_GROUPS; g++) theta[g][m] = c ? (double)(g == grp ? walk_probe_i : stored(g)) / MVHDP_WALK_BINS : 0.0;
        }
        *measure = false;
        // no view qualifies: look again every 16th sweep (the statistics come from the walk flavour only)
        if (!any && sweeps_done >= walk_refresh_at) { *measure = true; walk_refresh_at = sweeps_done + 16; }
        if (any_unknown(M)) *measure = true;                  // the first sweep measures (threshold 0)
    }

    // per-view statistics of a sweep whose kernels ran the walk flavour
    void measured(int M, int nseg, const unsigned long long* view_stats /* [M][MVHDP_VIEW_STATS] */)
    {
        double all = 0.0;
        for (int m = 0; m < M; m++) {
            const double n = (double)view_stats[m * MVHDP_VIEW_STATS];
            all += n;
            if (n >= 64) walk_f[m] = (double)view_stats[m * MVHDP_VIEW_STATS + 1] / n;
            else if (nseg == 1) walk_f[m] = 1.0;              // a view with next to no tokens is never steered
        }
        for (int b = 0; b < MVHDP_WALK_BINS; b++) {
            double c = 0.0;
            for (int m = 0; m < M; m++)
                if (controlled(m)) c += (double)view_stats[m * MVHDP_VIEW_STATS + 2 + b];
            walk_hist[b] = all > 0 ? c / all : 0.0;
        }
    }

    // the sweep is over: ns = sweep-kernel nanoseconds per token; comparable: it may be held against its neighbours (same kernel
    // configuration `cfg`, no debug / frozen / fixed thresholds)
    void observe(bool comparable, int cfg, int M, double ns)
    {
        sweeps_done++;
        const bool any = any_controlled(M);
        if (!comparable || walk_cfg != cfg) {
            walk_phase = 0;
            walk_cfg = comparable ? cfg : -1;
            if (comparable) walk_ns_a1 = 0.0;
        }
        if (comparable && !any) walk_phase = 0;
        if (!(comparable && any)) return;
        if (walk_phase == 0) {
            walk_ns_a1 = ns;
            if (sweeps_done >= walk_idle_until) walk_phase = 1;
        } else if (walk_phase == 1) {
            walk_ns_b = ns; walk_b_i = walk_probe_i; walk_phase = 2;
        } else {
            const double base = 0.5 * (walk_ns_a1 + ns);
            const int step = walk_b_i - walk_i;
            const bool far = walk_far;
            walk_far = false;
            // leaving threshold 0 also changes the kernel flavour: ask for more there (no flapping between the two)
            const double need = walk_i == 0 ? 0.005 : 0.0025;
            if (std::fabs(walk_ns_a1 - ns) > 0.025 * base) {          // the A sweeps disagree (a variant change, a jump of the chain): no verdict
                walk_ns_a1 = ns; walk_phase = 1; walk_far = far;
            } else if (step != 0 && walk_ns_b < base * (1.0 - need)) {
                walk_i = walk_b_i;
                walk_ns_a1 = walk_ns_b;                               // the B sweep is the first A sweep of the next step
                walk_fails = 0; walk_wait = 4; walk_phase = 1;
                if (step > 0) walk_maxj = std::min(6, walk_maxj * 2);
                if (step > 0 && walk_ns_b < base * (1.0 - 0.008)) walk_cap = std::min(0.05, walk_cap * 2.0);
                if (far) walk_far = true;                             // half again
            } else if (step > 1) {                                    // a long step that did not pay: a shorter one, same direction
                walk_ns_a1 = ns;
                walk_maxj = std::max(1, step / 2);
                walk_cap = std::max(0.004, walk_cap * 0.5);
                walk_phase = 1;
            } else if (far) {                                         // half the threshold is no better: back to single steps
                walk_ns_a1 = ns;
                walk_dir = 1;
                walk_phase = 1;
            } else {
                walk_ns_a1 = ns;
                walk_dir = -walk_dir;
                walk_phase = 1;
                if (++walk_fails >= 2) {
                    walk_fails = 0;
                    walk_idle_until = sweeps_done + walk_wait;
                    walk_wait = std::min(64, walk_wait * 2);
                    walk_phase = 0;
                    walk_far = true;
                }
            }
        }
    }

    // a host hands back thresholds learnt earlier (mvhdp_set_tuning): settled, no search for a while
    void restore(const int32_t* steps /*[WALK_GROUPS]*/, const double* f, int M)
    {
        for (int g = 0; g < WALK_GROUPS; g++) walk_i_by[g] = steps[g];
        walk_i = stored(walk_cls);
        for (int m = 0; m < M; m++) walk_f[m] = f[m];
        walk_phase = 0; walk_far = false; walk_fails = 0;
        walk_idle_until = sweeps_done + 16; walk_wait = 16;
        walk_cfg = -1;
    }
};

// One sweep's plan.  `wt` proposes the walk thresholds (its state moves: a proposal is part of the search).
static inline void plan_sweep(const PlanIn& in, const PlanTuning& tu, WalkTuner& wt, SweepPlan& p)
{
    p = SweepPlan();
    auto fail = [&](int code, const char* m) { p.err = code; p.msg = m; };
    uint32_t flags = in.flags;
    if (flags & MVHDP_SWEEP_FROZEN) flags |= MVHDP_SWEEP_REUSE_TREES | MVHDP_SWEEP_NO_APPLY;   // nut == 0: the model is read-only
    if (flags & ~(MVHDP_SWEEP_REUSE_TREES | MVHDP_SWEEP_NO_APPLY | MVHDP_SWEEP_EXACT_CHAIN | MVHDP_SWEEP_GENERIC_KERNEL | MVHDP_SWEEP_FROZEN |
                  MVHDP_SWEEP_LIVE | MVHDP_SWEEP_LIVE_SEGMENTS(0xff) | MVHDP_SWEEP_SEGMENT_APPLY | MVHDP_SWEEP_SEGMENT_OVERLAP | 0xff000000u)) return fail(MVHDP_ERR_INVALID_ARG, "sweep: unknown flag");
    p.flags = flags;
    p.live = (flags & MVHDP_SWEEP_LIVE) != 0;
    p.seg_apply = (flags & MVHDP_SWEEP_SEGMENT_APPLY) != 0;
    p.frozen = (flags & MVHDP_SWEEP_FROZEN) != 0;
    p.debug = in.debug;
    if (p.seg_apply && (flags & (MVHDP_SWEEP_LIVE | MVHDP_SWEEP_NO_APPLY | MVHDP_SWEEP_FROZEN | MVHDP_SWEEP_REUSE_TREES)))
        return fail(MVHDP_ERR_INVALID_ARG, "sweep: SEGMENT_APPLY excludes LIVE, NO_APPLY, FROZEN and REUSE_TREES");
    if (p.live && p.frozen) return fail(MVHDP_ERR_INVALID_ARG, "sweep: LIVE and FROZEN exclude each other");
    if ((flags & MVHDP_SWEEP_SEGMENT_OVERLAP) && !p.seg_apply) return fail(MVHDP_ERR_INVALID_ARG, "sweep: SEGMENT_OVERLAP goes with SEGMENT_APPLY");
    if ((flags & MVHDP_SWEEP_SEGMENT_OVERLAP) && in.first_inactive >= 0) return fail(MVHDP_ERR_UNSUPPORTED, "sweep: SEGMENT_OVERLAP with inactive topics (the activation needs the host between segments)");
    if ((flags & MVHDP_SWEEP_SEGMENT_OVERLAP) && in.debug) return fail(MVHDP_ERR_UNSUPPORTED, "sweep: SEGMENT_OVERLAP with debug outputs");
    // live / segmented sweeps: the entities are cut into nseg interleaved segments of the longest-first order
    // (a deferred sweep accepts a segment count too: same integers as one segment, the trees being those of the snapshot)
    // (the live-rows form of a live sweep: decided here, before the segment count and the grids, from what cannot change below -- no
    // entity can reach the generic LDS kernel, whose tree branch reads FTree.tree)
    bool want_rows = p.live && !p.frozen && !in.debug && tu.live_rows != 0 && !(flags & (MVHDP_SWEEP_REUSE_TREES | MVHDP_SWEEP_GENERIC_KERNEL)) &&
                     tu.force_primary != 32 && std::min<int64_t>(in.K, std::max<int64_t>(in.mdt, 1)) <= 1024 && in.mdt <= 65535;
    int nseg = (int)((flags >> 16) & 0xffu);
    // (a truncated HDP needs no more: in the live-rows form topics are born chunk by chunk, SweepLaunch::births -- the stored-tree form and a
    // document shard give birth to one topic per segment border / exchange)
    if (nseg == 0) nseg = want_rows ? std::max(1, tu.live_rows_segments) : (p.live || p.seg_apply) ? 4 : 1;
    p.only_seg = (int)(flags >> 24) - 1;                  // MVHDP_SWEEP_ONLY_SEGMENT(s): -1 = every segment
    // (more segments than entities: a whole sweep uses fewer; a single-segment call keeps the caller's count -- segments beyond the
    // last entity are empty -- so that document shards of different sizes walk through the same number of exchanges)
    if ((int64_t)nseg > in.D && p.only_seg < 0) nseg = (int)std::max<int64_t>(1, in.D);
    p.nseg = nseg;
    // two segments in flight (mvhdp_api.hip enqueue_overlapped): asked for (SEGMENT_OVERLAP), or a live sweep of several segments
    // whose borders need no host (no inactive topic waiting for its activation, no debug output)
    p.overlap = nseg > 1 && p.only_seg < 0 &&
                ((flags & MVHDP_SWEEP_SEGMENT_OVERLAP) || (p.live && !want_rows && !tu.single_wave && tu.live_overlap != 0 && in.first_inactive < 0 && !in.debug));
    if (p.only_seg >= 0) {
        if (p.live || p.seg_apply) return fail(MVHDP_ERR_INVALID_ARG, "sweep: ONLY_SEGMENT excludes LIVE and SEGMENT_APPLY");
        if (p.only_seg >= nseg) return fail(MVHDP_ERR_INVALID_ARG, "sweep: ONLY_SEGMENT beyond the segment count");
    }
    const int K = in.K, M = in.M;
    int S_cap = (int)std::min<int64_t>(K, std::max<int64_t>(in.mdt, 1));
    S_cap = (S_cap + 63) / 64 * 64;
    p.S_cap = S_cap;
    // the block's private n_k delta table: in LDS up to 24 KiB (C5: 20 KB), beyond that (e.g. K = 2048 with 8 views: 64 KB, which
    // would not leave room for the slot state) the deltas go straight to the delta buffer.  A live sweep keeps the private table
    // too: M*K hot words would take every token's two atomics one after the other at the memory side (measured on C3: 28 ms per
    // sweep instead of 5.7), so tokensPerTopic becomes current at each segment end -- together with the trees.
    p.nk_global = ((size_t)M * K * sizeof(int) > 24 * 1024) ? 1 : 0;
    p.block_shared_bytes = (uint32_t)((((size_t)(p.nk_global ? 0 : M * K) + MVHDP_HIST_BINS + MVHDP_ENT_BINS + MVHDP_MAXM * MVHDP_VIEW_STATS) * sizeof(int) + 15) & ~(size_t)15);
    // live-rows form: the coefficient table [M][Kp] in LDS where it is small (C4: 4.8 KB; not C5's 20 KB, whose wide variants need their LDS for
    // the slot state: those read it from global memory)
    {
        const size_t coef_bytes = (size_t)M * ((K + 7) & ~7) * sizeof(float);
        p.coef_lds = want_rows && coef_bytes <= (size_t)tu.coef_lds_max_bytes;
        if (p.coef_lds) p.block_shared_bytes += (uint32_t)((coef_bytes + 16 + 15) & ~(size_t)15);
    }

    // ---- which classes hold entities (this sweep's topic lists), and the primary variant ----
    double tok[MVHDP_N_CLASSES], tot = 0;
    for (int c = 0; c < MVHDP_N_CLASSES; c++) { tok[c] = plan_class_tokens(in.tok_hist, c); tot += tok[c]; }
    const bool unknown = in.ent_hist[MVHDP_N_CLASSES] != 0 || tot == 0 || in.batch || p.only_seg >= 0;   // the lists are not all known (or will move): launch whatever is reachable
    // widest class any entity can reach: a list is no longer than the entity (tokens) nor than K
    int c_max = 0;
    while (c_max < 5 && S_cap > (64 << c_max)) c_max++;
    const bool view_16 = in.mdt > 65535;                     // some view may be beyond the 16-bit slot counts of the wide variants (over-estimate: the entity's total)
    bool fast = !(flags & MVHDP_SWEEP_GENERIC_KERNEL) && tu.force_primary != 32;
    int pc = 0;
    if (fast) {
        if (tu.force_primary >= 1 && tu.force_primary <= 16) {
            while ((1 << pc) < tu.force_primary) pc++;
        } else if (tot > 0) {
            double cum = 0;
            pc = std::min(c_max, 4);
            for (int c = 0; c <= std::min(c_max, 4); c++) {
                cum += tok[c];
                if (cum >= tu.primary_min_share * tot) { pc = c; break; }
            }
            if (cum < tu.primary_min_share * tot && c_max == 5) fast = false;          // nearly everything beyond 1024 slots: the generic kernel alone
        }
        pc = std::min(pc, std::min(c_max, 4));
    }
    // the launched class holding most tokens (the primary also takes every narrower list): its group's walk threshold is searched
    p.dominant = pc;
    {
        double best = 0;
        for (int q = 0; q <= pc; q++) best += tok[q];
        for (int c = pc + 1; c < MVHDP_N_CLASSES; c++) if (tok[c] > best) { best = tok[c]; p.dominant = c; }
    }

    // ---- geometry of every class that may run ----
    auto geometry = [&](bool is_fast, int c, int walk, ClassLaunch& g) -> bool {
        const int r = 1 << c;
        g.fast = is_fast; g.r = is_fast ? r : 0;
        g.S_cap = is_fast ? std::min(S_cap, 64 * r) : S_cap;
        g.wave_bytes = (uint32_t)(is_fast ? mvhdp_sweep_fast_wave_bytes(M, g.S_cap, r) : mvhdp_sweep_wave_bytes(M, S_cap));
        g.wpb = 4;
        while (g.wpb > 1 && p.block_shared_bytes + (size_t)g.wpb * g.wave_bytes > in.max_lds) g.wpb >>= 1;
        g.lds = p.block_shared_bytes + (size_t)g.wpb * g.wave_bytes;
        if (g.lds > in.max_lds) return false;
        const int regs = is_fast ? in.regs.regs[c][in.debug ? 2 : (walk ? 1 : 0)] : in.regs.regs[5][in.debug ? 2 : 0];
        const int bpc = plan_blocks_per_cu(regs, 64 * g.wpb, g.lds);
        const int64_t need = (in.D + (int64_t)g.wpb * MVHDP_DOC_BATCH - 1) / ((int64_t)g.wpb * MVHDP_DOC_BATCH);
        // (two segments in flight: one block per CU stays free for the updater's kernels and the next segment's first blocks)
        g.grid = (int)std::max<int64_t>(1, std::min<int64_t>(need, (int64_t)in.num_cus * ((p.overlap && bpc > 1) ? bpc - 1 : bpc)));
        if (tu.single_wave) { g.wpb = 1; g.lds = p.block_shared_bytes + g.wave_bytes; g.grid = 1; }
        return true;
    };

    // ---- walk thresholds of this sweep and the kernel flavours that go with them ----
    // which threshold group a class's kernel belongs to: decided by its flavour (below), which for class 0 depends on the mirror
    const bool mirror_ok = fast && !in.debug && tu.narrow != 0 && (!(flags & MVHDP_SWEEP_REUSE_TREES) || in.trees_current) &&
                           !((flags & MVHDP_SWEEP_SEGMENT_OVERLAP) && in.unassigned);   // (there the mirror follows by deltas: a row's weight class must not move)
    // (not while a token may be unassigned: its first visit only ADDS to its row, so a row classified light when its tree was built
    // could reach 65535 in a cell -- read as "see the 32-bit table" -- or carry into the neighbouring cell of the packed word)
    const bool want_live16 = mirror_ok && p.live && !p.frozen && !in.unassigned && (tu.live16 > 0 || (tu.live16 < 0 && K >= 256));
    auto group_of = [&](int c) { return c != 0 ? 2 : ((mirror_ok && (!p.live || want_live16)) ? 0 : 1); };
    double theta[WALK_GROUPS][MVHDP_MAXM] = {{0}};
    bool measure = false;
    if (tu.walk_fixed) {
        for (int g = 0; g < WALK_GROUPS; g++) for (int m = 0; m < MVHDP_MAXM; m++) theta[g][m] = tu.walk_theta[m];
        measure = true;
    } else wt.propose(fast ? group_of(p.dominant) : 2, M, theta, &measure);
    // The 16-bit mirror belongs to the walk flavour (NARROW is compiled for it only): where a row of the counts is long enough for its
    // lines to matter (K >= 256: at least 1 KiB a row) the walk flavour runs even when every threshold is 0 -- C5 (K = 1000, tree-branch
    // share 0.44: no view is steered) sweeps in 51.9 ms on the mirror against 64.4 ms on the 32-bit rows.
    const bool mirror_pays = mirror_ok && !p.live && K >= 256;
    auto walk_of = [&](int c) { const int g = group_of(c); bool w = measure || in.debug || mirror_pays; for (int m = 0; m < M; m++) w = w || theta[g][m] > 0.0; return w ? 1 : 0; };

    ClassLaunch gen;
    if (!geometry(false, 5, 0, gen)) return fail(MVHDP_ERR_UNSUPPORTED, "per-entity LDS state exceeds 160 KiB (K * modalities too large)");
    gen.used = false;
    if (fast) {
        ClassLaunch pg;
        if (!geometry(true, pc, walk_of(pc), pg)) fast = false;        // no room for the slot state of the primary variant: generic kernel
    }
    p.fast = fast;
    p.pc = fast ? pc : 5;
    // the 4-round class: a stream of its own where it carries weight (C5: a tenth of the tokens -- 42.5 against 43.4 ms a sweep, an 8-way
    // shard 6.85 against 7.9), behind the 2-round class on stream C where it is a handful of entities (C4: 0.7 % of the tokens -- beside
    // the others its few blocks only get in their way: an 8-way shard 3.66 against 3.85 ms); gpurun_out/r5e
    const bool four_on_c = tu.four_round_on_c > 0 || (tu.four_round_on_c < 0 && !unknown && tok[2] < 0.03 * tot);
    const int stream_of[MVHDP_N_CLASSES] = {PLAN_STREAM_C, PLAN_STREAM_C, four_on_c ? PLAN_STREAM_C : PLAN_STREAM_D, PLAN_STREAM_B, PLAN_STREAM_A, PLAN_STREAM_A};
    if (!fast) {
        p.cls[5] = gen; p.cls[5].used = true; p.cls[5].stream = PLAN_STREAM_MAIN;
        for (int c = 0; c < MVHDP_N_CLASSES; c++) p.class_map[c] = 5;
        p.route = false; p.H = 0;
        p.need_full = true;
    } else {
        // entities that may exceed the primary variant: more tokens than it has slots -- a static prefix of the longest-first order
        p.H = (S_cap > (64 << pc)) ? (in.have_order ? in.n_longer[pc] : in.D) : 0;
        if (view_16 && pc >= 3) p.H = in.have_order ? std::max<int64_t>(p.H, in.n_longer[4]) : in.D;   // (beyond 65535 tokens: beyond 1024 too)
        p.route = p.H > 0;
        for (int c = pc; c < MVHDP_N_CLASSES; c++) {
            const bool reachable = c == pc || (p.route && (c <= c_max || (c == 5 && view_16)));
            const bool populated = c == pc || unknown || in.ent_hist[c] != 0 || (c == 5 && view_16);
            if (!(reachable && populated)) continue;
            ClassLaunch g;
            const int w = walk_of(c);
            if (c < 5 && geometry(true, c, w, g)) { g.walk = w; }
            else { g = gen; g.walk = 0; }                               // no room for that variant's slot state (or class 5): the generic kernel
            g.used = true;
            g.stream = (c == pc || tu.single_stream || tu.single_wave) ? PLAN_STREAM_MAIN : stream_of[c];
            for (int m = 0; m < MVHDP_MAXM; m++) g.theta[m] = theta[group_of(c)][m];
            p.cls[c] = g;
        }
        if (!tu.single_stream && !tu.single_wave && tu.widest_on_main) {
            // widest first: the handle's stream, then side streams A, B, C in turn (the last one shared by whatever is left)
            int next = PLAN_STREAM_MAIN;
            for (int c = MVHDP_N_CLASSES - 1; c >= pc; c--) if (p.cls[c].used) { p.cls[c].stream = next; next = std::min(next + 1, (int)PLAN_STREAM_D); }
        }
        // a list of a class nobody launched goes to the next wider launched class (a wider variant holds narrower lists); the widest
        // reachable class is always launched when the sizes are not all known
        for (int c = 0; c < MVHDP_N_CLASSES; c++) {
            int t = std::max(c, pc);
            while (t < MVHDP_N_CLASSES && !p.cls[t].used) t++;
            p.class_map[c] = t < MVHDP_N_CLASSES ? t : -1;
        }
        // FTree.tree itself is read by the generic kernel (and the debug trace) only: when no entity can reach it, the rebuild
        // refreshes just the descent table (0.13 instead of 0.24 ms at C4)
        p.need_full = in.debug;
        for (int c = pc; c < MVHDP_N_CLASSES; c++) if (p.cls[c].used && !p.cls[c].fast) p.need_full = true;
        if (p.route && p.class_map[5] < 0 && (c_max == 5 || view_16)) {     // (cannot happen: class 5 is launched whenever it is reachable and not known empty)
            p.cls[5] = gen; p.cls[5].used = true; p.cls[5].stream = tu.single_stream ? PLAN_STREAM_MAIN : PLAN_STREAM_A; p.class_map[5] = 5; p.need_full = true;
        }
    }
    // the 16-bit mirror of n_wk (written with the trees) for the 1-round walk flavour -- the bandwidth-bound one: half the lines of
    // every gathered row.  Needs the mirror to be this sweep's start counts: trees built in this call or still current.  A live sweep
    // keeps the mirror current itself (packed 16-bit atomics, see sweep_fast_kernel's LIVE16 path).
    if (mirror_ok && !p.live && p.cls[0].used && p.cls[0].fast && p.cls[0].walk) p.cls[0].narrow = 1;
    if (mirror_ok && !p.live && tu.narrow_wide && tu.narrow != 1)                      // the wider variants too (since the row's weight class travels with the type id: 1 % on C4)
        for (int c = 1; c < 5; c++) if (p.cls[c].used && p.cls[c].fast && p.cls[c].walk) p.cls[c].narrow = 1;
    // A live sweep updates n_wk while it samples: the mirror stays usable only if the sweep's own atomics keep it current, which takes
    // every kernel of the sweep in the NARROW (hence walk) flavour -- no generic kernel among them.
    // The deltas of a plain deferred sweep (one segment, applied by this call) in 16-bit cells where the row allows it: the kernels of the
    // NARROW flavour know the row's class; whatever else runs in the sweep writes the 32-bit table as ever, the apply pass adds both.
    // Not while a token may be unassigned: the row's class is taken from its counts, and first visits only add.
    p.delta16 = tu.delta16 != 0 && mirror_ok && !p.live && !p.frozen && !p.seg_apply && nseg == 1 && p.only_seg < 0 &&
                !(flags & MVHDP_SWEEP_NO_APPLY) && !in.unassigned;                 // (a group of document shards sweeps with NO_APPLY: 32-bit deltas for the all-reduce)
    if (want_live16) {
        bool all_fast = true;
        for (int c = 0; c < MVHDP_N_CLASSES; c++) if (p.cls[c].used && !p.cls[c].fast) all_fast = false;
        if (all_fast) {
            p.live16 = true;
            for (int c = 0; c < MVHDP_N_CLASSES; c++) if (p.cls[c].used) { p.cls[c].walk = 1; p.cls[c].narrow = 1; }
        }
    }
    if (want_rows) {
        bool all_fast = p.fast;
        for (int c = 0; c < MVHDP_N_CLASSES; c++) if (p.cls[c].used && !p.cls[c].fast) all_fast = false;
        if (all_fast) {
            // every kernel the walk flavour (the tree branch lives in its walk-on-demand arm), nothing walked at the chunk head: u1 < 1 < theta
            // every kernel the walk flavour (the tree branch lives beside its walk-on-demand arm).  The view's threshold: the u1 from which a
            // token's row is loaded ahead of its turn (and a heavy word's stored tree walked at the chunk head) -- 0 where the tree branch
            // takes a third of the view's tokens or more (the short side views), else tu.live_rows_theta
            p.live_rows = true;
            p.need_full = false;
            for (int c = 0; c < MVHDP_N_CLASSES; c++)
                if (p.cls[c].used) {
                    p.cls[c].walk = 1;
                    for (int m = 0; m < MVHDP_MAXM; m++) p.cls[c].theta[m] = (m < M && wt.controlled(m)) ? tu.live_rows_theta : 0.0;
                }
        } else p.coef_lds = false;
    } else p.coef_lds = false;
    p.walk_cfg = (p.live_rows ? 1 << 20 : 0) + (p.fast ? (1 << p.dominant) : 32) * 2 + (p.route ? 1 : 0) + 64 * nseg + (p.live ? 1 << 16 : 0) + (p.seg_apply ? 1 << 17 : 0) + (p.live16 ? 1 << 18 : 0) + (p.overlap ? 1 << 19 : 0);
}
