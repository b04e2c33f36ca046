// mvhdp_sweep_fast.hip — the register-resident form of the sweep kernel for entities whose topic list
// fits 64*RMAX slots (RMAX = 1, 2, 4, 8, 16; longer lists take the generic LDS kernel of
// mvhdp_kernels.hip).  Same arithmetic, same order, same results as the generic kernel; what changes is
// where the per-entity state lives and when things are fetched:
//
//   slot i = lane*R + r  (R = 1 .. 16 slots per lane, R <= RMAX)  -> lane registers
//     koff[r] topic of the slot, as the byte offset of its count inside an n_wk row (topic * 4)
//     live_m[r]  (scalar registers, one bit per lane) the slot is on the list: cleared when the topic is removed (WRK:451-468),
//             never set for a slot the list does not reach -- a scalar mask, so a removal is scalar arithmetic and the token loop
//             selects on it without a comparison
//     cn[r]   localTopicCounts[m][topic] of the view being sampled (WRK:357,437,560)
//     oth[r]  totalMassOtherModalities[topic]                       (WRK:399-410)
//     den[r]  tokensPerTopic[m][topic] + betaSum[m]                 (WRK:507)
//     onz     bit r: some other view still holds the topic (for WRK:441-448)
//   per 64-token chunk, one lane per token ("chunk head"): type, old topic and its slot, the Philox draw, and the
//     whole F+tree descent of the token's type through the descent table (MvModel::dtab) -- none of it depends
//     on the entity's state, so 64 descents run side by side instead of one per tree-branch token
//   per token (mvhdp_sweep_fast_token.inc): the n_wk values of the listed topics, gathered one token ahead (two
//     for the wide variants, in two register buffers used in turn) -- legal because the sweep reads a snapshot
//     of n_wk, the deltas go to a separate buffer; topicDocWordMasses (WRK:511) = an in-lane running sum over
//     the lane's R slots plus ONE DPP prefix scan of the lane totals, never stored
//   per chunk end: the deltas of the chunk's tokens as two wave-wide atomics, z written back coalesced
//
// LDS per wave: bitmap + prefix + slot->topic + per-view slot counts (2.6 KB at K=400, M=3; 16-bit counts in
// the 8- and 16-round variants), so occupancy is bound by VGPRs, not LDS.
#include "mvhdp_device.h"
#include "../../include/mvhdp.h"
#include "mvhdp_wave.h"

// The 8- and 16-round variants keep the per-view slot counts as 16-bit values (its kernel diverts an entity with
// a view of more than 65535 tokens): 78 KB instead of 120 KB per block at K = 1000 with 5 views, i.e. the
// two blocks per CU its 256 VGPRs allow.
size_t mvhdp_sweep_fast_wave_bytes(int M, int S_cap, int rmax)
{
    size_t counts = (size_t)M * S_cap * (rmax >= 8 ? 2 : 4);
    size_t b = (size_t)(64 + 64 + 64 + 16 + S_cap) * 4 + counts;      // bitmap, prefix, bitmap of the new assignments, wlen, sk
    b += (size_t)M * M * 12;                                           // per-entity view-pair constants: M*M doubles (qsh) + M*M floats (wsh)
    return (b + 15) & ~(size_t)15;
}

// Minimum waves per SIMD asked of the register allocator for each variant (measured on C4; the 2-slot variant lost 15 % at 7 waves
// in round 2, when 72 registers meant heavy spilling -- with the round-3 token loop it gains 1.6 % there over 6).
#ifndef MVHDP_LB1
#define MVHDP_LB1 1          // the plain 1-slot variant allocates 70 VGPRs by itself (7 waves); bounding it costs 2 % on C2
#endif
#ifndef MVHDP_LB1W
#define MVHDP_LB1W 7         // with the thresholded walk it would take 75 (6 waves): C3 loses 4 % there
#endif
#ifndef MVHDP_LB2
#define MVHDP_LB2 7
#endif
#ifndef MVHDP_LB2_ROOMY
#define MVHDP_LB2_ROOMY 6
#endif
#ifndef MVHDP_LB4
#define MVHDP_LB4 4
#endif
#ifndef MVHDP_NB2_FROM
#define MVHDP_NB2_FROM 4     // variants with at least this many slots per lane gather two tokens ahead into two register buffers
#endif
#ifndef MVHDP_LB16
#define MVHDP_LB16 1
#endif
#ifndef MVHDP_LB_ROWS
#define MVHDP_LB_ROWS 7      // the live-rows flavours of the 1- and 2-round variants (four more registers: the token's row)
#endif
#ifndef MVHDP_LB8
#define MVHDP_LB8 1          // 3 waves/SIMD (168 VGPRs) spills 100 B/lane and is 13 % slower on C5 than 2 waves at 199
#endif
// WALK: the flavour with the thresholded tree walk of the chunk head (SweepLaunch::walk_theta), the walk on demand in the token
// loop and the per-view branch statistics the threshold search feeds on.  Without it every token is walked up front and nothing
// is counted: where the best threshold is 0 (C2, C3) that code is 2-5 % faster for not carrying the rest.
// NARROW (walk flavour only): the n_wk gather reads the 16-bit mirror of the counts (MvModel::counts16, written with the trees at
// the start of the sweep) -- half the lines of the row -- for a light row, and the 32-bit table for a heavy one (MvModel::heavy: a
// type with more than 65534 tokens, whose mirror cells all read 65535); the row's class travels with the token's type id (W_HEAVY),
// so the choice is a scalar branch at the gather and the values need no check when they are used.  Same numbers, so same results.  A deferred sweep uses it for the 1-round variant only (the 2-round variant is 4 %
// slower with it: two 2-byte loads per lane cost it more than the lines are worth).  A live sweep (SweepLaunch::live16) uses it for
// every variant: there the chunk-end atomics of the light rows land IN the mirror -- two 16-bit cells per 32-bit word, +-1 or +-65536,
// which cannot carry: a light row's cell stays below 65535 and a decrement only ever takes back a token that was counted -- so the
// mirror is what every later token of the sweep reads (UPD:197-207 applied while the workers sample), at half the gather traffic.
// How a live sweep's gathers see the other waves' atomics (diagnostics; tools/microbench/live_staleness.hip):
// MVHDP_GATHER_SC = 0 plain loads, 1 agent scope (global_load ... sc1: past the CU's L1), 3 system scope (sc0 sc1), 4 non-temporal
#ifndef MVHDP_GATHER_SC
#define MVHDP_GATHER_SC 0
#endif
template <typename T>
__device__ __forceinline__ int gather_cell(gptr_t p)
{
    const __attribute__((address_space(1))) T* q = (const __attribute__((address_space(1))) T*)p;
#if MVHDP_GATHER_SC == 1
    return (int)__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#elif MVHDP_GATHER_SC == 3
    return (int)__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#elif MVHDP_GATHER_SC == 4
    return (int)__builtin_nontemporal_load(q);
#else
    return (int)*q;
#endif
}
#define W_HEAVY 0x40000000                      // bit 30 of a lane's type id: the row is heavy (type ids stay below 2^29: mvhdp_create checks)
#define W_BIG   0x20000000                      // bit 29: the row's deltas do not fit 16 bits for sure (MVHDP_ROW_BIG or heavy): they go to the 32-bit delta table
#define W_ROW(w) ((w) & 0x1fffffff)
// The tree branch of a live sweep in its live-rows form (SweepLaunch::live_rows; WRK:533-535 against what UPD:242-260 keeps current):
// a topic with probability proportional to leaf_k = coef_k * (n_wk + beta) over ALL K topics, from the word's LIVE row.  The leaf splits
// into a smoothing part coef_k * beta -- the same for every word of the view: its running sums smp[k] are a table of the segment -- and a
// count part coef_k * n_wk, zero wherever the word has no token.  The target u2 * tree[1] falls into the smoothing part with probability
// S / tree[1] -- small for any word with more than a handful of tokens -- and is searched in the table; otherwise in the row, which the
// wave holds in registers: ONE 16-byte load per lane covers 512 cells of the 16-bit mirror (256 of the 32-bit table), lane l holding
// the cells CPL*l .. CPL*l + CPL-1 in topic order; every lane sums its cells times their coefficients (from LDS), ONE DPP scan of the lane
// sums finds the lane whose range holds the target, and that lane's running sums find the cell.
// What a tree-branch token costs the wave is LATENCY (a wave samples its tokens one after the other), and a vector load issued once the
// branch is known waits, in the in-order vmcnt counter, behind the next token's gather that is already in flight -- 3800 cycles per
// tree-branch token, measured (-DMVHDP_TIMING), whatever the arithmetic behind it.  So the row is loaded at the TOP of the token's
// body, before that gather is issued, for every token whose u1 reaches the view's threshold (only a large u1 can reach the tree branch:
// the walk threshold of the other flavours, used the same way) -- the lines are in the L2 already, the token's own gather has just
// brought them -- and the coefficients come from LDS (lgkmcnt): nothing on the tree branch's path waits for the memory system.
// fp32 throughout: a live sweep is not reproducible run to run (the other waves' atomics land while it samples), so there is no fp64
// sequence to certify against; with ONE resident wave the arithmetic below IS a definition, which the oracle restates operation for
// operation (the sequential-live mode of the test oracle).  A target beyond the row's mass (tree[1] is the segment start's; rounding)
// takes the last topic that has mass.
typedef unsigned int rowq_t __attribute__((ext_vector_type(4), aligned(4)));     // (a row of K cells need not start on a 16-byte boundary)
typedef float coefq_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int smoothing_sample_live(const float* __restrict__ smp, int K, float t, int lane)
{
    for (int k0 = 0; k0 < K; k0 += WAVE) {
        const int k = k0 + lane;
        const float p = (k < K) ? smp[k] : 3.0e38f;
        const unsigned long long hit = __builtin_amdgcn_ballot_w64(p > t && k < K);
        if (hit) return k0 + (int)__builtin_ctzll(hit);
    }
    return K - 1;
}

// this lane's cells of batch b of a row (CELL16: eight 16-bit cells of the mirror, else four 32-bit cells), zero beyond the row's end
template <bool CELL16>
__device__ __forceinline__ rowq_t row_batch_load(gptr_t rowp, int K, int b, int lane)
{
    constexpr int CPL = CELL16 ? 8 : 4;
    const int k = (b * WAVE + lane) * CPL;
    rowq_t q = {0u, 0u, 0u, 0u};
    if (k < K) q = *(const __attribute__((address_space(1))) rowq_t*)(rowp + (unsigned int)k * (CELL16 ? 2u : 4u));
    return q;
}

// The batch in registers: the topic whose range holds `target` (>= 0), or -1 when the batch's mass ends below it; force: the last
// cell with mass, whatever the target.  c0, c1: the coefficients of this lane's cells (zero beyond K, so that whatever the load brought
// from beyond the row's end counts for nothing).  tot: the batch's mass; any: some cell of it has mass.
// What this costs is instructions (seven waves share a SIMD: a wave gets an issue slot every twenty cycles or so), so the lane's cells
// go two at a time: the products and their running sums are NS = CPL/2 packed fused multiply-adds (v_pk_fma_f32), the even cells in one
// half, the odd cells in the other -- the lane's cells are therefore taken in the order even ones first (0, 2, 4, 6), then odd ones
// (1, 3, 5, 7): any fixed order of the K leaves samples the same distribution --, and the cell inside the chosen lane is the NUMBER of
// running sums that do not pass the target (they never decrease): a compare and an add-with-carry each, no select.
typedef float f2_t __attribute__((ext_vector_type(2)));
template <bool CELL16>
__device__ __forceinline__ int row_batch_pick(rowq_t q, coefq_t c0, coefq_t c1, int kb, float base, float target, bool force, float& tot, bool& any)
{
    constexpr int CPL = CELL16 ? 8 : 4, NS = CPL / 2;
    f2_t n2[NS], k2[NS], A[NS];
    if (CELL16) {
        n2[0] = f2_t{(float)(q.x & 0xffffu), (float)(q.x >> 16)}; n2[1] = f2_t{(float)(q.y & 0xffffu), (float)(q.y >> 16)};
        n2[NS - 2] = f2_t{(float)(q.z & 0xffffu), (float)(q.z >> 16)}; n2[NS - 1] = f2_t{(float)(q.w & 0xffffu), (float)(q.w >> 16)};
        k2[0] = f2_t{c0.x, c0.y}; k2[1] = f2_t{c0.z, c0.w}; k2[NS - 2] = f2_t{c1.x, c1.y}; k2[NS - 1] = f2_t{c1.z, c1.w};
    } else {
        n2[0] = f2_t{(float)(int)q.x, (float)(int)q.y}; n2[NS - 1] = f2_t{(float)(int)q.z, (float)(int)q.w};
        k2[0] = f2_t{c0.x, c0.y}; k2[NS - 1] = f2_t{c0.z, c0.w};
    }
    A[0] = n2[0] * k2[0];
#pragma unroll
    for (int s = 1; s < NS; s++) A[s] = __builtin_elementwise_fma(n2[s], k2[s], A[s - 1]);
    const float ev = A[NS - 1].x;                                          // the even cells' sum; A[s].y: the running sums of the odd ones
    const float acc = ev + A[NS - 1].y;
    const float incl = wave_incl_scan_f_dpp(acc);
    tot = bcast_f(incl, 63);
    const unsigned long long pos = __builtin_amdgcn_ballot_w64(acc > 0.0f);
    any = pos != 0;
    if (!pos || !(force || base + tot > target)) return -1;
    const unsigned long long hit = force ? 0ull : (__builtin_amdgcn_ballot_w64(base + incl > target) & pos);
    const int hl = hit ? (int)__builtin_ctzll(hit) : 63 - (int)__builtin_clzll(pos);
    const float thr = target - (base + (hl > 0 ? bcast_f(incl, hl - 1) : 0.0f));
    int cnt = 0;
#pragma unroll
    for (int s = 0; s < NS; s++) cnt += (A[s].x <= thr) ? 1 : 0;
#pragma unroll
    for (int s = 0; s < NS; s++) cnt += (ev + A[s].y <= thr) ? 1 : 0;
    int p = force ? CPL : bcast_i(cnt, hl);
    if (p >= CPL) {
        // every running sum of the lane stays at or below the target (rounding; or `force`): the last cell of the lane, in its order, that has mass
        int lp = 0;
#pragma unroll
        for (int s = 0; s < NS; s++) if (n2[s].x > 0.0f && k2[s].x > 0.0f) lp = s;
#pragma unroll
        for (int s = 0; s < NS; s++) if (n2[s].y > 0.0f && k2[s].y > 0.0f) lp = NS + s;
        p = bcast_i(lp, hl);
    }
    return kb + hl * CPL + (p < NS ? 2 * p : 2 * (p - NS) + 1);
}

// in_lds: the view's coefficients (row padded to a multiple of 8 with zeros) come from LDS (cf_lds), else from global memory (cf_g)
// A row of TWO batches (K above 512 cells of the mirror, 256 of the 32-bit table: C5) would need its second batch on demand -- a load behind
// the next token's gather, 6400 cycles per tree-branch token at C5 --, so the FIRST batch's mass of the segment start is stored with the
// root (MvModel::mass0): a target at or beyond it starts in the second batch with that mass as its base (`bq` = the batch chosen that way,
// decided at the chunk head; q0 holds THAT batch when the token's row was loaded ahead).  Like the root itself the stored mass is not
// followed during the segment.  When the chosen batch does not hold the target after all (the live masses have moved), the batches are
// scanned in order with their live masses, as rows of any other length are.
template <bool CELL16, bool TWOB>
__device__ __forceinline__ int row_sample_live(rowq_t q0, bool have_q0, bool in_lds, const __attribute__((address_space(3))) float* cf_lds, const float* __restrict__ cf_g, const float* __restrict__ smp, float S,
                                                gptr_t rowp, int K, float u2f, float rootf, int bq, float mass0, int lane, unsigned long long* t_after_wait)
{
    constexpr int CPL = CELL16 ? 8 : 4;
    (void)t_after_wait;
    float target = u2f * rootf;
    if (target < S) return smoothing_sample_live(smp, K, target, lane);
    target -= S;
    float base = 0.0f;
    int lastb = -1;
    const int nb = (K + WAVE * CPL - 1) / (WAVE * CPL);
    if (!TWOB || nb != 2) bq = 0;                                          // (TWOB: the kernel flavour compiled for rows of two batches; the others scan in order)
    const float bbase = bq ? mass0 : 0.0f;
    if (have_q0 && in_lds) {
        // The common path, kept apart from the general loop below: row in registers since the top of the token's turn, coefficients from
        // LDS -- no vector-memory operation is issued here, so the wait in front of the arithmetic is for the row alone (vmcnt counts in
        // order: a load issued on ANY path through here would make the compiler wait for everything, the next token's gather included).
        int kl = (bq * WAVE + lane) * CPL;
        asm volatile("" : "+v"(kl));                                       // (formed here: hoisted out of the token loop it is spilled, and its reload is a vector-memory operation)
        coefq_t c0 = {0.0f, 0.0f, 0.0f, 0.0f}, c1 = {0.0f, 0.0f, 0.0f, 0.0f};
        if (kl < K) {                                                      // (a lane beyond the row has no cells: the view's table ends at Kp, the last view's at the end of the block's)
            c0 = *(const __attribute__((address_space(3))) coefq_t*)(cf_lds + kl);
            if (CELL16) c1 = *(const __attribute__((address_space(3))) coefq_t*)(cf_lds + kl + 4);
        }
        float tot; bool any;
#ifdef MVHDP_TIMING
        { unsigned int probe = q0.x; asm volatile("v_mov_b32 %0, %0" : "+v"(probe)); *t_after_wait = __builtin_amdgcn_s_memtime(); }   // (the row has arrived)
#endif
        const int r = row_batch_pick<CELL16>(q0, c0, c1, bq * WAVE * CPL, bbase, target, false, tot, any);
        if (r >= 0) return r;
        if (nb == 1) {
            if (any) return row_batch_pick<CELL16>(q0, c0, c1, 0, 0.0f, 0.0f, true, tot, any);   // the target lies beyond the row's mass: the last cell that has any
            return smoothing_sample_live(smp, K, u2f * S, lane);          // (a word without a counted token: a first visit of an unassigned one)
        }
    } else if (TWOB && nb == 2) {
        // the same first try for a token whose row was not loaded ahead (or whose coefficients are not in LDS): the chosen batch, its stored base
        const rowq_t q = have_q0 ? q0 : row_batch_load<CELL16>(rowp, K, bq, lane);
        const int kl = (bq * WAVE + lane) * CPL;
        coefq_t c0 = {0.0f, 0.0f, 0.0f, 0.0f}, c1 = {0.0f, 0.0f, 0.0f, 0.0f};
        if (kl >= K) { }
        else if (in_lds) {
            c0 = *(const __attribute__((address_space(3))) coefq_t*)(cf_lds + kl);
            if (CELL16) c1 = *(const __attribute__((address_space(3))) coefq_t*)(cf_lds + kl + 4);
        } else {
            c0 = *(const __attribute__((address_space(1))) coefq_t*)(cf_g + kl);
            if (CELL16) c1 = *(const __attribute__((address_space(1))) coefq_t*)(cf_g + kl + 4);
        }
        float tot; bool any;
        const int r = row_batch_pick<CELL16>(q, c0, c1, bq * WAVE * CPL, bbase, target, false, tot, any);
        if (r >= 0) return r;
    }
    for (int pass = 0; pass < 2; pass++) {                                  // (pass 1: the target lies beyond the row's mass -- the last cell that has any)
        for (int b = (pass ? lastb : 0); b < (pass ? lastb + 1 : nb); b++) {
            const rowq_t q = (b == bq && have_q0) ? q0 : row_batch_load<CELL16>(rowp, K, b, lane);
            const int kl = (b * WAVE + lane) * CPL;                         // (the view's table is padded to Kp = K rounded up to 8: the lane that holds cell K-1 reads zeros behind it)
            coefq_t c0 = {0.0f, 0.0f, 0.0f, 0.0f}, c1 = {0.0f, 0.0f, 0.0f, 0.0f};
            if (kl >= K) { }                                                // (a lane beyond the row has no cells)
            else if (in_lds) {                                              // (ds_read_b128: counted by lgkmcnt, never behind a vector-memory operation)
                c0 = *(const __attribute__((address_space(3))) coefq_t*)(cf_lds + kl);
                if (CELL16) c1 = *(const __attribute__((address_space(3))) coefq_t*)(cf_lds + kl + 4);
            } else {
                c0 = *(const __attribute__((address_space(1))) coefq_t*)(cf_g + kl);
                if (CELL16) c1 = *(const __attribute__((address_space(1))) coefq_t*)(cf_g + kl + 4);
            }
            float tot; bool any;
            const int r = row_batch_pick<CELL16>(q, c0, c1, b * WAVE * CPL, base, target, pass != 0, tot, any);
            if (r >= 0) return r;
            if (any) lastb = b;
            base += tot;
        }
        if (lastb < 0) break;
    }
    return smoothing_sample_live(smp, K, u2f * S, lane);                  // (a word without a counted token: a first visit of an unassigned one)
}

// ROOMY (the 2-round variant on the mirror only): the same kernel compiled for 6 waves per SIMD (80 registers, a third of the scratch
// of the 72-register build): where a row of the mirror is 1 KiB or more (K >= 512; C5: K = 1000) the seventh wave hides less than the
// spills cost -- C5's 2-round kernel 16.9 ms at 6 waves, 18.4 at 7; C4's (K = 400) gains 2 % at 7.
// LIVEROWS (walk flavour only): the live-rows form of a live sweep (SweepLaunch::live_rows) -- a flavour of its own so that the kernels of
// every other mode stay what they were, register for register.
// LIVEROWS = 2: the same with the two-batch shortcut of row_sample_live compiled in (rows of the mirror longer than one register batch,
// K in 513 .. 1024: C5) -- a flavour of its own because its few extra scalars cost the K <= 512 kernels 5 % through their spills (C4: 24.1 -> 25.4 ms)
template <int RMAX, bool DEBUG, bool WALK, bool NARROW, bool ROOMY = false, int LIVEROWS = 0>
__global__ __launch_bounds__(256, (RMAX == 8 ? MVHDP_LB8 : (RMAX == 4 ? MVHDP_LB4 : (RMAX == 2 ? (ROOMY ? MVHDP_LB2_ROOMY : (LIVEROWS ? MVHDP_LB_ROWS : MVHDP_LB2)) : (RMAX == 1 ? (LIVEROWS ? MVHDP_LB_ROWS : WALK ? MVHDP_LB1W : MVHDP_LB1) : MVHDP_LB16))))) void sweep_fast_kernel(MvModel mm, SweepLaunch sl)
{
    extern __shared__ __align__(16) unsigned char smem[];
#ifdef MVHDP_TIMING
    const unsigned long long t_begin0 = __builtin_amdgcn_s_memtime();
#endif
    const int lane = threadIdx.x & 63;
    const int wave = uniform_i(threadIdx.x >> 6);
    const int K = mm.K, M = mm.M, S = sl.S_cap;
    const int NW = (K + 31) >> 5;
    const bool exact_only = (sl.flags & MVHDP_SWEEP_EXACT_CHAIN) != 0;
    // the wide variants hold the longest entities -- the sweep's critical path -- and share their SIMDs
    // with the bulk kernel's waves: let the arbiter issue them first
    if (RMAX >= 8) __builtin_amdgcn_s_setprio(3);

    // [M*K] n_k deltas of this block, privatised in LDS -- or none, the deltas going straight to the delta buffer,
    // when M*K is too large for that (sl.nk_global); then [MVHDP_HIST_BINS] tokens by topic-list size class
    const int nkd_len = sl.nk_global ? 0 : M * K;
    int* nkd = (int*)smem;
    unsigned int* hist_s = (unsigned int*)(nkd + nkd_len);
    unsigned int* ent_s = hist_s + MVHDP_HIST_BINS;       // [MVHDP_ENT_BINS] entities by the kernel class of their NEW topic list
    unsigned int* vstat_s = ent_s + MVHDP_ENT_BINS;       // [MVHDP_MAXM][MVHDP_VIEW_STATS] per-view branch statistics of this block
    for (int i = threadIdx.x; i < nkd_len + MVHDP_HIST_BINS + MVHDP_ENT_BINS + (WALK ? MVHDP_MAXM * MVHDP_VIEW_STATS : 0); i += blockDim.x) nkd[i] = 0;
    // live-rows form: the coefficient table of the segment (MvModel::coef, [M][Kp], Kp = K rounded up to 8, zero-padded) in LDS where the
    // plan found room for it (SweepLaunch::coef_lds): the tree branch reads its coefficients without a vector-memory operation
    const int Kp = (K + 7) & ~7;
    float* const coef_s = (LIVEROWS && sl.coef_lds) ? (float*)(((uintptr_t)(vstat_s + MVHDP_MAXM * MVHDP_VIEW_STATS) + 15) & ~(uintptr_t)15) : nullptr;
    if (LIVEROWS && coef_s) for (int i = threadIdx.x; i < M * Kp; i += blockDim.x) coef_s[i] = mm.coef[i];
    int32_t* const dnk_g = mm.delta + mm.rowbase[M] * K;    // n_k part of the delta buffer
    __syncthreads();
#ifdef MVHDP_TIMING
    const unsigned long long t_init_end = __builtin_amdgcn_s_memtime();
#endif

    unsigned char* wb = smem + sl.block_shared_bytes + (size_t)wave * sl.wave_bytes;
    uint32_t* bitmap = (uint32_t*)wb;
    uint32_t* prefix = bitmap + 64;
    uint32_t* bitmap2 = prefix + 64;                       // topics of the entity's NEW assignments (MvModel::nslots)
    int* wlen = (int*)(bitmap2 + 64);
    // per-entity constants of the view pairs (m, j), formed once per entity with one lane per pair instead of once per view and slot:
    //   qsh[m*M+j] = p[m][j] * gamma_j * alpha_j[K] / (len_j + gamma_j * alphaSum_j)     the terms of WRK:413-418, in fp64
    //   wsh[m*M+j] = p[m][j] / (len_j + gamma_j * alphaSum_j)  (0 for j == m or an empty view)   the weights of WRK:399-410, rounded to
    //                fp32: what the fp32 screening forms totalMassOtherModalities from (the fp64 path forms it exactly, slot_consts)
    double* qsh = (double*)(wlen + 16);
    float* wsh = (float*)(qsh + M * M);
    int* sk = (int*)(wsh + M * M);
    constexpr bool PACK = RMAX >= 8;                       // per-view slot counts as 16-bit values
    // fp32 screening of the token loop's decisions (mvhdp_sweep_fast_token.inc): two more registers per slot, so not for the
    // 8- and 16-round variants, which sit at their register limit; the debug flavour reports fp64 masses and decides in fp64
    constexpr bool SCREEN = !DEBUG;
    int* sn = sk + S;
#define sn_get(idx) (PACK ? (int)((const unsigned short*)sn)[(idx)] : sn[(idx)])
#define sn_set(idx, v) do { if (PACK) ((unsigned short*)sn)[(idx)] = (unsigned short)(v); else sn[(idx)] = (v); } while (0)

    // no __restrict__: with MVHDP_SWEEP_LIVE the atomics below update this very array (mm.delta == mm.counts)
    const int32_t* nwk = mm.counts;
    const uint16_t* nwk16 = mm.counts16;
    const int32_t* nk_all = mm.counts + mm.rowbase[M] * K;
    int32_t* dnwk = mm.delta;

    unsigned int n_tok = 0, n_chg = 0, c_new = 0, c_tree = 0, n_oov = 0, n_abort = 0, n_fb = 0, n_od = 0;

    // work queue: each wave pulls MVHDP_DOC_BATCH entities at a time from one global head
    const long long q_n1 = sl.q_list_count ? (long long)*sl.q_list_count : 0;
    const long long q_total = q_n1 + sl.q_order_count;
    unsigned int n_misclass = 0;
#ifdef MVHDP_TIMING
    // diagnostics: where a wave's cycles go (s_memtime stamps at the segment borders; the stamps cost ~10 %)
    unsigned long long tq = 0, tp = 0, tv = 0, th = 0, tt = 0, te = 0, t_ent[3] = {0, 0, 0}, n_ent[3] = {0, 0, 0}, t_rows = 0, t_rows_wait = 0;
    int e_ord = 0;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    unsigned long long t_last = t_begin;
#define MVHDP_TSEG(acc) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); (acc) += now_ - t_last; t_last = now_; } while (0)
#if defined(MVHDP_TSPLIT) && MVHDP_TSPLIT == 1   /* the prologue alone, cut in four (reported as prologue / view setup / chunk head / tokens), the rest as chunk end */
#define MVHDP_TSUB(acc) MVHDP_TSEG(acc)
#define MVHDP_TSUB2(acc) do { } while (0)
#define MVHDP_TMAIN(acc, s1, s2) MVHDP_TSEG(s1)
#elif defined(MVHDP_TSPLIT) && MVHDP_TSPLIT == 2 /* the chunk head alone, cut in four, the rest as chunk end */
#define MVHDP_TSUB(acc) do { } while (0)
#define MVHDP_TSUB2(acc) MVHDP_TSEG(acc)
#define MVHDP_TMAIN(acc, s1, s2) MVHDP_TSEG(s2)
#else
#define MVHDP_TSUB(acc) do { } while (0)
#define MVHDP_TSUB2(acc) do { } while (0)
#define MVHDP_TMAIN(acc, s1, s2) MVHDP_TSEG(acc)
#endif
#else
#define MVHDP_TSEG(acc) do { } while (0)
#define MVHDP_TSUB(acc) do { } while (0)
#define MVHDP_TSUB2(acc) do { } while (0)
#define MVHDP_TMAIN(acc, s1, s2) do { } while (0)
#endif
    // (the last pulls of the queue take one entity at a time: the launch ends within one entity's time of its last pull)
    const long long q_single = q_total - 2LL * gridDim.x * (blockDim.x >> 6);
    long long q_seen = 0;
    for (;;) {
      const long long batch = (q_seen >= q_single) ? 1 : MVHDP_DOC_BATCH;    // (4 or 8 per pull far from the end: no gain, gpurun_out/r4v)
      long long q0 = 0;
      if (lane == 0) q0 = (long long)atomicAdd(sl.doc_counter, (unsigned long long)batch);
      q0 = ((long long)__builtin_amdgcn_readfirstlane((int)(q0 >> 32)) << 32) | (unsigned int)__builtin_amdgcn_readfirstlane((int)q0);
      if (q0 >= q_total) break;
      q_seen = q0;
      const long long q1 = (q0 + batch < q_total) ? q0 + batch : q_total;
      for (long long q = q0; q < q1; q++) {
        int64_t d;
        if (q < q_n1) d = (int64_t)sl.q_list[q];
        else { const int64_t o = sl.q_order_start + (q - q_n1) * sl.q_order_stride; d = sl.q_order ? (int64_t)sl.q_order[o] : o; }
        const int64_t dg = mm.doc_id_base + d;
#ifdef MVHDP_TIMING
        const unsigned long long t_e0 = __builtin_amdgcn_s_memtime();
        const unsigned int n_tok_e0 = n_tok;
#endif

        MVHDP_TSEG(tq);
        // ---- WRK:339-391: gather the entity's topics into the slot list ----
        bitmap[lane] = 0;
        bitmap2[lane] = 0;
        LDS_FENCE();
        int doc_tokens = 0, longest_view = 0;
        for (int m = 0; m < M; m++) {
            const int64_t b = mm.doc_off[m][d], e = mm.doc_off[m][d + 1];
            doc_tokens += (int)(e - b);
            longest_view = max(longest_view, (int)(e - b));
            if (lane == 0) wlen[m] = (int)(e - b);
            for (int64_t i = b + lane; i < e; i += WAVE) {
                int zz = mm.z[m][i];
                if (zz >= 0) atomicOr(&bitmap[zz >> 5], 1u << (zz & 31));
            }
        }
        LDS_FENCE();
        MVHDP_TSUB(tp);
        int S_used;
        {
            uint32_t wbits = (lane < NW) ? bitmap[lane] : 0u;
            int cnt = __popc(wbits);
            int incl = wave_incl_scan_i(cnt, lane);
            prefix[lane] = (uint32_t)(incl - cnt);
            S_used = bcast_i(incl, 63);
        }
        LDS_FENCE();
        // Slots per lane: RMAX consecutive slots (slot i = lane*RMAX + r), whatever the list's size: a list shorter than the variant's
        // 64*RMAX slots simply occupies fewer lanes -- the per-token work is RMAX rounds either way -- and the shifts and masks by
        // R_eff in the token loop are compile-time constants (round 3; it used to be the smallest power of two that holds the list).
        constexpr int lg = (RMAX == 1) ? 0 : (RMAX == 2) ? 1 : (RMAX == 4) ? 2 : (RMAX == 8) ? 3 : 4;
        constexpr int R_eff = RMAX;
        if (S_used > 64 * RMAX || (PACK && longest_view > 65535)) {   // too many topics (or tokens) for this variant:
            n_misclass++;                                                        // route_kernel sent it here by mistake; the host fails the sweep
            continue;
        }
        for (int k0 = 0; k0 < K; k0 += WAVE) {
            int k = k0 + lane;
            if (k < K) {
                uint32_t w = bitmap[k >> 5];
                if ((w >> (k & 31)) & 1u) sk[prefix[k >> 5] + __popc(w & ((1u << (k & 31)) - 1u))] = k;
            }
        }
        for (int m = 0; m < M; m++)
            for (int i = lane; i < S_used; i += WAVE) sn_set(m * S + i, 0);
        LDS_FENCE();
        MVHDP_TSUB(tv);
        for (int m = 0; m < M; m++) {
            const int64_t b = mm.doc_off[m][d], e = mm.doc_off[m][d + 1];
            for (int64_t i = b + lane; i < e; i += WAVE) {
                int zz = mm.z[m][i];
                if (zz >= 0) {
                    uint32_t w = bitmap[zz >> 5];
                    int slot = prefix[zz >> 5] + __popc(w & ((1u << (zz & 31)) - 1u));
                    if (PACK) { const int idx = m * S + slot; atomicAdd((unsigned int*)&sn[idx >> 1], 1u << ((idx & 1) * 16)); }
                    else atomicAdd(&sn[m * S + slot], 1);                  // WRK:357
                }
            }
        }
        LDS_FENCE();
        MVHDP_TSUB(th);
        int koff[RMAX];
        unsigned long long live_m[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; r++) {
            const int i = lane * R_eff + r;
            const bool used = r < R_eff && i < S_used;
            koff[r] = used ? sk[i] << 2 : 0;                               // byte offset of the topic inside an n_wk row (unused slot: topic 0, never live)
            live_m[r] = __builtin_amdgcn_ballot_w64(used);
        }

        const double* pd = (M > 1) ? (mm.p + d * M * M) : nullptr;        // WRK:327-337
        if (lane < M * M) {                                               // (M <= 8: one lane per view pair)
            const int pm = lane / M, pj = lane - pm * M;
            const double dd = (double)wlen[pj] + mm.gamma[pj] * mm.alpha_sum[pj];
            const double pmj = pd ? pd[lane] : 1.0;
            qsh[lane] = pmj * (mm.gamma[pj] * mm.alpha[(int64_t)pj * (K + 1) + K]) / dd;          // WRK:416, operation for operation
            wsh[lane] = (pj != pm && wlen[pj] != 0) ? (float)(pmj / dd) : 0.0f;
        }
        LDS_FENCE();
        MVHDP_TMAIN(tp, tt, te);
        bool aborted = false;

        for (int m = 0; m < M && !aborted; m++) {                         // WRK:393
            const int lenm = uniform_i(wlen[m]);
            if (lenm == 0) continue;
            const double beta_m = mm.beta[m];
            const double scale_m = (double)lenm + mm.gamma[m] * mm.alpha_sum[m];
            const double p_mm = pd ? pd[m * M + m] : 1.0;
            const int32_t* nk = nk_all + (int64_t)m * K;

            // per-view slot registers; WRK:395-410 totalMassOtherModalities (frozen for this view, Q3)
            int cn[RMAX];
            double oth[RMAX], den[RMAX];          // (kept in registers by the flavours without screening; see slot_consts)
            unsigned int onz = 0;
            // oth and den of slot r of this lane, from what does not change while the view is sampled (the other views' counts of the
            // entity, the view lengths, n_k): the flavours with fp32 screening keep only the fp32 forms in registers and call this again
            // for the rare token that goes to fp64 -- same inputs, same operations, same doubles.
            auto slot_consts = [&](int r, double& o, double& dn) {
                o = 0.0; dn = 1.0;
                const int i = lane * R_eff + r;
                if (r < R_eff && i < S_used) {
                    const int k = koff[r] >> 2;
                    double acc = 0.0;
                    for (int j = 0; j < M; j++) {
                        if (j == m) continue;
                        const int cj = sn_get(j * S + i);
                        const int lj = wlen[j];
                        if (lj != 0)
                            acc += pd[m * M + j] * ((double)cj + mm.gamma[j] * mm.alpha[(int64_t)j * (K + 1) + k])
                                   / ((double)lj + mm.gamma[j] * mm.alpha_sum[j]);
                    }
                    o = acc * scale_m;
                    dn = (double)nk[k] + mm.beta_sum[m];
                }
            };
            float rden32[RMAX], brden32[RMAX], oth32[RMAX];      // the fp32 forms the screening works with: 1/den, beta/den, oth
            const float beta32 = (float)beta_m, scale32 = (float)scale_m;
            const float* const smp_m = mm.coef + (int64_t)M * Kp + (int64_t)m * K;                   // running sums of the smoothing parts coef_k * beta of the view's leaves
            const float smS = LIVEROWS ? uniform_f(smp_m[K - 1]) : 0.0f;                             // S_m: their total
#pragma unroll
            for (int r = 0; r < RMAX; r++) {
                cn[r] = 0;
                const int i = lane * R_eff + r;
                const bool used = r < R_eff && i < S_used;
                if (SCREEN) {
                    // fp32 forms directly: a division per slot and other view (30 instructions) becomes a multiply-add with the pair's
                    // weight -- on a corpus whose side views hold a handful of tokens (C5) this setup was a fifth of the wave's time
                    float acc = 0.0f;
                    double dn = 1.0;
                    if (used) {
                        const int k = koff[r] >> 2;
                        cn[r] = sn_get(m * S + i);
                        for (int j = 0; j < M; j++) {
                            if (j == m) continue;
                            const int cj = sn_get(j * S + i);
                            if (cj != 0) onz |= 1u << r;
                            acc = __builtin_fmaf(wsh[m * M + j], (float)cj + (float)(mm.gamma[j] * mm.alpha[(int64_t)j * (K + 1) + k]), acc);
                        }
                        dn = (double)nk[k] + mm.beta_sum[m];
                    }
                    oth[r] = 0.0; den[r] = 1.0;
                    rden32[r] = __builtin_amdgcn_rcpf((float)dn);
                    brden32[r] = beta32 * rden32[r];
                    oth32[r] = acc * scale32;
                } else {
                    if (used) {
                        cn[r] = sn_get(m * S + i);
                        for (int j = 0; j < M; j++) if (j != m && sn_get(j * S + i) != 0) onz |= 1u << r;
                    }
                    slot_consts(r, oth[r], den[r]);
                    rden32[r] = 0.0f; brden32[r] = 0.0f; oth32[r] = 0.0f;
                }
            }
            // WRK:413-418 newTopicMassAllModalities (the terms were formed with the entity's view pairs, summed here in the reference's order)
            double newAll = 0.0;
            for (int j = 0; j < M; j++) newAll += qsh[m * M + j];
            newAll = newAll * scale_m;
            const double newMass = (mm.first_inactive < 0) ? 0.0 : newAll / (double)K;   // WRK:515
            // the fp32 copies the screening works with (wave-uniform ones as scalars)
            const float pmm32 = uniform_f((float)p_mm), newMass32 = uniform_f((float)newMass);

            const int64_t base = mm.doc_off[m][d];
            const int64_t row0 = mm.rowbase[m];
            const char* nwk_v = (const char*)(nwk + row0 * K);           // the view's rows of the counts, and of their 16-bit mirror
            const char* nwk16_v = (const char*)(nwk16 + row0 * K);
            int koffh[RMAX];                                             // (the slot's byte offset inside a mirror row)
#pragma unroll
            for (int r = 0; r < RMAX; r++) koffh[r] = koff[r] >> 1;
            const int Vm = mm.V[m];
            const double walk_theta = sl.walk_theta[m];
            const unsigned int v_tok0 = n_tok, v_tree0 = c_tree;
            unsigned int whist_l = 0;                                        // lane b < MVHDP_WALK_BINS: tree-branch tokens of this view with u1 in bin b

            MVHDP_TMAIN(tv, te, te);
            for (int c0 = 0; c0 < lenm && !aborted; c0 += WAVE) {
                // one lane per token of the chunk: token id, old topic, its slot, RNG, tree root
                const int ti = c0 + lane;
                const bool tvalid = ti < lenm;
                int w_l = tvalid ? mm.tok[m][base + ti] : -1;
                int z_l = tvalid ? mm.z[m][base + ti] : -1;
                int so_l = -1;
                if (z_l >= 0) {
                    uint32_t w = bitmap[z_l >> 5];
                    so_l = prefix[z_l >> 5] + __popc(w & ((1u << (z_l & 31)) - 1u));
                }
                double u1_l, u2_l;
                {
                    uint32_t x[4];
                    philox4x32_10((uint32_t)ti, (uint32_t)m, (uint32_t)dg, sl.sweep_idx,
                                  sl.seed_lo, sl.seed_hi ^ (uint32_t)((unsigned long long)dg >> 32), x);
                    u1_l = bits_to_unit(x[0], x[1]);
                    u2_l = bits_to_unit(x[2], x[3]);
                }
                MVHDP_TSUB2(tp);                                             // (split 2: token / z loads, the slot lookup, Philox)
                if (w_l >= Vm) w_l = -1;                                     // WRK:427-428 marks OOV
                // NARROW: the row's weight class rides in bit 30 of the type id (a light row's counts are in the 16-bit mirror, a heavy row's
                // only in the 32-bit table, MvModel::heavy), so that the gather of a token's row knows the table to read from a scalar of
                // the broadcast it does anyway -- nothing to check or resolve when the values are used.  (W_ROW strips the bit.)
                if (NARROW && w_l >= 0) {
                    const int hv = mm.heavy[row0 + w_l];
                    if (hv) w_l |= (hv == MVHDP_ROW_HEAVY) ? (W_HEAVY | W_BIG) : W_BIG;
                }
                const float u1f_l = (float)u1_l;                             // (may round to 1.0f: the screening then hands the token to fp64)
                int znew_l = z_l;

                // FT:118-132 for the whole chunk, one lane per token.  The trees do not change during a
                // sweep and u2 belongs to the token, so the topic the tree branch WOULD return (WRK:533-535)
                // does not depend on the entity's state: every lane walks its own word's tree here, 64
                // dependent-load chains in flight at once, and the sequential loop below only picks the
                // result up.  root_l = tree[1] (WRK:519), zt_l = the sampled topic, st_l = its slot or -1.
                // Only a large u1 reaches the tree branch (s1 >= mass, WRK:529), so a token whose u1 is below the view's
                // threshold is not walked here: it takes tree[1] from the 8-byte-per-type root array and, should it reach the
                // branch after all, walks on demand in the token loop (same table, same arithmetic, same topic).
                double root_l = 0.0;
                int zt_l = -1, st_l = -1;
                // live-rows form: only a HEAVY word's token walks a stored tree (heavy rows are not in the mirror, and a word of 65535 tokens and
                // more moves its leaves by 1e-5 a token: its tree of the segment start is current enough); any other token that may reach
                // the tree branch has its row loaded at the top of its turn (specm)
                const bool heavy_l = NARROW && w_l >= 0 && (w_l & W_HEAVY);
                const bool walk_l = tvalid && w_l >= 0 && (!WALK || u1_l >= walk_theta) && (!LIVEROWS || heavy_l);
                const unsigned long long walked = WALK ? __ballot(walk_l) : ~0ull;
                const unsigned long long specm = LIVEROWS ? __ballot(tvalid && w_l >= 0 && !heavy_l && u1_l >= walk_theta) : 0ull;
                rowq_t rowq = {0u, 0u, 0u, 0u};
                if (WALK && tvalid && w_l >= 0 && !walk_l) root_l = mm.root[row0 + W_ROW(w_l)];
                {
                    const bool act = walk_l;
                    const double* __restrict__ dt = mm.dtab + (row0 + W_ROW(max(w_l, 0))) * (int64_t)mm.dt_nblk * 8;
                    double u = 0.0;
                    int i = 1;
                    // descent by 64-byte blocks of the descent table: three levels of the path per sector
                    // (the first block: tree[1] and dt_f levels), every lane on its own word's table
                    for (int bd = 0; bd < mm.dt_nbd; bd++) {
                        if (act && (bd == 0 || i < K)) {
                            const double2* __restrict__ blk = (const double2*)(dt + (int64_t)(mm.dt_base[bd] + (i - (1 << mm.dt_depth[bd]))) * 8);
                            double2 q0, q1, q2, q3;
                            if (LIVEROWS) {
                                // (the heavy words' trees are being rewritten by heavy_refresh_kernel on another XCD while this kernel reads them:
                                // loads of agent scope, past the L2 of this XCD, which plain stores of another one never reach)
                                const double* bp = (const double*)blk;
                                q0.x = __hip_atomic_load(bp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); q0.y = __hip_atomic_load(bp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                q1.x = __hip_atomic_load(bp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); q1.y = __hip_atomic_load(bp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                q2.x = __hip_atomic_load(bp + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); q2.y = __hip_atomic_load(bp + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                q3.x = __hip_atomic_load(bp + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); q3.y = __hip_atomic_load(bp + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            } else { q0 = blk[0]; q1 = blk[1]; q2 = blk[2]; q3 = blk[3]; }
                            const int levels = (bd == 0) ? mm.dt_f : 3;
                            if (bd == 0) { root_l = q3.y; u = u2_l * root_l; }       // FT:120
                            int path = 0;
                            if (i < K && levels > 0) {                                // FT:122-130, level 1 of the block
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
                    if (act) {
                        zt_l = i - K;                                        // FT:132
                        const uint32_t wbit = bitmap[zt_l >> 5];
                        if ((wbit >> (zt_l & 31)) & 1u) st_l = (int)(prefix[zt_l >> 5] + __popc(wbit & ((1u << (zt_l & 31)) - 1u)));
                    }
                }

                MVHDP_TSUB2(tv);                                             // (split 2: heavy flags, roots, the tree walk, the sampled topic's slot)
                const float root32_l = (float)root_l;
                // live-rows form: the tree branch's uniform is kept (the walk on demand of the other flavours draws it again: ten Philox rounds per tree-branch token)
                const float u2f_l = LIVEROWS ? (float)u2_l : 0.0f;
                // live-rows form, rows of two register batches: the batch a tree-branch token of this lane would start in (row_sample_live: the
                // first batch's stored mass against the token's target) -- known here, so that the token's turn loads THAT batch ahead
                unsigned long long bselm = 0ull;
                if (LIVEROWS == 2) {
                    constexpr int BCELLS = WAVE * (NARROW ? 8 : 4);
                    if (K > BCELLS && K <= 2 * BCELLS) {
                        float mass0_l = 0.0f;                                // (needed again by the token that does take the branch: read then, through the scalar cache)
                        if (tvalid && w_l >= 0 && !heavy_l) mass0_l = mm.mass0[row0 + W_ROW(w_l)];
                        bselm = __ballot(tvalid && w_l >= 0 && !heavy_l && (u2f_l * root32_l) - smS >= mass0_l);
                    }
                }
                MVHDP_TMAIN(th, te, tt);
                // software pipeline: the n_wk values of the listed topics are gathered NB tokens ahead, into NB
                // register buffers used in turn (the token loop is unrolled NB times so that no buffer is ever
                // copied while its load is in flight).  Two buffers where few waves share a SIMD and a wave's
                // own latency is what counts; one where six waves hide it and registers are what counts.
                constexpr int NB = (RMAX >= MVHDP_NB2_FROM) ? 2 : 1;
                int gn[RMAX], gn2[RMAX];
                // The loop visits the chunk's tokens of known types only, in position order, off a scalar mask: a token of a type
                // outside the vocabulary (WRK:427-428 skips it) is counted here and never enters the loop -- no path through the loop body
                // leaves the slot registers untouched, which is what lets the compiler update them in place.
                unsigned long long rem = __ballot(tvalid && w_l >= 0);
                n_oov += (unsigned int)__popcll(__ballot(tvalid && w_l < 0));
                n_tok += (unsigned int)__popcll(rem);                       // (counted here, not per token; an abandoned entity fails the sweep anyway)
                const int t_first = rem ? (int)__builtin_ctzll(rem) : 0;
#pragma unroll
                for (int a = 0; a < NB; a++) {
                    const unsigned long long ra = (a == 0) ? rem : (rem & (rem - 1));
                    const int w0 = bcast_i(w_l, ra ? (int)__builtin_ctzll(ra) : t_first);
                    const unsigned int r0 = (unsigned int)W_ROW(max(w0, 0));
                    const bool h0 = w0 >= 0 && (w0 & W_HEAVY);
                    if (NARROW && !h0) {
                        const gptr_t c0q = scalar_row(nwk16_v, r0, (unsigned int)K * 2u);
#pragma unroll
                        for (int r = 0; r < RMAX; r++) { const int v = gather_cell<uint16_t>(c0q + (unsigned int)koffh[r]); if (a == 0) gn[r] = v; else gn2[r] = v; }
                    } else {
                        const gptr_t c0p = scalar_row(nwk_v, r0, (unsigned int)K * 4u);
#pragma unroll
                        for (int r = 0; r < RMAX; r++) { const int v = gather_cell<int32_t>(c0p + (unsigned int)koff[r]); if (a == 0) gn[r] = v; else gn2[r] = v; }
                    }
                }

                while (rem) {                                               // WRK:425
                    const int t = (int)__builtin_ctzll(rem);
                    rem &= ~(1ull << t);
#define TOK_T t
#define TOK_G gn
#include "mvhdp_sweep_fast_token.inc"
#undef TOK_T
#undef TOK_G
                    if (NB == 2 && rem) {
                        const int t1 = (int)__builtin_ctzll(rem);
                        rem &= ~(1ull << t1);
#define TOK_T t1
#define TOK_G gn2
#include "mvhdp_sweep_fast_token.inc"
#undef TOK_T
#undef TOK_G
                    }
                }

                MVHDP_TMAIN(tt, te, te);
                // WRK:587-589 + UPD:197-218 for the whole chunk at once: lane t owns token t (old topic z_l,
                // new topic znew_l), so the FastQDelta records of up to 64 tokens become two wave-wide
                // atomic instructions on the delta rows plus two on the block's n_k table.  Issued after
                // the token loop so that no token ever waits on an atomic's round trip.
                {
                    const bool chg = tvalid && (w_l >= 0) && (znew_l != z_l) && !(sl.flags & MVHDP_SWEEP_FROZEN);
                    n_chg += (unsigned int)__popcll(__ballot(chg));
                    if (chg) {
                        const int64_t rowK = (row0 + W_ROW(w_l)) * K;
                        if (NARROW && sl.live16 && !(w_l & W_HEAVY)) {
                            // a light row of a live16 sweep: its counts live in the mirror, two cells per word
                            unsigned int* m32 = (unsigned int*)mm.counts16;
                            if (z_l >= 0) { const int64_t c = rowK + z_l; __hip_atomic_fetch_add(&m32[c >> 1], 0u - (1u << ((c & 1) * 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                            { const int64_t c = rowK + znew_l; __hip_atomic_fetch_add(&m32[c >> 1], 1u << ((c & 1) * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                        } else if (NARROW && sl.delta16 && !(w_l & W_BIG)) {
                            // a row whose deltas of one sweep stay within +-32767 (its type holds no more tokens than that): 16-bit cells
                            // biased by 0x8000, two a word -- a cell never reaches 0 or 65536 on the way, so +-1 / +-65536 cannot carry --
                            // in a table half the size of the 32-bit one: more of it stays in the Infinity Cache, where the memory-side
                            // atomics are performed (profiles/r04_atomic_cost.md)
                            unsigned int* d32 = (unsigned int*)mm.delta16;
                            if (z_l >= 0) { const int64_t c = rowK + z_l; __hip_atomic_fetch_add(&d32[c >> 1], 0u - (1u << ((c & 1) * 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                            { const int64_t c = rowK + znew_l; __hip_atomic_fetch_add(&d32[c >> 1], 1u << ((c & 1) * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                        } else {
#ifdef MVHDP_PROBE      /* measurement build (tools/mode_times.py --mode atomic_probe): what about the chunk-end atomics costs -- WRONG counts by design */
                            const int probe = (int)((sl.flags >> 16) & 7u);
                            int64_t c_old = rowK + z_l, c_new = rowK + znew_l;
                            if (probe == 1) { c_old &= 0xffff; c_new &= 0xffff; }                 // a 256 KB table: stays in the caches
                            if (probe == 7) { c_old &= 0xfffff; c_new &= 0xfffff; }               // 4 MB
                            if (probe == 5) { c_old &= 0x7fffff; c_new &= 0x7fffff; }             // 32 MB of the delta table
                            if (probe == 6) { c_old &= 0xffffff; c_new &= 0xffffff; }             // 64 MB
                            if (probe == 3) { if (z_l >= 0) dnwk[c_old] = -1; dnwk[c_new] = 1; }  // plain stores to the same cells
                            else if (probe == 4) { if (z_l >= 0) __hip_atomic_fetch_add(&dnwk[c_old], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_fetch_add(&dnwk[c_new], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                            else {
                                if (z_l >= 0) __hip_atomic_fetch_add(&dnwk[c_old], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (probe != 2) __hip_atomic_fetch_add(&dnwk[c_new], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
#else
                            if (z_l >= 0) __hip_atomic_fetch_add(&dnwk[rowK + z_l], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_fetch_add(&dnwk[rowK + znew_l], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
                        }
                        if (z_l >= 0) {
                            if (sl.nk_global) __hip_atomic_fetch_add(&dnk_g[m * K + z_l], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            else __hip_atomic_fetch_add(&nkd[m * K + z_l], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        if (sl.nk_global) __hip_atomic_fetch_add(&dnk_g[m * K + znew_l], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else __hip_atomic_fetch_add(&nkd[m * K + znew_l], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (mm.first_inactive >= 0 && mm.inactive[znew_l]) {          // UPD:263
                            long long key = (long long)MVHDP_ACT_KEY(dg, m, ti, znew_l);
                            atomicMin(sl.act_key, key);
                            if (sl.births) {                                 // the samplers move on to the next inactive index (WRK:523-526)
                                const int r_ = sl.births[2 + K + znew_l];
                                if (r_ >= 0) { atomicMin(&sl.birth_keys[r_], key); atomicMax(&sl.births[0], r_ + 1); }
                            }
                        }
                    }
                }
                if (tvalid) mm.z[m][base + ti] = znew_l;                     // coalesced write-back of the chunk
                if (tvalid && znew_l >= 0) atomicOr(&bitmap2[znew_l >> 5], 1u << (znew_l & 31));
                if (sl.flags & MVHDP_SL_STRICT_LIVE) {                       // (diagnostics: the chunk's updates have landed before anything else is read)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                MVHDP_TMAIN(te, te, te);
            }

            // the view's counts go back to LDS: later views read them (WRK:404) and test them (WRK:445)
#pragma unroll
            for (int r = 0; r < RMAX; r++) {
                const int i = lane * R_eff + r;
                if (r < R_eff && i < S_used) sn_set(m * S + i, cn[r]);
            }
            if (WALK && n_tok != v_tok0) {                                           // per-view branch statistics (block-local): one LDS atomic
                const unsigned int v = lane < MVHDP_WALK_BINS ? whist_l : lane == MVHDP_WALK_BINS ? n_tok - v_tok0 : c_tree - v_tree0;
                const int idx = lane < MVHDP_WALK_BINS ? 2 + lane : lane - MVHDP_WALK_BINS;
                if (lane < MVHDP_WALK_BINS + 2 && v) atomicAdd(&vstat_s[m * MVHDP_VIEW_STATS + idx], v);
            }
            LDS_FENCE();
        }
        if (aborted) n_abort++;
        LDS_FENCE();
        {   // the entity's topic list at its NEXT visit: distinct topics of the assignments just written
            int c2 = __popc(lane < NW ? bitmap2[lane] : 0u);
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) c2 += __shfl_xor(c2, sft, WAVE);
            if (lane == 0) {
                mm.nslots[d] = aborted ? (uint16_t)MVHDP_NSLOTS_UNKNOWN : (uint16_t)c2;   // (an abandoned entity keeps assignments this wave never saw)
                if (sl.slot_hist) {
                    if (c2 > 0) atomicAdd(&hist_s[min((c2 + 63) >> 6, MVHDP_HIST_BINS) - 1], (unsigned int)doc_tokens);
                    atomicAdd(&ent_s[aborted ? MVHDP_N_CLASSES : mvhdp_class_of(c2, longest_view > 65535)], 1u);
                }
            }
        }
        LDS_FENCE();
#ifdef MVHDP_TIMING
        { const int b = e_ord < 2 ? e_ord : 2; t_ent[b] += __builtin_amdgcn_s_memtime() - t_e0; n_ent[b] += n_tok - n_tok_e0; e_ord++; }
#endif
      }
    }

#ifdef MVHDP_TIMING
    const unsigned long long t_loop_end = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
    for (int i = threadIdx.x; i < nkd_len; i += blockDim.x)
        if (nkd[i]) atomicAdd(&dnk_g[i], nkd[i]);
    if (sl.slot_hist && threadIdx.x < MVHDP_HIST_BINS + MVHDP_ENT_BINS && hist_s[threadIdx.x]) atomicAdd(&sl.slot_hist[threadIdx.x], (unsigned long long)hist_s[threadIdx.x]);
    if (WALK && threadIdx.x < MVHDP_MAXM * MVHDP_VIEW_STATS && vstat_s[threadIdx.x])
        atomicAdd(&sl.stats[ST_VIEW_BASE + threadIdx.x], (unsigned long long)vstat_s[threadIdx.x]);
    if (lane == 0) {
        if (n_tok) atomicAdd(&sl.stats[ST_TOKENS], (unsigned long long)n_tok);
        if (n_chg) atomicAdd(&sl.stats[ST_CHANGED], (unsigned long long)n_chg);
        if (c_new) atomicAdd(&sl.stats[ST_NEW], (unsigned long long)c_new);
        const unsigned int c_doc = n_tok - c_new - c_tree;              // (the common branch is not counted per token)
        if (c_doc) atomicAdd(&sl.stats[ST_DOC], (unsigned long long)c_doc);
        if (c_tree) atomicAdd(&sl.stats[ST_TREE], (unsigned long long)c_tree);
        if (n_oov) atomicAdd(&sl.stats[ST_OOV], (unsigned long long)n_oov);
        if (n_abort) atomicAdd(&sl.stats[ST_ABORT], (unsigned long long)n_abort);
        if (n_fb) atomicAdd(&sl.stats[ST_FALLBACK], (unsigned long long)n_fb);
        if (n_od) atomicAdd(&sl.stats[ST_ONDEMAND], (unsigned long long)n_od);
        if (n_misclass) atomicAdd(&sl.stats[ST_MISCLASS], (unsigned long long)n_misclass);
#ifdef MVHDP_TIMING
        atomicAdd(&sl.stats[ST_T_QUEUE], tq); atomicAdd(&sl.stats[ST_T_PROLOGUE], tp); atomicAdd(&sl.stats[ST_T_VIEW], tv);
        atomicAdd(&sl.stats[ST_T_CHUNK_HEAD], th); atomicAdd(&sl.stats[ST_T_TOKENS], tt); atomicAdd(&sl.stats[ST_T_CHUNK_END], te);
        atomicAdd(&sl.stats[ST_T_TOTAL], (unsigned long long)__builtin_amdgcn_s_memtime() - t_begin);
        for (int b = 0; b < 3; b++) { atomicAdd(&sl.stats[ST_T_ENT0 + b], t_ent[b]); atomicAdd(&sl.stats[ST_N_ENT0 + b], n_ent[b]); }
        atomicAdd(&sl.stats[ST_T_INIT], t_init_end - t_begin0); atomicAdd(&sl.stats[ST_T_FLUSH], (unsigned long long)__builtin_amdgcn_s_memtime() - t_loop_end);
        atomicAdd(&sl.stats[ST_N_WAVES], 1ull);
        atomicAdd(&sl.stats[ST_T_ROWS], t_rows); atomicAdd(&sl.stats[ST_T_ROWS_WAIT], t_rows_wait);
#endif
    }
}

#undef sn_get
#undef sn_set

// debug launches always take the WALK flavour (one instantiation fewer per variant; a threshold of 0 walks every token)
static bool roomy_build(int rmax, int K) { return rmax == 2 && K >= 512; }
// live-rows flavour with the two-batch shortcut: a row of the 16-bit mirror in exactly two register batches of 512 cells (the roomy 2-round
// build, which exists for K >= 512 only, is always that flavour: the shortcut checks the batch count at run time as well)
static bool two_batch_rows(int K) { return K > 512 && K <= 1024; }

template <int RMAX>
static const void* fast_kernel_ptr(bool debug, bool walk, bool narrow, int K = 0, bool live_rows = false)
{
    if (live_rows && !debug) {
        if constexpr (RMAX == 2) { if (narrow && roomy_build(RMAX, K)) return (const void*)sweep_fast_kernel<2, false, true, true, true, 2>; }
        if (narrow && two_batch_rows(K)) return (const void*)sweep_fast_kernel<RMAX, false, true, true, false, 2>;
        return narrow ? (const void*)sweep_fast_kernel<RMAX, false, true, true, false, 1> : (const void*)sweep_fast_kernel<RMAX, false, true, false, false, 1>;
    }
    if (narrow && walk && !debug) {
        if constexpr (RMAX == 2) { if (roomy_build(RMAX, K)) return (const void*)sweep_fast_kernel<2, false, true, true, true>; }
        return (const void*)sweep_fast_kernel<RMAX, false, true, true>;
    }
    return debug ? (const void*)sweep_fast_kernel<RMAX, true, true, false>
                 : walk ? (const void*)sweep_fast_kernel<RMAX, false, true, false> : (const void*)sweep_fast_kernel<RMAX, false, false, false>;
}

template <int RMAX>
static hipError_t launch_fast(const MvModel& mm, const SweepLaunch& sl, int grid_blocks, bool debug, hipStream_t s)
{
    size_t lds = sl.block_shared_bytes + (size_t)sl.waves_per_block * sl.wave_bytes;
    dim3 block(64 * sl.waves_per_block);
    const bool narrow = sl.narrow && sl.walk && !debug;
    const bool rows = sl.live_rows && sl.walk && !debug;
    if (lds > 65536) {
        hipError_t e = hipFuncSetAttribute(fast_kernel_ptr<RMAX>(debug, sl.walk != 0, narrow, mm.K, rows), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (debug)        hipLaunchKernelGGL((sweep_fast_kernel<RMAX, true, true, false>), dim3(grid_blocks), block, lds, s, mm, sl);
    else if (rows) {
        if (narrow && roomy_build(RMAX, mm.K)) { if constexpr (RMAX == 2) hipLaunchKernelGGL((sweep_fast_kernel<2, false, true, true, true, 2>), dim3(grid_blocks), block, lds, s, mm, sl); }
        else if (narrow && two_batch_rows(mm.K)) hipLaunchKernelGGL((sweep_fast_kernel<RMAX, false, true, true, false, 2>), dim3(grid_blocks), block, lds, s, mm, sl);
        else if (narrow) hipLaunchKernelGGL((sweep_fast_kernel<RMAX, false, true, true, false, 1>), dim3(grid_blocks), block, lds, s, mm, sl);
        else             hipLaunchKernelGGL((sweep_fast_kernel<RMAX, false, true, false, false, 1>), dim3(grid_blocks), block, lds, s, mm, sl);
    }
    else if (narrow && roomy_build(RMAX, mm.K)) {
        // (the plan sized the grid for the 72-register build: the seventh block of a CU waits for a free slot and finds the queue empty)
        if constexpr (RMAX == 2) hipLaunchKernelGGL((sweep_fast_kernel<2, false, true, true, true>), dim3(grid_blocks), block, lds, s, mm, sl);
    }
    else if (narrow)  hipLaunchKernelGGL((sweep_fast_kernel<RMAX, false, true, true>), dim3(grid_blocks), block, lds, s, mm, sl);
    else if (sl.walk) hipLaunchKernelGGL((sweep_fast_kernel<RMAX, false, true, false>), dim3(grid_blocks), block, lds, s, mm, sl);
    else              hipLaunchKernelGGL((sweep_fast_kernel<RMAX, false, false, false>), dim3(grid_blocks), block, lds, s, mm, sl);
    return hipGetLastError();
}

hipError_t mvhdp_launch_sweep_fast(const MvModel& mm, const SweepLaunch& sl, int rmax, int grid_blocks, bool debug, hipStream_t s)
{
    switch (rmax) {
    case 1: return launch_fast<1>(mm, sl, grid_blocks, debug, s);
    case 2: return launch_fast<2>(mm, sl, grid_blocks, debug, s);
    case 3:
    case 4: return launch_fast<4>(mm, sl, grid_blocks, debug, s);
    case 8: return launch_fast<8>(mm, sl, grid_blocks, debug, s);
    case 16: return launch_fast<16>(mm, sl, grid_blocks, debug, s);
    default: return hipErrorInvalidValue;
    }
}

// Resident blocks per CU from the kernel's own register count and LDS need (the occupancy API
// mis-reports both directions for these kernels; an over-estimate only queues blocks, an
// under-estimate idles SIMDs).
static int blocks_per_cu_from(const void* func, int threads, size_t lds)
{
    hipFuncAttributes a;
    if (hipFuncGetAttributes(&a, func) != hipSuccess) return 1;
    int regs = (a.numRegs + 7) / 8 * 8;
    int waves_simd = regs > 0 ? 512 / regs : 8;
    if (waves_simd > 8) waves_simd = 8;
    if (waves_simd < 1) waves_simd = 1;
    int wpb = threads / 64;
    int by_regs = waves_simd * 4 / wpb;
    int by_lds = (int)((160 * 1024) / (lds > 0 ? lds : 1));
    int by_waves = 32 / wpb;
    int b = by_regs < by_lds ? by_regs : by_lds;
    if (by_waves < b) b = by_waves;
    return b < 1 ? 1 : b;
}

template <int RMAX>
static int occ_fast(bool debug, bool walk, int threads, size_t lds)
{
    return blocks_per_cu_from(fast_kernel_ptr<RMAX>(debug, walk, false), threads, lds);
}

int mvhdp_sweep_fast_occupancy(int rmax, bool debug, bool walk, int block_threads, size_t lds_bytes)
{
    switch (rmax) {
    case 1: return occ_fast<1>(debug, walk, block_threads, lds_bytes);
    case 2: return occ_fast<2>(debug, walk, block_threads, lds_bytes);
    case 3:
    case 4: return occ_fast<4>(debug, walk, block_threads, lds_bytes);
    case 8: return occ_fast<8>(debug, walk, block_threads, lds_bytes);
    case 16: return occ_fast<16>(debug, walk, block_threads, lds_bytes);
    default: return 0;
    }
}

template <int RMAX>
static int regs_fast(int flavour)
{
    hipFuncAttributes a;
    if (hipFuncGetAttributes(&a, fast_kernel_ptr<RMAX>(flavour == 2, flavour >= 1, false)) != hipSuccess) return 128;
    return a.numRegs;
}

int mvhdp_sweep_kernel_regs(int cls, int flavour)
{
    switch (cls) {
    case 0: return regs_fast<1>(flavour);
    case 1: return regs_fast<2>(flavour);
    case 2: return regs_fast<4>(flavour);
    case 3: return regs_fast<8>(flavour);
    case 4: return regs_fast<16>(flavour);
    default: return mvhdp_sweep_generic_regs(flavour == 2);
    }
}
