// mvhdp_api.hip — host side of the C ABI declared in include/mvhdp.h.
// Owns the device state of one model shard (one HIP device), launches the
// kernels of mvhdp_kernels.hip on one stream and copies results back.
// There is NO CPU fallback: without a gfx950 device mvhdp_create fails.
#include "mvhdp_device.h"
#include "../../include/mvhdp.h"

#include <algorithm>
#include <functional>
#include <climits>
#include <mutex>
#include <set>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static thread_local std::string g_create_error;

struct mvhdp_ctx {
    mvhdp_config cfg{};
    MvModel mm{};
    int device = 0;
    int num_cus = 256;
    size_t max_lds = 65536;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev[4]{};
    std::string err;

    int64_t N[MVHDP_MAXM]{};                 // tokens per view
    bool have_corpus[MVHDP_MAXM]{};
    std::vector<int64_t> h_doc_off[MVHDP_MAXM];
    void* d_doc_off[MVHDP_MAXM]{};
    void* d_tok[MVHDP_MAXM]{};
    void* d_z[MVHDP_MAXM]{};
    int64_t max_doc_tokens = -1;             // over all views, lazily computed

    double* d_alpha = nullptr;
    uint8_t* d_inactive = nullptr;
    std::vector<double> h_alpha;
    std::vector<uint8_t> h_inactive;
    bool have_hyper = false, have_counts = false, have_trees = false;
    bool full_trees = false;                 // the FTree.tree arrays are current too (a sweep may refresh only the descent table)
    bool trees_inference = false;            // leaves of the last build: p_wt alone (INF:576)
    bool delta_clean = false;                // the delta buffer is known to be all zero
    bool delta_pending = false;              // a NO_APPLY sweep has left deltas that mvhdp_apply_delta has not consumed yet
    bool last_need_full = true;              // the last sweep's kernels could reach the generic kernel (needs FTree.tree itself)
    int64_t rows_applied = -1;               // mvhdp_apply_delta_rows progress of the current begin/end bracket (-1: no bracket open)
    bool device_released = false;            // release_device_resources has run (mvhdp_destroy, or the exit handler)

    unsigned long long* d_stats = nullptr;   // [ST_COUNT]
    long long* d_act_key = nullptr;
    unsigned long long* d_doc_counter = nullptr;
    int32_t* d_doc_order = nullptr;          // entities by decreasing token count (work-queue order)
    int32_t* d_overflow = nullptr;           // [D] entities handed from the primary register-resident variant to the next pass
    int32_t* d_overflow2 = nullptr;          // [D] entities that exceed even the 16-slot variant: generic LDS kernel
    unsigned int* d_ovf_meta = nullptr;      // u32 overflow counts of pass 1 and 2, at byte 8: u64[17] tokens by ceil(topic list/64), then the count of pass 3
    int32_t* d_lists = nullptr;              // classified mode: [MVHDP_N_CLASSES][D] entity lists written by classify_kernel
    hipStream_t side[MVHDP_N_CLASSES]{};     // one stream per wider kernel class (created on first use)
    hipEvent_t ev_fork = nullptr, ev_join[MVHDP_N_CLASSES]{};
    std::vector<int64_t> tokens_desc;        // entity token counts, descending (the order of d_doc_order)
    int64_t* d_carry[MVHDP_MAXM]{};          // doc_topic_proportions: per view, the entity whose view-m counts score entity d (lazily built)
    unsigned long long last_hist[MVHDP_HIST_BINS]{};   // tokens by topic-list size class, from the last sweep (or the probe)
    int rmax_hint = 0;                       // slots/64 the next sweep's register-resident kernel is sized for (0 = estimate)
    // 1-round or 2-round primary variant?  Measured, not tabulated: once half of the tokens sit in topic lists of at most 64 slots
    // the 1-round variant runs for ONE sweep (the longer lists on their own class kernels, or -- below 0.5 % -- in an optimistic
    // overflow pass); if its kernel time per token is not better than the 2-round variant's of the sweep before, the choice goes
    // back to 2 rounds and is not tried again for 4 sweeps (8, 16, 32 after repeated failures).
    int last_primary = 0;                    // primary variant of the last plain sweep (0: none)
    double last_ns_per_token = 0;            //   and its sweep-kernel time per token
    double two_round_ns_per_token = 0;       // the 2-round variant's time per token just before a 1-round trial
    long long sweeps_done = 0, one_round_banned_until = 0, one_round_ban = 4;
    // Walk threshold of the chunk head (SweepLaunch::walk_theta, in steps of 1/MVHDP_WALK_BINS): which tokens have their word tree
    // walked up front.  It changes when the walk is done, never what is sampled, so it is steered by the clock: sweeps at the current
    // threshold (A) alternate with sweeps a step away (B); B replaces A when its kernel time per token beats the mean of the A sweeps
    // on either side (the chain's own drift cancels).  An upward step is as long as the last sweep's histogram of the tree-branch
    // tokens' u1 says is nearly free (<= 0.4 % of the tokens more to walk on demand; the allowance doubles after a step that paid
    // more than 0.8 % and halves after a long step that did not pay), a downward step is one bin.  A step that does
    // not pay turns the search around; two in a row let it rest for a growing number of sweeps, and the first B sweep after a rest
    // tries half the threshold (a slope too shallow for single steps to see, e.g. where the kernel is not bandwidth-bound and
    // the best threshold is 0).  Views where most tokens take the tree branch anyway (walk_f >= 0.35) are always walked.
    int walk_i = 0, walk_probe_i = 0, walk_b_i = 0, walk_phase = 0, walk_dir = 1, walk_fails = 0, walk_wait = 4, walk_cfg = -1, walk_maxj = 6;
    bool walk_far = false;                   // the next B sweep after a rest tries half the threshold (slopes too shallow for single steps)
    double walk_ns_a1 = 0.0, walk_ns_b = 0.0;
    int walk_cls = 1, walk_i_by[2] = {-1, -1}; // the threshold is kept per variant class ([0] the 1-round variant, [1] the wider ones; -1: inherits on first use)
    bool one_round_retry = false;            // the 1-round trial failed at a low threshold: one more sweep at a high one before the ban
    double walk_cap = 0.004;                 // share of the tokens an upward step may add to the walks on demand: doubles after a step that paid well
    long long walk_idle_until = 0, walk_refresh_at = 0;
    double walk_f[MVHDP_MAXM] = {-1, -1, -1, -1, -1, -1, -1, -1};   // tree-branch share per view in the last sweep (< 0: not known yet)
    double walk_hist[MVHDP_WALK_BINS] = {0};  // tree-branch tokens of those views by u1 bin, as a share of all tokens (last sweep)
    size_t lds_attr_set = 0;
};

// ---- handle registry and process exit ------------------------------------------------------------------
// Every live handle is listed here.  The first mvhdp_create registers an atexit handler; it is registered AFTER the
// HIP runtime initialised (hipGetDeviceCount in mvhdp_create comes first), so at exit it runs BEFORE the runtime's own
// teardown: it releases the device resources of every handle still open and marks the process as exiting.  A host that
// closes a handle later than that -- a JVM finalizer or shutdown hook calling NativeSampler.close(), a static destructor
// of the embedding program -- reaches mvhdp_destroy with g_exiting set: no HIP call is made any more, only host memory is
// released.  mvhdp_destroy of a pointer that is not (or no longer) a live handle is refused instead of dereferenced.
static std::mutex g_reg_mutex;
static std::set<mvhdp_ctx*>* g_live = nullptr;          // heap-allocated and never freed: usable during static destruction
static bool g_exiting = false, g_atexit_registered = false;
static void release_device_resources(mvhdp_ctx* h);

static void mvhdp_at_exit()
{
    std::lock_guard<std::mutex> lk(g_reg_mutex);
    g_exiting = true;
    if (g_live) for (mvhdp_ctx* h : *g_live) release_device_resources(h);     // the runtime is still alive here
}

static void register_handle(mvhdp_ctx* h)
{
    std::lock_guard<std::mutex> lk(g_reg_mutex);
    if (!g_live) g_live = new std::set<mvhdp_ctx*>();
    g_live->insert(h);
    if (!g_atexit_registered) { atexit(mvhdp_at_exit); g_atexit_registered = true; }
}

static bool is_live(mvhdp_ctx* h)
{
    std::lock_guard<std::mutex> lk(g_reg_mutex);
    return g_live && g_live->count(h) != 0;
}

#define CHECK_H(h) do { if (!(h) || !is_live(h)) return MVHDP_ERR_INVALID_ARG; \
                        if ((h)->device_released) return MVHDP_ERR_STATE; /* the process is exiting */ } while (0)
#define HIPC(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    (h)->err = std::string(#call) + ": " + hipGetErrorString(e_); return MVHDP_ERR_HIP; } } while (0)
#define FAIL(h, code, msg) do { (h)->err = (msg); return (code); } while (0)

// Which register-resident variant (64*r topic slots per entity, r = 1,2,4,8,16) should the next sweep's
// first pass use?  hist[b] = tokens of the entities whose topic list needs b+1 rounds of 64 slots
// (b = 16: more than 1024 slots).  A larger r costs registers, i.e. resident waves (relative cost per
// token below, measured on C4/C5); entities that do not fit go to a second pass with the 16-slot
// variant and, beyond 1024 slots, to the generic LDS kernel.  Returns 32 when the generic kernel alone
// is the cheapest.
static int rmax_from_hist(const unsigned long long* hist)
{
    static const int variants[5] = {1, 2, 4, 8, 16};
    // The 1-round variant against the 2-round one is config-dependent (12 % faster on C3, K = 200; 0.6 % SLOWER on C4,
    // K = 400, where its seventh wave per SIMD buys nothing): the table only proposes it, the sweep measures it
    // (try_one_round below).
    static const double cost[5] = {0.95, 1.0, 1.45, 2.6, 4.5};
    const double cost_generic = 6.0;
    double tot = 0;
    for (int i = 0; i < MVHDP_HIST_BINS; i++) tot += (double)hist[i];
    if (tot == 0) return 1;
    int best = 32; double best_cost = cost_generic * tot;
    for (int v = 0; v < 5; v++) {
        double c = 0;
        for (int b = 0; b < MVHDP_HIST_BINS; b++) {
            const double t = (double)hist[b];
            if (b + 1 <= variants[v]) c += t * cost[v];
            else c += t * (0.02 * cost[v] + (b + 1 <= 8 ? cost[3] : (b + 1 <= 16 ? cost[4] : cost_generic)));   // prologue in pass 1 + the later pass
        }
        if (c < best_cost) { best_cost = c; best = variants[v]; }
    }
    return best;
}

static int64_t counts_len(const mvhdp_ctx* h) { return h->mm.rowbase[h->mm.M] * h->mm.K + (int64_t)h->mm.M * h->mm.K; }

extern "C" const char* mvhdp_version(void) { return "mvhdp 0.1 (gfx950)"; }

extern "C" const char* mvhdp_last_error(mvhdp_handle h) { return (h && is_live(h)) ? h->err.c_str() : g_create_error.c_str(); }

extern "C" int mvhdp_create(const mvhdp_config* cfg, mvhdp_handle* out)
{
    if (!cfg || !out) { g_create_error = "null argument"; return MVHDP_ERR_INVALID_ARG; }
    *out = nullptr;
    const int K = cfg->num_topics, M = cfg->num_modalities;
    if (K < 1 || K > MVHDP_MAX_TOPICS || M < 1 || M > MVHDP_MAX_MODALITIES) {
        g_create_error = "num_topics must be in [1,2048] and num_modalities in [1,8]";
        return MVHDP_ERR_INVALID_ARG;
    }
    for (int m = 0; m < M; m++) if (cfg->num_types[m] < 1) { g_create_error = "num_types[m] must be >= 1"; return MVHDP_ERR_INVALID_ARG; }
    if (cfg->doc_id_base < 0 || cfg->doc_id_base >= (1LL << 29)) { g_create_error = "doc_id_base out of range"; return MVHDP_ERR_INVALID_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
        g_create_error = "no usable HIP device (the sweep has no CPU fallback)";
        return MVHDP_ERR_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) { g_create_error = "hipGetDeviceProperties failed"; return MVHDP_ERR_HIP; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return MVHDP_ERR_NO_DEVICE;
    }
    mvhdp_ctx* h = new mvhdp_ctx();
    register_handle(h);
    h->cfg = *cfg;
    h->device = cfg->device;
    h->num_cus = prop.multiProcessorCount;
    h->max_lds = 160 * 1024;
    MvModel& mm = h->mm;
    mm.K = K; mm.M = M;
    mm.rowbase[0] = 0;
    for (int m = 0; m < M; m++) { mm.V[m] = cfg->num_types[m]; mm.rowbase[m + 1] = mm.rowbase[m] + cfg->num_types[m]; }
    mm.doc_id_base = cfg->doc_id_base;
    mm.first_inactive = -1;
    mm.D = -1;
#define CREATE_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    g_create_error = std::string(#call) + ": " + hipGetErrorString(e_); mvhdp_destroy(h); return MVHDP_ERR_HIP; } } while (0)
    CREATE_HIP(hipSetDevice(h->device));
    CREATE_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
    for (auto& e : h->ev) CREATE_HIP(hipEventCreate(&e));
    const int64_t nrows = mm.rowbase[M];
    const size_t cbytes = (size_t)(nrows * K + (int64_t)M * K) * sizeof(int32_t);
    CREATE_HIP(hipMalloc(&mm.counts, cbytes));
    CREATE_HIP(hipMalloc(&mm.delta, cbytes));
    CREATE_HIP(hipMemset(mm.counts, 0, cbytes));
    CREATE_HIP(hipMalloc(&mm.counts16, (size_t)nrows * K * sizeof(uint16_t)));
    CREATE_HIP(hipMemset(mm.counts16, 0, (size_t)nrows * K * sizeof(uint16_t)));
    CREATE_HIP(hipMemset(mm.delta, 0, cbytes));
    CREATE_HIP(hipMalloc(&mm.trees, (size_t)nrows * 2 * K * sizeof(double)));
    CREATE_HIP(hipMalloc(&mm.root, (size_t)nrows * sizeof(double)));
    {
        // descent table layout (MvModel::dtab): internal levels nlev, first block dt_f levels, then blocks of three
        const int nlev = (K > 1) ? (32 - __builtin_clz((unsigned)(K - 1))) : 0;
        mm.dt_f = nlev ? ((nlev - 1) % 3) + 1 : 0;
        mm.dt_nbd = 1; mm.dt_base[0] = 0; mm.dt_depth[0] = 0;
        int nblk = 1;
        for (int dep = mm.dt_f; nlev && dep < nlev; dep += 3) {
            mm.dt_base[mm.dt_nbd] = nblk; mm.dt_depth[mm.dt_nbd] = dep; mm.dt_nbd++;
            nblk += 1 << dep;
        }
        mm.dt_nblk = nblk;
        CREATE_HIP(hipMalloc(&mm.dtab, (size_t)nrows * nblk * 8 * sizeof(double)));
    }
    CREATE_HIP(hipMalloc(&h->d_alpha, (size_t)M * (K + 1) * sizeof(double)));
    CREATE_HIP(hipMalloc(&h->d_inactive, (size_t)K));
    CREATE_HIP(hipMemset(h->d_inactive, 0, (size_t)K));
    CREATE_HIP(hipMalloc(&h->d_stats, ST_COUNT * sizeof(unsigned long long)));
    CREATE_HIP(hipMalloc(&h->d_act_key, sizeof(long long)));
    CREATE_HIP(hipMalloc(&h->d_doc_counter, 8 * sizeof(unsigned long long)));     // one work-queue head per kernel class
    CREATE_HIP(hipMalloc(&h->d_ovf_meta, 256));                                       // see META_* below
    mm.alpha = h->d_alpha;
    mm.inactive = h->d_inactive;
    h->h_alpha.assign((size_t)M * (K + 1), 0.0);
    h->h_inactive.assign((size_t)K, 0);
    *out = h;
    return MVHDP_OK;
}

// frees everything the handle holds on the device; idempotent (every pointer is cleared)
static void release_device_resources(mvhdp_ctx* h)
{
    if (h->device_released) return;
    h->device_released = true;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    auto fr = [](auto*& p) { if (p) { hipFree((void*)p); p = nullptr; } };
    for (int m = 0; m < MVHDP_MAXM; m++) { fr(h->d_doc_off[m]); fr(h->d_tok[m]); fr(h->d_z[m]); fr(h->d_carry[m]); }
    fr(h->mm.counts); fr(h->mm.counts16); fr(h->mm.delta); fr(h->mm.trees); fr(h->mm.root); fr(h->mm.dtab); fr(h->mm.p);
    fr(h->d_alpha); fr(h->d_inactive); fr(h->d_stats); fr(h->d_act_key); fr(h->d_doc_counter);
    fr(h->d_doc_order); fr(h->d_overflow); fr(h->d_overflow2); fr(h->d_ovf_meta); fr(h->d_lists);
    for (auto& e : h->ev) if (e) { hipEventDestroy(e); e = nullptr; }
    if (h->ev_fork) { hipEventDestroy(h->ev_fork); h->ev_fork = nullptr; }
    for (auto& e : h->ev_join) if (e) { hipEventDestroy(e); e = nullptr; }
    for (auto& st : h->side) if (st) { hipStreamDestroy(st); st = nullptr; }
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    h->stream = nullptr;
}

extern "C" int mvhdp_destroy(mvhdp_handle h)
{
    if (!h) return MVHDP_OK;
    bool exiting;
    {
        std::lock_guard<std::mutex> lk(g_reg_mutex);
        if (!g_live || g_live->erase(h) == 0) return MVHDP_ERR_INVALID_ARG;      // not a live handle (closed twice?)
        exiting = g_exiting;
    }
    // after the exit handler has run the HIP runtime may be gone: it released the device side already, touch nothing
    if (!exiting) release_device_resources(h);
    delete h;
    return MVHDP_OK;
}

extern "C" int mvhdp_set_stream(mvhdp_handle h, void* hip_stream)
{
    CHECK_H(h);
    HIPC(h, hipSetDevice(h->device));
    HIPC(h, hipStreamSynchronize(h->stream));
    if (h->own_stream) { hipStreamDestroy(h->stream); h->own_stream = false; }
    if (hip_stream) h->stream = (hipStream_t)hip_stream;
    else { HIPC(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
    return MVHDP_OK;
}

extern "C" int mvhdp_synchronize(mvhdp_handle h)
{
    CHECK_H(h);
    HIPC(h, hipSetDevice(h->device));
    HIPC(h, hipStreamSynchronize(h->stream));
    return MVHDP_OK;
}

extern "C" int mvhdp_set_corpus(mvhdp_handle h, int32_t m, int64_t D, const int64_t* doc_off, const int32_t* tokens)
{
    CHECK_H(h);
    MvModel& mm = h->mm;
    if (m < 0 || m >= mm.M || D < 0 || !doc_off) FAIL(h, MVHDP_ERR_INVALID_ARG, "set_corpus: bad view, num_docs or doc_off");
    if (mm.D >= 0 && D != mm.D) {
        bool any_other = false;
        for (int j = 0; j < mm.M; j++) if (j != m && h->have_corpus[j]) any_other = true;
        if (any_other) FAIL(h, MVHDP_ERR_INVALID_ARG, "set_corpus: every view must list the same entities (empty span = view absent)");
    }
    if (mm.doc_id_base + D >= (1LL << 29)) FAIL(h, MVHDP_ERR_INVALID_ARG, "set_corpus: too many entities");
    if (doc_off[0] != 0) FAIL(h, MVHDP_ERR_INVALID_ARG, "set_corpus: doc_off[0] must be 0");
    for (int64_t d = 0; d < D; d++) {
        int64_t len = doc_off[d + 1] - doc_off[d];
        if (len < 0) FAIL(h, MVHDP_ERR_INVALID_ARG, "set_corpus: doc_off must be non-decreasing");
        if (len >= (1 << 20)) FAIL(h, MVHDP_ERR_INVALID_ARG, "set_corpus: a view of one entity is limited to 2^20-1 tokens");
    }
    const int64_t N = doc_off[D];
    if (N > 0 && !tokens) FAIL(h, MVHDP_ERR_INVALID_ARG, "set_corpus: tokens is null");
    HIPC(h, hipSetDevice(h->device));
    HIPC(h, hipStreamSynchronize(h->stream));
    if (h->d_doc_off[m]) { hipFree(h->d_doc_off[m]); h->d_doc_off[m] = nullptr; }
    if (h->d_tok[m]) { hipFree(h->d_tok[m]); h->d_tok[m] = nullptr; }
    if (h->d_z[m]) { hipFree(h->d_z[m]); h->d_z[m] = nullptr; }
    HIPC(h, hipMalloc(&h->d_doc_off[m], (size_t)(D + 1) * sizeof(int64_t)));
    HIPC(h, hipMemcpy(h->d_doc_off[m], doc_off, (size_t)(D + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    const size_t nb = (size_t)std::max<int64_t>(N, 1) * sizeof(int32_t);
    HIPC(h, hipMalloc(&h->d_tok[m], nb));
    HIPC(h, hipMalloc(&h->d_z[m], nb));
    if (N > 0) HIPC(h, hipMemcpy(h->d_tok[m], tokens, (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPC(h, hipMemset(h->d_z[m], 0xff, nb));               // UNASSIGNED_TOPIC (-1), PTM:63
    h->h_doc_off[m].assign(doc_off, doc_off + D + 1);
    h->N[m] = N;
    h->have_corpus[m] = true;
    h->max_doc_tokens = -1;
    if (h->d_doc_order) { hipFree(h->d_doc_order); h->d_doc_order = nullptr; }
    if (h->d_overflow) { hipFree(h->d_overflow); h->d_overflow = nullptr; }
    if (h->d_overflow2) { hipFree(h->d_overflow2); h->d_overflow2 = nullptr; }
    if (h->d_lists) { hipFree(h->d_lists); h->d_lists = nullptr; }
    for (auto& c : h->d_carry) if (c) { hipFree(c); c = nullptr; }
    h->rmax_hint = 0;
    mm.D = D;
    mm.doc_off[m] = (const int64_t*)h->d_doc_off[m];
    mm.tok[m] = (const int32_t*)h->d_tok[m];
    mm.z[m] = (int32_t*)h->d_z[m];
    if (mm.p) { hipFree(mm.p); mm.p = nullptr; }
    h->have_counts = false; h->have_trees = false;
    return MVHDP_OK;
}

static int require_corpus(mvhdp_ctx* h)
{
    for (int m = 0; m < h->mm.M; m++)
        if (!h->have_corpus[m]) FAIL(h, MVHDP_ERR_STATE, "set_corpus has not been called for every view");
    return MVHDP_OK;
}

extern "C" int mvhdp_set_assignments(mvhdp_handle h, int32_t m, const int32_t* z)
{
    CHECK_H(h);
    if (m < 0 || m >= h->mm.M) FAIL(h, MVHDP_ERR_INVALID_ARG, "set_assignments: bad view");
    if (!h->have_corpus[m]) FAIL(h, MVHDP_ERR_STATE, "set_assignments before set_corpus");
    if (h->N[m] > 0 && !z) FAIL(h, MVHDP_ERR_INVALID_ARG, "set_assignments: null z");
    for (int64_t i = 0; i < h->N[m]; i++)
        if (z[i] < -1 || z[i] >= h->mm.K) FAIL(h, MVHDP_ERR_INVALID_ARG, "set_assignments: topic out of range");
    HIPC(h, hipSetDevice(h->device));
    HIPC(h, hipStreamSynchronize(h->stream));
    if (h->N[m] > 0) HIPC(h, hipMemcpy(h->d_z[m], z, (size_t)h->N[m] * sizeof(int32_t), hipMemcpyHostToDevice));
    h->rmax_hint = 0;
    return MVHDP_OK;
}

extern "C" int mvhdp_get_assignments(mvhdp_handle h, int32_t m, int32_t* z)
{
    CHECK_H(h);
    if (m < 0 || m >= h->mm.M) FAIL(h, MVHDP_ERR_INVALID_ARG, "get_assignments: bad view");
    if (!h->have_corpus[m]) FAIL(h, MVHDP_ERR_STATE, "get_assignments before set_corpus");
    HIPC(h, hipSetDevice(h->device));
    HIPC(h, hipStreamSynchronize(h->stream));
    if (h->N[m] > 0) HIPC(h, hipMemcpy(z, h->d_z[m], (size_t)h->N[m] * sizeof(int32_t), hipMemcpyDeviceToHost));
    return MVHDP_OK;
}

extern "C" int mvhdp_set_hyper(mvhdp_handle h, const mvhdp_hyper* hy)
{
    CHECK_H(h);
    if (!hy || !hy->alpha) FAIL(h, MVHDP_ERR_INVALID_ARG, "set_hyper: null");
    MvModel& mm = h->mm;
    const int K = mm.K, M = mm.M;
    for (int m = 0; m < M; m++) {
        // (n_wk+beta)/(n_k+betaSum) is evaluated with the unscaled IEEE division sequence (div_inrange):
        // keep both operands far from the exponent limits
        if (!(hy->beta_sum[m] >= 1e-30 && hy->beta_sum[m] <= 1e30) || !(hy->beta[m] >= 1e-30 && hy->beta[m] <= 1e30))
            FAIL(h, MVHDP_ERR_INVALID_ARG, "set_hyper: beta and beta_sum must be in [1e-30, 1e30]");
        mm.alpha_sum[m] = hy->alpha_sum[m]; mm.beta[m] = hy->beta[m];
        mm.beta_sum[m] = hy->beta_sum[m];   mm.gamma[m] = hy->gamma[m];
        for (int j = 0; j < M; j++) { mm.p_a[m][j] = hy->p_a[m][j]; mm.p_b[m][j] = hy->p_b[m][j]; }
    }
    h->h_alpha.assign(hy->alpha, hy->alpha + (size_t)M * (K + 1));
    if (hy->inactive) h->h_inactive.assign(hy->inactive, hy->inactive + K);
    else h->h_inactive.assign((size_t)K, 0);
    mm.first_inactive = -1;
    for (int k = 0; k < K; k++) if (h->h_inactive[k]) { mm.first_inactive = k; break; }
    HIPC(h, hipSetDevice(h->device));
    HIPC(h, hipStreamSynchronize(h->stream));
    HIPC(h, hipMemcpy(h->d_alpha, h->h_alpha.data(), (size_t)M * (K + 1) * sizeof(double), hipMemcpyHostToDevice));
    HIPC(h, hipMemcpy(h->d_inactive, h->h_inactive.data(), (size_t)K, hipMemcpyHostToDevice));
    h->have_hyper = true;
    h->have_trees = false;
    return MVHDP_OK;
}

extern "C" int mvhdp_get_alpha(mvhdp_handle h, double* alpha, uint8_t* inactive)
{
    CHECK_H(h);
    if (!h->have_hyper) FAIL(h, MVHDP_ERR_STATE, "get_alpha before set_hyper");
    if (alpha) memcpy(alpha, h->h_alpha.data(), h->h_alpha.size() * sizeof(double));
    if (inactive) memcpy(inactive, h->h_inactive.data(), h->h_inactive.size());
    return MVHDP_OK;
}

extern "C" int mvhdp_build_counts(mvhdp_handle h)
{
    CHECK_H(h);
    int rc = require_corpus(h); if (rc) return rc;
    HIPC(h, hipSetDevice(h->device));
    HIPC(h, mvhdp_launch_build_counts(h->mm, h->N, h->stream));
    if (h->delta_pending) {
        // a NO_APPLY sweep's deltas were never applied: z already holds its assignments, so the recount above includes
        // them -- drop the deltas instead of leaving them to be added on top
        HIPC(h, hipMemsetAsync(h->mm.delta, 0, (size_t)counts_len(h) * sizeof(int32_t), h->stream));
        h->delta_pending = false; h->delta_clean = true;
    }
    HIPC(h, hipStreamSynchronize(h->stream));
    h->have_counts = true; h->have_trees = false;
    return MVHDP_OK;
}

// The trees are current (have_trees) but the last sweep refreshed only the descent table: write the FTree.tree
// arrays as well, from the same counts and hyper-parameters (nothing has changed them since, or have_trees were false).
static int ensure_full_trees(mvhdp_ctx* h)
{
    if (!h->have_trees || h->full_trees) return MVHDP_OK;
    HIPC(h, mvhdp_launch_build_trees(h->mm, h->trees_inference, true, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    h->full_trees = true;
    return MVHDP_OK;
}

extern "C" int mvhdp_build_trees(mvhdp_handle h)
{
    CHECK_H(h);
    if (!h->have_hyper) FAIL(h, MVHDP_ERR_STATE, "build_trees before set_hyper");
    if (!h->have_counts) FAIL(h, MVHDP_ERR_STATE, "build_trees before build_counts/set_counts");
    HIPC(h, hipSetDevice(h->device));
    HIPC(h, mvhdp_launch_build_trees(h->mm, false, true, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    h->have_trees = true; h->full_trees = true; h->trees_inference = false;
    return MVHDP_OK;
}

extern "C" int mvhdp_build_inference_trees(mvhdp_handle h)
{
    CHECK_H(h);
    if (!h->have_hyper) FAIL(h, MVHDP_ERR_STATE, "build_inference_trees before set_hyper");
    if (!h->have_counts) FAIL(h, MVHDP_ERR_STATE, "build_inference_trees before build_counts/set_counts");
    HIPC(h, hipSetDevice(h->device));
    HIPC(h, mvhdp_launch_build_trees(h->mm, true, true, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    h->have_trees = true; h->full_trees = true; h->trees_inference = true;
    return MVHDP_OK;
}

extern "C" int mvhdp_init_assignments_from_trees(mvhdp_handle h, uint64_t seed)
{
    CHECK_H(h);
    int rc = require_corpus(h); if (rc) return rc;
    if (!h->have_trees) FAIL(h, MVHDP_ERR_STATE, "init_assignments_from_trees before build_trees/build_inference_trees");
    HIPC(h, hipSetDevice(h->device));
    HIPC(h, mvhdp_launch_init_from_trees(h->mm, (uint32_t)seed, (uint32_t)(seed >> 32), h->stream));     // reads the descent table only
    HIPC(h, hipStreamSynchronize(h->stream));
    h->rmax_hint = 0;
    return MVHDP_OK;
}

extern "C" int mvhdp_get_counts(mvhdp_handle h, int32_t m, int32_t* n_wk, int32_t* n_k)
{
    CHECK_H(h);
    MvModel& mm = h->mm;
    if (m < 0 || m >= mm.M) FAIL(h, MVHDP_ERR_INVALID_ARG, "get_counts: bad view");
    HIPC(h, hipSetDevice(h->device));
    HIPC(h, hipStreamSynchronize(h->stream));
    const int K = mm.K;
    if (n_wk) HIPC(h, hipMemcpy(n_wk, mm.counts + mm.rowbase[m] * K, (size_t)mm.V[m] * K * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (n_k) HIPC(h, hipMemcpy(n_k, mm.counts + mm.rowbase[mm.M] * K + (int64_t)m * K, (size_t)K * sizeof(int32_t), hipMemcpyDeviceToHost));
    return MVHDP_OK;
}

extern "C" int mvhdp_set_counts(mvhdp_handle h, int32_t m, const int32_t* n_wk, const int32_t* n_k)
{
    CHECK_H(h);
    MvModel& mm = h->mm;
    if (m < 0 || m >= mm.M) FAIL(h, MVHDP_ERR_INVALID_ARG, "set_counts: bad view");
    HIPC(h, hipSetDevice(h->device));
    HIPC(h, hipStreamSynchronize(h->stream));
    const int K = mm.K;
    if (n_wk) HIPC(h, hipMemcpy(mm.counts + mm.rowbase[m] * K, n_wk, (size_t)mm.V[m] * K * sizeof(int32_t), hipMemcpyHostToDevice));
    if (n_k) HIPC(h, hipMemcpy(mm.counts + mm.rowbase[mm.M] * K + (int64_t)m * K, n_k, (size_t)K * sizeof(int32_t), hipMemcpyHostToDevice));
    h->have_counts = true; h->have_trees = false;
    return MVHDP_OK;
}

extern "C" int mvhdp_get_tree(mvhdp_handle h, int32_t m, int32_t type, double* tree)
{
    CHECK_H(h);
    MvModel& mm = h->mm;
    if (m < 0 || m >= mm.M || type < 0 || type >= mm.V[m] || !tree) FAIL(h, MVHDP_ERR_INVALID_ARG, "get_tree: bad argument");
    if (!h->have_trees) FAIL(h, MVHDP_ERR_STATE, "get_tree before build_trees");
    HIPC(h, hipSetDevice(h->device));
    { int rc2 = ensure_full_trees(h); if (rc2) return rc2; }
    HIPC(h, hipStreamSynchronize(h->stream));
    HIPC(h, hipMemcpy(tree, mm.trees + (mm.rowbase[m] + type) * 2 * mm.K, (size_t)2 * mm.K * sizeof(double), hipMemcpyDeviceToHost));
    return MVHDP_OK;
}

extern "C" int mvhdp_get_doc_topic_hist(mvhdp_handle h, int32_t m, int32_t* hist, int32_t hist_len,
                                        int32_t* doc_len_counts, int32_t len_len)
{
    CHECK_H(h);
    MvModel& mm = h->mm;
    if (m < 0 || m >= mm.M || (hist && hist_len < 1) || (doc_len_counts && len_len < 1))
        FAIL(h, MVHDP_ERR_INVALID_ARG, "get_doc_topic_hist: bad argument");
    int rc = require_corpus(h); if (rc) return rc;
    HIPC(h, hipSetDevice(h->device));
    int32_t *d_hist = nullptr, *d_len = nullptr;
    hipError_t e = hipSuccess;
    if (hist) e = hipMalloc(&d_hist, (size_t)mm.K * hist_len * sizeof(int32_t));
    if (e == hipSuccess && doc_len_counts) e = hipMalloc(&d_len, (size_t)len_len * sizeof(int32_t));
    if (e == hipSuccess) e = mvhdp_launch_doc_topic_hist(mm, m, d_hist, hist_len, d_len, len_len, h->stream);
    if (e == hipSuccess && hist) e = hipMemcpy(hist, d_hist, (size_t)mm.K * hist_len * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && doc_len_counts) e = hipMemcpy(doc_len_counts, d_len, (size_t)len_len * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (d_hist) hipFree(d_hist);
    if (d_len) hipFree(d_len);
    HIPC(h, e);
    return MVHDP_OK;
}

// Largest entity (sizes the slot list) and the work-queue order: longest entities first,
// so that the tail of the sweep is made of short ones (power-law lengths).
static int64_t compute_max_doc_tokens(mvhdp_ctx* h)
{
    if (h->max_doc_tokens >= 0) return h->max_doc_tokens;
    int64_t mx = 0, mn = INT64_MAX;
    const MvModel& mm = h->mm;
    std::vector<int64_t> tot((size_t)mm.D);
    for (int64_t d = 0; d < mm.D; d++) {
        int64_t t = 0;
        for (int m = 0; m < mm.M; m++) t += h->h_doc_off[m][d + 1] - h->h_doc_off[m][d];
        tot[d] = t;
        mx = std::max(mx, t); mn = std::min(mn, t);
    }
    if (h->d_doc_order) { hipFree(h->d_doc_order); h->d_doc_order = nullptr; }
    h->tokens_desc.clear();
    if (mm.D > 0 && mm.D < (1LL << 31)) {
        // counting sort by decreasing length (stable: ties keep entity order)
        std::vector<int64_t> start((size_t)mx + 2, 0);
        for (int64_t d = 0; d < mm.D; d++) start[(size_t)(mx - tot[d]) + 1]++;
        for (size_t i = 1; i < start.size(); i++) start[i] += start[i - 1];
        std::vector<int32_t> order((size_t)mm.D);
        for (int64_t d = 0; d < mm.D; d++) order[(size_t)start[(size_t)(mx - tot[d])]++] = (int32_t)d;
        // without the order on the device the sweep runs in natural entity order (correct, only less balanced)
        if (hipMalloc(&h->d_doc_order, (size_t)mm.D * sizeof(int32_t)) != hipSuccess) h->d_doc_order = nullptr;
        else if (hipMemcpy(h->d_doc_order, order.data(), (size_t)mm.D * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
            hipFree(h->d_doc_order); h->d_doc_order = nullptr;
        } else {
            h->tokens_desc.resize((size_t)mm.D);
            for (int64_t q = 0; q < mm.D; q++) h->tokens_desc[(size_t)q] = tot[(size_t)order[(size_t)q]];
        }
    }
    (void)mn;
    h->max_doc_tokens = mx;
    return mx;
}

// UPD:263-270: the topic leaves inActiveTopicIndex and its alpha[m][k] takes alpha[m][K]
static int apply_activation(mvhdp_ctx* h, int32_t activated_topic, int32_t activated_modality)
{
    MvModel& mm = h->mm;
    if (activated_topic < 0) return MVHDP_OK;
    if (activated_topic >= mm.K || activated_modality < 0 || activated_modality >= mm.M)
        FAIL(h, MVHDP_ERR_INVALID_ARG, "apply_delta: bad activation");
    if (h->h_inactive[activated_topic]) {
        h->h_inactive[activated_topic] = 0;
        h->h_alpha[(size_t)activated_modality * (mm.K + 1) + activated_topic] = h->h_alpha[(size_t)activated_modality * (mm.K + 1) + mm.K];
        mm.first_inactive = -1;
        for (int k = 0; k < mm.K; k++) if (h->h_inactive[k]) { mm.first_inactive = k; break; }
        HIPC(h, hipMemcpy(h->d_alpha, h->h_alpha.data(), h->h_alpha.size() * sizeof(double), hipMemcpyHostToDevice));
        HIPC(h, hipMemcpy(h->d_inactive, h->h_inactive.data(), (size_t)mm.K, hipMemcpyHostToDevice));
    }
    return MVHDP_OK;
}

extern "C" int mvhdp_apply_delta(mvhdp_handle h, int32_t activated_topic, int32_t activated_modality)
{
    CHECK_H(h);
    MvModel& mm = h->mm;
    HIPC(h, hipSetDevice(h->device));
    HIPC(h, hipMemsetAsync(h->d_stats + ST_NEGATIVE, 0, sizeof(unsigned long long), h->stream));
    HIPC(h, mvhdp_launch_apply_delta(mm, h->d_stats, h->stream));
    unsigned long long neg = 0;
    HIPC(h, hipMemcpyAsync(&neg, h->d_stats + ST_NEGATIVE, sizeof neg, hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    h->have_trees = false;
    h->delta_clean = true;                                       // apply_delta_kernel zeroes what it adds
    h->delta_pending = false;
    { int rc = apply_activation(h, activated_topic, activated_modality); if (rc) return rc; }
    if (neg) FAIL(h, MVHDP_ERR_NEGATIVE_COUNT, "a topic count went below zero (UPD:202-215)");
    return MVHDP_OK;
}

// ---- the multi-GPU pipeline: apply + tree rebuild by row ranges, stream-ordered (see include/mvhdp.h) ----
extern "C" int mvhdp_apply_delta_begin(mvhdp_handle h)
{
    CHECK_H(h);
    if (!h->have_hyper || !h->have_counts) FAIL(h, MVHDP_ERR_STATE, "apply_delta_begin before set_hyper / counts");
    HIPC(h, hipSetDevice(h->device));
    HIPC(h, hipMemsetAsync(h->d_stats + ST_NEGATIVE, 0, sizeof(unsigned long long), h->stream));
    HIPC(h, mvhdp_launch_apply_nk(h->mm, h->d_stats + ST_NEGATIVE, h->stream));
    h->rows_applied = 0;
    h->have_trees = false;
    return MVHDP_OK;
}

extern "C" int mvhdp_apply_delta_rows(mvhdp_handle h, int64_t row_begin, int64_t row_end)
{
    CHECK_H(h);
    const int64_t nrows = h->mm.rowbase[h->mm.M];
    if (h->rows_applied < 0) FAIL(h, MVHDP_ERR_STATE, "apply_delta_rows outside an apply_delta_begin / apply_delta_end bracket");
    if (row_begin < 0 || row_end > nrows || row_begin > row_end) FAIL(h, MVHDP_ERR_INVALID_ARG, "apply_delta_rows: bad row range");
    HIPC(h, hipSetDevice(h->device));
    HIPC(h, mvhdp_launch_build_trees_rows(h->mm, false, h->last_need_full, row_begin, row_end, true, h->d_stats + ST_NEGATIVE, h->stream));
    h->rows_applied += row_end - row_begin;
    return MVHDP_OK;
}

extern "C" int mvhdp_apply_delta_end(mvhdp_handle h, int32_t activated_topic, int32_t activated_modality)
{
    CHECK_H(h);
    const int64_t nrows = h->mm.rowbase[h->mm.M];
    if (h->rows_applied != nrows) { h->rows_applied = -1; FAIL(h, MVHDP_ERR_STATE, "apply_delta_end: the row ranges applied do not cover every row exactly once"); }
    h->rows_applied = -1;
    HIPC(h, hipSetDevice(h->device));
    unsigned long long neg = 0;
    HIPC(h, hipMemcpyAsync(&neg, h->d_stats + ST_NEGATIVE, sizeof neg, hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    h->delta_clean = true; h->delta_pending = false;
    // the trees were rebuilt from the updated counts row by row: current, unless an activation now changes alpha
    h->have_trees = true; h->full_trees = h->last_need_full; h->trees_inference = false;
    if (activated_topic >= 0) {
        h->have_trees = false;
        int rc = apply_activation(h, activated_topic, activated_modality);
        if (rc) return rc;
    }
    if (neg) FAIL(h, MVHDP_ERR_NEGATIVE_COUNT, "a topic count went below zero (UPD:202-215)");
    return MVHDP_OK;
}

extern "C" int mvhdp_trees_current(mvhdp_handle h)
{
    CHECK_H(h);
    return h->have_trees ? 1 : 0;
}

extern "C" int mvhdp_sweep(mvhdp_handle h, uint32_t sweep_idx, uint64_t seed, uint32_t flags,
                           const double* p_override, const mvhdp_debug* dbg, mvhdp_sweep_stats* stats)
{
    CHECK_H(h);
    MvModel& mm = h->mm;
    int rc = require_corpus(h); if (rc) return rc;
    if (!h->have_hyper) FAIL(h, MVHDP_ERR_STATE, "sweep before set_hyper");
    if (!h->have_counts) FAIL(h, MVHDP_ERR_STATE, "sweep before build_counts/set_counts");
    if (flags & MVHDP_SWEEP_FROZEN) flags |= MVHDP_SWEEP_REUSE_TREES | MVHDP_SWEEP_NO_APPLY;   // nut == 0: the model is read-only
    if ((flags & MVHDP_SWEEP_REUSE_TREES) && !h->have_trees) FAIL(h, MVHDP_ERR_STATE, "REUSE_TREES / FROZEN without trees");
    if (flags & ~(MVHDP_SWEEP_REUSE_TREES | MVHDP_SWEEP_NO_APPLY | MVHDP_SWEEP_EXACT_CHAIN | MVHDP_SWEEP_GENERIC_KERNEL | MVHDP_SWEEP_FROZEN |
                  MVHDP_SWEEP_LIVE | MVHDP_SWEEP_LIVE_SEGMENTS(0xff) | MVHDP_SWEEP_SEGMENT_APPLY)) FAIL(h, MVHDP_ERR_INVALID_ARG, "sweep: unknown flag");
    const bool live = (flags & MVHDP_SWEEP_LIVE) != 0;
    const bool seg_apply = (flags & MVHDP_SWEEP_SEGMENT_APPLY) != 0;
    if (seg_apply && (flags & (MVHDP_SWEEP_LIVE | MVHDP_SWEEP_NO_APPLY | MVHDP_SWEEP_FROZEN | MVHDP_SWEEP_REUSE_TREES)))
        FAIL(h, MVHDP_ERR_INVALID_ARG, "sweep: SEGMENT_APPLY excludes LIVE, NO_APPLY, FROZEN and REUSE_TREES");
    if (live && (flags & MVHDP_SWEEP_FROZEN)) FAIL(h, MVHDP_ERR_INVALID_ARG, "sweep: LIVE and FROZEN exclude each other");
    if (h->rows_applied >= 0) FAIL(h, MVHDP_ERR_STATE, "sweep: an mvhdp_apply_delta_begin bracket is open (call mvhdp_apply_delta_end)");
    if (h->delta_pending && !(flags & MVHDP_SWEEP_FROZEN))
        FAIL(h, MVHDP_ERR_STATE, "sweep: the previous NO_APPLY sweep's deltas have not been applied (mvhdp_apply_delta)");
    // live sweep: the entities are cut into nseg interleaved segments of the longest-first order, the trees are rebuilt
    // from the live counts before each
    // (a deferred sweep accepts a segment count too: same integers as one segment, the trees being those of the snapshot)
    int nseg = (int)((flags >> 16) & 0xffu);
    if (nseg == 0) nseg = (live || seg_apply) ? 4 : 1;
    if ((int64_t)nseg > mm.D) nseg = (int)std::max<int64_t>(1, mm.D);
    const int K = mm.K, M = mm.M;
    HIPC(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;

    // launch geometry
    int64_t mdt = compute_max_doc_tokens(h);
    int S_cap = (int)std::min<int64_t>(K, std::max<int64_t>(mdt, 1));
    S_cap = (S_cap + 63) / 64 * 64;
    SweepLaunch sl{};
    sl.sweep_idx = sweep_idx; sl.seed_lo = (uint32_t)seed; sl.seed_hi = (uint32_t)(seed >> 32);
    sl.flags = flags & 0xffffu; sl.S_cap = S_cap;
    sl.q_order_stride = 1;
    // the block's private n_k delta table: in LDS up to 24 KiB (C5: 20 KB), beyond that (e.g. K = 2048 with 8 views:
    // 64 KB, which would not leave room for the slot state) the deltas go straight to the delta buffer.  A live sweep
    // keeps the private table too: M*K hot words would take every token's two atomics one after the other at the
    // memory side (measured on C3: 28 ms per sweep instead of 5.7), so tokensPerTopic becomes current at each
    // segment end -- together with the trees -- while n_wk is updated in place (UPD:197-207).
    sl.nk_global = ((size_t)M * K * sizeof(int) > 24 * 1024) ? 1 : 0;
    sl.block_shared_bytes = (uint32_t)((((size_t)(sl.nk_global ? 0 : M * K) + MVHDP_HIST_BINS + MVHDP_MAXM * MVHDP_VIEW_STATS) * sizeof(int) + 15) & ~(size_t)15);
    const bool debug = dbg != nullptr;
    // Primary kernel variant: the register-resident kernel with 64*rmax topic slots per entity that is
    // cheapest for the topic-list histogram of the previous sweep (first time: of a probe pass over z).
    bool fast = !(flags & MVHDP_SWEEP_GENERIC_KERNEL);
    int rmax = 0;
    if (fast) {
        if (h->rmax_hint <= 0) {
            // first sweep on these assignments: measure the topic lists (one pass over z)
            unsigned long long hist[1 + MVHDP_HIST_BINS] = {0};
            HIPC(h, hipMemsetAsync(h->d_ovf_meta, 0, 256, h->stream));
            HIPC(h, mvhdp_launch_slot_hist(mm, (unsigned long long*)(h->d_ovf_meta + 2), h->stream));
            HIPC(h, hipMemcpyAsync(hist, h->d_ovf_meta, sizeof hist, hipMemcpyDeviceToHost, h->stream));
            HIPC(h, hipStreamSynchronize(h->stream));
            std::copy(hist + 1, hist + 1 + MVHDP_HIST_BINS, h->last_hist);
            h->rmax_hint = rmax_from_hist(hist + 1);
        }
        rmax = h->rmax_hint;
        if (rmax == 1 && (nseg != 1 || dbg || (flags & (MVHDP_SWEEP_FROZEN | MVHDP_SWEEP_EXACT_CHAIN)))) {
            // an early 1-round proposal (up to half of the tokens still in longer lists) stands only where the clock can confirm it;
            // this sweep cannot be timed against its neighbours (segments, debug, frozen): the old rule, at most 0.5 % beyond
            double tot = 0, beyond = 0;
            for (int b = 0; b < MVHDP_HIST_BINS; b++) { tot += (double)h->last_hist[b]; if (b >= 1) beyond += (double)h->last_hist[b]; }
            if (beyond > 0.005 * tot && S_cap > 64) rmax = 2;
        }
        if (const char* f = getenv("MVHDP_FORCE_RMAX")) { int v = atoi(f); if (v >= 1 && v <= 16) rmax = v; }   // diagnostics only
        if (rmax > 16) fast = false;
        else {
            { int v = 1; while (v < rmax) v <<= 1; rmax = v; }              // variants exist for 1, 2, 4, 8, 16
            while (rmax > 1 && 64 * (rmax / 2) >= S_cap) rmax /= 2;          // no larger than the corpus can need
        }
    }
    // Two ways to deal with the entities whose topic list exceeds the primary variant's 64*rmax slots.
    //  * optimistic (few of them): the primary kernel runs over every entity and appends the ones that do
    //    not fit to an overflow list; the 8-round, the 16-round and the generic kernel then take the rest,
    //    one pass after another.
    //  * classified (many of them, e.g. power-law lengths with K = 1000): only an entity with more tokens
    //    than slots can overflow -- a static prefix of the longest-first order.  classify_kernel measures
    //    that prefix and lists every entity for the narrowest kernel that holds it; the kernels of all
    //    classes then run side by side on their own streams, the widest (longest entities) first, so the
    //    long sequential chains overlap the bulk instead of following it.
    int pc = 0;                                              // class of the primary variant: rmax == 1 << pc
    while ((1 << pc) < rmax) pc++;
    // walk thresholds of this sweep (see mvhdp_ctx::walk_i) and the kernel flavour that goes with them
    const char* theta_env = getenv("MVHDP_WALK_THETA");         // diagnostics: "t0,t1,..." fixes the thresholds
    {   // the 1-round variant and the wider ones want different thresholds: each keeps its own
        const int cls = (fast && rmax == 1) ? 0 : 1;
        if (cls != h->walk_cls) {
            h->walk_i_by[h->walk_cls] = h->walk_i;
            if (h->walk_i_by[cls] >= 0) h->walk_i = h->walk_i_by[cls];
            h->walk_cls = cls;
            h->walk_phase = 0; h->walk_far = false;
        }
    }
    auto walk_controlled = [&](int m) { return h->walk_f[m] >= 0.0 && h->walk_f[m] < 0.35; };
    bool walk_any = false, walk_unknown = false;                 // (walk_f < 0: not measured yet)
    for (int m = 0; m < M; m++) { walk_any = walk_any || walk_controlled(m); walk_unknown = walk_unknown || h->walk_f[m] < 0.0; }
    if (theta_env) {
        int m = 0;
        for (const char* q = theta_env; *q && m < MVHDP_MAXM; m++) {
            sl.walk_theta[m] = atof(q);
            while (*q && *q != ',') q++;
            if (*q == ',') q++;
        }
        for (; m < MVHDP_MAXM; m++) sl.walk_theta[m] = 0.0;
        sl.walk = 1;
    } else {
        const int top = MVHDP_WALK_BINS;                         // thresholds up to 1 (= no token of the view walked up front)
        h->walk_probe_i = h->walk_i;
        if (h->walk_phase == 1 && walk_any) {
            if (h->walk_far && h->walk_i < 4) h->walk_far = false;
            if (h->walk_far) h->walk_dir = -1;
            if (h->walk_dir > 0 && h->walk_i >= top) h->walk_dir = -1;
            if (h->walk_dir < 0 && h->walk_i <= 0) h->walk_dir = 1;
            if (h->walk_dir > 0) {
                int j = 1;
                double extra = h->walk_hist[h->walk_i];
                while (h->walk_i + j < top && j < h->walk_maxj && extra + h->walk_hist[h->walk_i + j] <= h->walk_cap) { extra += h->walk_hist[h->walk_i + j]; j++; }
                h->walk_probe_i = h->walk_i + j;
            } else h->walk_probe_i = h->walk_far ? h->walk_i / 2 : h->walk_i - 1;
        }
        sl.walk = 0;
        for (int m = 0; m < MVHDP_MAXM; m++) {
            sl.walk_theta[m] = (m < M && walk_controlled(m)) ? (double)h->walk_probe_i / MVHDP_WALK_BINS : 0.0;
            if (sl.walk_theta[m] > 0.0) sl.walk = 1;
        }
        // no view qualifies: look again every 16th sweep (the statistics come from the walk flavour only)
        if (!walk_any && h->sweeps_done >= h->walk_refresh_at) { sl.walk = 1; h->walk_refresh_at = h->sweeps_done + 16; }
        if (walk_unknown) sl.walk = 1;                           // the first sweep measures (threshold 0)
    }
    if (debug) sl.walk = 1;
    // the 16-bit mirror of n_wk (written with the trees) for the 1-round walk flavour -- the bandwidth-bound one: half the lines
    // of every gathered row.  Needs the mirror to be this sweep's start counts: trees built in this call or still current, and
    // no live updates (the mirror is a snapshot).
    sl.narrow = (fast && rmax == 1 && sl.walk && !debug && !live &&
                 (!(flags & MVHDP_SWEEP_REUSE_TREES) || h->have_trees)) ? 1 : 0;
    if (const char* f = getenv("MVHDP_NARROW")) sl.narrow = (sl.narrow && atoi(f) != 0) ? 1 : 0;     // diagnostics: 0 switches it off
    int64_t H = 0;                                           // entities that may exceed the primary variant
    bool classified = false;
    if (fast && h->d_doc_order && !h->tokens_desc.empty()) {
        H = std::upper_bound(h->tokens_desc.begin(), h->tokens_desc.end(), (int64_t)64 * rmax, std::greater<int64_t>()) - h->tokens_desc.begin();
        double tot = 0, beyond = 0;
        for (int b = 0; b < MVHDP_HIST_BINS; b++) { tot += (double)h->last_hist[b]; if (b + 1 > rmax) beyond += (double)h->last_hist[b]; }
        classified = H > 0 && beyond > 0.005 * tot;
        if (const char* f = getenv("MVHDP_FORCE_MODE")) {                 // diagnostics / tests
            if (!strcmp(f, "classified")) classified = H > 0;
            else if (!strcmp(f, "optimistic")) classified = false;
        }
    }
    int chain[3] = {0, 0, 0}, n_chain = 0;
    if (fast && !classified) {
        chain[n_chain++] = rmax;
        if (rmax < 8 && S_cap > 64 * rmax) chain[n_chain++] = 8;
        if (rmax < 16 && S_cap > 512) chain[n_chain++] = 16;
    }
    struct Geo { uint32_t wave_bytes; int wpb; size_t lds; int grid; };
    auto geometry = [&](bool is_fast, int r, Geo& g) -> int {
        // a register-resident variant never holds more than 64*r slots (longer lists are diverted before any slot write)
        g.wave_bytes = (uint32_t)(is_fast ? mvhdp_sweep_fast_wave_bytes(M, std::min(S_cap, 64 * r), r) : mvhdp_sweep_wave_bytes(M, S_cap));
        g.wpb = 4;
        while (g.wpb > 1 && sl.block_shared_bytes + (size_t)g.wpb * g.wave_bytes > h->max_lds) g.wpb >>= 1;
        g.lds = sl.block_shared_bytes + (size_t)g.wpb * g.wave_bytes;
        if (g.lds > h->max_lds) return MVHDP_ERR_UNSUPPORTED;
        int bpc = is_fast ? mvhdp_sweep_fast_occupancy(r, debug, sl.walk != 0, 64 * g.wpb, g.lds) : mvhdp_sweep_generic_occupancy(debug, 64 * g.wpb, g.lds);
        if (bpc < 1) bpc = 1;
        int64_t need = (mm.D + (int64_t)g.wpb * MVHDP_DOC_BATCH - 1) / ((int64_t)g.wpb * MVHDP_DOC_BATCH);
        g.grid = (int)std::max<int64_t>(1, std::min<int64_t>(need, (int64_t)h->num_cus * bpc));
        return MVHDP_OK;
    };
    Geo gen{}, fst[3] = {}, cls[MVHDP_N_CLASSES] = {};
    bool cls_fast[MVHDP_N_CLASSES] = {};
    {
        // the generic kernel may need > 64 KiB of dynamic LDS
        uint32_t wb = (uint32_t)mvhdp_sweep_wave_bytes(M, S_cap);
        int wpb = 4;
        while (wpb > 1 && sl.block_shared_bytes + (size_t)wpb * wb > h->max_lds) wpb >>= 1;
        size_t lds = sl.block_shared_bytes + (size_t)wpb * wb;
        if (lds > h->max_lds) FAIL(h, MVHDP_ERR_UNSUPPORTED, "per-entity LDS state exceeds 160 KiB (K * modalities too large)");
        if (lds > 65536 && lds > h->lds_attr_set) { HIPC(h, mvhdp_sweep_set_max_lds(lds)); h->lds_attr_set = lds; }
    }
    if (geometry(false, 0, gen) != MVHDP_OK) FAIL(h, MVHDP_ERR_UNSUPPORTED, "per-entity LDS state exceeds 160 KiB");
    if (fast && geometry(true, rmax, fst[0]) != MVHDP_OK) { fast = false; classified = false; n_chain = 0; }
    for (int p = 1; p < n_chain; p++)
        if (geometry(true, chain[p], fst[p]) != MVHDP_OK) { n_chain = p; break; }     // later passes fall to the generic kernel
    if (classified) {
        cls[pc] = fst[0]; cls_fast[pc] = true;
        for (int c = pc + 1; c < MVHDP_N_CLASSES; c++) {
            cls_fast[c] = c < 5 && geometry(true, 1 << c, cls[c]) == MVHDP_OK;
            if (!cls_fast[c]) cls[c] = gen;                                           // no room for that variant: generic kernel
        }
        if (!h->d_lists) HIPC(h, hipMalloc(&h->d_lists, (size_t)MVHDP_N_CLASSES * mm.D * sizeof(int32_t)));
        if (!h->ev_fork) HIPC(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    } else if (fast) {
        if (!h->d_overflow && mm.D > 0) HIPC(h, hipMalloc(&h->d_overflow, (size_t)mm.D * sizeof(int32_t)));
        if (n_chain > 1 && !h->d_overflow2 && mm.D > 0) HIPC(h, hipMalloc(&h->d_overflow2, (size_t)mm.D * sizeof(int32_t)));
    }
    if (getenv("MVHDP_DEBUG")) {
        double tot = 0, b1 = 0, b2 = 0;                           // token share by topic-list size (last sweep): <= 64, <= 128 slots
        for (int b = 0; b < MVHDP_HIST_BINS; b++) { tot += (double)h->last_hist[b]; if (b < 1) b1 += (double)h->last_hist[b]; if (b < 2) b2 += (double)h->last_hist[b]; }
        fprintf(stderr, "[mvhdp] sweep %u: fast=%d rmax=%d (hint %d) %s H=%lld chain=%d,%d,%d S_cap=%d | fast grid=%d wpb=%d lds=%zu | generic grid=%d wpb=%d lds=%zu | tokens in lists <=64: %.4f <=128: %.4f\n",
                sweep_idx, (int)fast, rmax, h->rmax_hint, classified ? "classified" : "optimistic", (long long)H, chain[0], chain[1], chain[2], S_cap,
                fst[0].grid, fst[0].wpb, fst[0].lds, gen.grid, gen.wpb, gen.lds, b1 / std::max(1.0, tot), b2 / std::max(1.0, tot));
    }
    sl.stats = h->d_stats;
    sl.act_key = h->d_act_key;
    // d_ovf_meta (u32 words): 0,1 = overflow counts of passes 1,2; 2..35 = u64 hist[17]; 36 = overflow count of pass 3;
    // 40..45 = entities per class (classified mode)
    unsigned int* class_counts = h->d_ovf_meta + 40;
    sl.doc_counter = h->d_doc_counter;
    sl.q_list = nullptr; sl.q_list_count = nullptr;
    sl.q_order = h->d_doc_order; sl.q_order_start = 0; sl.q_order_count = mm.D;      // default: every entity, longest first
    sl.overflow_list = nullptr; sl.overflow_count = nullptr;
    sl.slot_hist = (unsigned long long*)(h->d_ovf_meta + 2);
    // debug buffers
    std::vector<void*> to_free;
    auto cleanup = [&]() { for (void* p : to_free) hipFree(p); };
    if (debug) {
        for (int m = 0; m < M; m++) {
            if (dbg->tok_dbg[m] && h->N[m] > 0) {
                void* p = nullptr;
                hipError_t e = hipMalloc(&p, (size_t)h->N[m] * 4 * sizeof(double));
                if (e != hipSuccess) { cleanup(); HIPC(h, e); }
                to_free.push_back(p);
                hipMemsetAsync(p, 0, (size_t)h->N[m] * 4 * sizeof(double), s);
                sl.tok_dbg[m] = (double*)p;
            }
        }
        if (dbg->n_trace > 0) {
            void *a = nullptr, *b = nullptr, *c = nullptr, *o = nullptr;
            size_t n = (size_t)dbg->n_trace;
            if (hipMalloc(&a, n * 8) != hipSuccess || hipMalloc(&b, n * 4) != hipSuccess ||
                hipMalloc(&c, n * 4) != hipSuccess || hipMalloc(&o, n * (K + 1) * 8) != hipSuccess) {
                cleanup(); FAIL(h, MVHDP_ERR_HIP, "debug trace allocation failed");
            }
            to_free.push_back(a); to_free.push_back(b); to_free.push_back(c); to_free.push_back(o);
            hipMemcpyAsync(a, dbg->trace_doc, n * 8, hipMemcpyHostToDevice, s);
            hipMemcpyAsync(b, dbg->trace_view, n * 4, hipMemcpyHostToDevice, s);
            hipMemcpyAsync(c, dbg->trace_pos, n * 4, hipMemcpyHostToDevice, s);
            hipMemsetAsync(o, 0, n * (K + 1) * 8, s);
            sl.n_trace = dbg->n_trace;
            sl.trace_doc = (const int64_t*)a; sl.trace_view = (const int32_t*)b; sl.trace_pos = (const int32_t*)c;
            sl.trace_out = (double*)o;
        }
    }

    hipError_t e = hipSuccess;
    auto step = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    step(hipEventRecord(h->ev[0], s));
    if (M > 1) {
        if (!mm.p && mm.D > 0) step(hipMalloc(&mm.p, (size_t)mm.D * M * M * sizeof(double)));
        if (e == hipSuccess) {
            if (p_override) step(hipMemcpyAsync(mm.p, p_override, (size_t)mm.D * M * M * sizeof(double), hipMemcpyHostToDevice, s));
            else step(mvhdp_launch_draw_p(mm, sweep_idx, sl.seed_lo, sl.seed_hi, s));
        }
    }
    // FTree.tree itself is read by the generic kernel (and the debug trace) only: when no entity can reach it, the
    // rebuild refreshes just the descent table (0.13 instead of 0.24 ms at C4)
    const bool generic_reachable = S_cap > 1024 || mdt > 65535;      // lists beyond the 16-round variant, or views beyond its 16-bit counts
    bool need_full = !fast || debug || generic_reachable;
    if (fast && classified) {
        for (int c = pc + 1; c < MVHDP_N_CLASSES; c++)
            if (!(S_cap <= (32 << c) && !(c == 5 && mdt > 65535)) && !cls_fast[c]) need_full = true;
    } else if (fast) {
        need_full = need_full || n_chain == 0 || S_cap > 64 * chain[n_chain - 1];
    }
    h->last_need_full = need_full;
    auto rebuild_trees = [&]() {
        step(mvhdp_launch_build_trees(mm, false, need_full, s));
        h->have_trees = true; h->full_trees = need_full; h->trees_inference = false;
    };
    if (!(flags & MVHDP_SWEEP_REUSE_TREES)) rebuild_trees();
    else if (need_full && !h->full_trees) {
        step(mvhdp_launch_build_trees(mm, h->trees_inference, true, s));
        h->full_trees = true;
    }
    // where the sweep's atomics land: the delta replica (deferred), or the shared counts themselves (live)
    MvModel mk = mm;
    if (live) {
        mk.delta = mm.counts;
        if (flags & MVHDP_SWEEP_NO_APPLY) { step(mvhdp_launch_live_helper(mm, 0, h->d_stats, s)); h->delta_clean = false; }
    } else if (!(flags & MVHDP_SWEEP_FROZEN)) {                  // a frozen sweep queues nothing (WRK:587) and leaves the buffer alone
        if (!h->delta_clean) step(hipMemsetAsync(mm.delta, 0, (size_t)counts_len(h) * sizeof(int32_t), s));
        h->delta_clean = false;
    }
    step(hipMemsetAsync(h->d_stats, 0, ST_COUNT * sizeof(unsigned long long), s));
    const long long kmax = LLONG_MAX;
    step(hipMemcpyAsync(h->d_act_key, &kmax, sizeof kmax, hipMemcpyHostToDevice, s));
    step(hipMemsetAsync(h->d_ovf_meta, 0, 256, s));
    step(hipEventRecord(h->ev[1], s));
    unsigned long long ovf[1 + MVHDP_HIST_BINS] = {0};      // [0]: two u32 overflow counts, [1..17]: token histogram
    static const int ovf_word[3] = {0, 1, 2 + 2 * MVHDP_HIST_BINS};
    auto blocks_for = [&](int64_t n, const Geo& g) {
        int64_t need = (n + (int64_t)g.wpb * MVHDP_DOC_BATCH - 1) / ((int64_t)g.wpb * MVHDP_DOC_BATCH);
        return (int)std::max<int64_t>(1, std::min<int64_t>(need, g.grid));
    };
    // entities of the longest-first order with more than `tokens` tokens: a prefix of that order (0 without the order)
    auto longer_than = [&](int64_t tokens) -> int64_t {
        if (h->tokens_desc.empty()) return mm.D;
        return std::upper_bound(h->tokens_desc.begin(), h->tokens_desc.end(), tokens, std::greater<int64_t>()) - h->tokens_desc.begin();
    };
    int n_activations = 0;                                   // topics activated at segment borders, and the key of the first
    long long first_act = LLONG_MAX;
    for (int seg = 0; seg < nseg && e == hipSuccess && mm.D > 0; seg++) {
        // segment seg = positions seg, seg + nseg, ... of the order; a prefix [0, P) of the order holds share(P) of them
        auto share = [&](int64_t P) -> int64_t { return P > seg ? (P - seg + nseg - 1) / nseg : 0; };
        const int64_t n_seg = share(mm.D);
        if (seg > 0) {
            // UPD:263-270 acts as soon as a delta lands on an inactive topic, and the samplers then draw the NEXT inactive
            // index (WRK:523-526): a sweep whose counts are kept current does the same at every segment border -- the
            // segment's first such delta (by entity, view, position) activates its topic before the next segment starts.
            // Not with MVHDP_SWEEP_NO_APPLY: there the caller reduces the key over all document shards first.
            if ((live || seg_apply) && !(flags & MVHDP_SWEEP_NO_APPLY) && mm.first_inactive >= 0 && e == hipSuccess) {
                long long key = LLONG_MAX;
                step(hipMemcpyAsync(&key, h->d_act_key, sizeof key, hipMemcpyDeviceToHost, s));
                step(hipStreamSynchronize(s));
                if (e == hipSuccess && key != LLONG_MAX) {
                    const int rc = apply_activation(h, MVHDP_ACT_KEY_TOPIC(key), MVHDP_ACT_KEY_VIEW(key));
                    if (rc != MVHDP_OK) { cleanup(); return rc; }
                    if (n_activations++ == 0) first_act = key;
                    mk.first_inactive = mm.first_inactive;
                    const long long none = LLONG_MAX;
                    step(hipMemcpyAsync(h->d_act_key, &none, sizeof none, hipMemcpyHostToDevice, s));
                    step(hipStreamSynchronize(s));                       // (`none` lives on this stack frame)
                }
            }
            if ((live && !(flags & MVHDP_SWEEP_REUSE_TREES)) || seg_apply) rebuild_trees();   // from the live / just-updated counts
            step(hipMemsetAsync(h->d_ovf_meta, 0, 2 * sizeof(unsigned int), s));         // overflow counts of passes 1, 2
            step(hipMemsetAsync(h->d_ovf_meta + ovf_word[2], 0, sizeof(unsigned int), s));
            step(hipMemsetAsync(class_counts, 0, MVHDP_N_CLASSES * sizeof(unsigned int), s));
        }
        step(hipMemsetAsync(h->d_doc_counter, 0, 8 * sizeof(unsigned long long), s));
        if (e != hipSuccess) break;
        if (classified) {
            const int64_t H_seg = share(H);
            ClassifyArgs ca{};
            ca.order = h->d_doc_order; ca.n = H_seg; ca.start = seg; ca.stride = nseg; ca.primary = pc; ca.counts = class_counts;
            for (int c = 0; c < MVHDP_N_CLASSES; c++) ca.lists[c] = h->d_lists + (size_t)c * mm.D;
            step(mvhdp_launch_classify(mm, ca, s));
            step(hipEventRecord(h->ev_fork, s));
            // Streams: the generic and the 16-round class share side stream 4, the 8-round class has side
            // stream 3 (the runtime multiplexes streams onto few hardware queues: more side streams only
            // serialise behind each other); narrower classes run on the sweep's own stream ahead of the primary.
            bool used[MVHDP_N_CLASSES] = {};
            for (int c = MVHDP_N_CLASSES - 1; c > pc && e == hipSuccess; c--) {       // widest first
                // class c holds lists of more than 32 << c topics: skipped when the corpus has none -- except that the
                // generic class also takes the entities with a view too long for the 16-bit counts of the wide variants
                if (S_cap <= (32 << c) && !(c == 5 && mdt > 65535)) continue;
                if (H_seg == 0) continue;
                const int si = (c >= 4) ? 4 : c;
                hipStream_t st = s;
                if (c >= 3) {
                    if (!h->side[si]) step(hipStreamCreateWithFlags(&h->side[si], hipStreamNonBlocking));
                    if (!h->ev_join[si]) step(hipEventCreateWithFlags(&h->ev_join[si], hipEventDisableTiming));
                    if (e != hipSuccess) break;
                    if (!used[si]) step(hipStreamWaitEvent(h->side[si], h->ev_fork, 0));
                    used[si] = true;
                    st = h->side[si];
                }
                SweepLaunch sc = sl;
                sc.q_list = ca.lists[c]; sc.q_list_count = class_counts + c;
                sc.q_order = nullptr; sc.q_order_start = 0; sc.q_order_count = 0;
                sc.doc_counter = h->d_doc_counter + c;
                sc.wave_bytes = cls[c].wave_bytes; sc.waves_per_block = cls[c].wpb;
                if (cls_fast[c]) {
                    sc.S_cap = std::min(S_cap, 64 << c);
                    step(mvhdp_launch_sweep_fast(mk, sc, 1 << c, blocks_for(H_seg, cls[c]), debug, st));
                } else {
                    sc.S_cap = S_cap;
                    step(mvhdp_launch_sweep(mk, sc, blocks_for(H_seg, cls[c]), debug, st));
                }
            }
            for (int si = 0; si < MVHDP_N_CLASSES; si++) if (used[si]) step(hipEventRecord(h->ev_join[si], h->side[si]));
            if (e == hipSuccess) {
                // the primary variant: the measured entities that fit it, then everything too short to overflow
                SweepLaunch sp = sl;
                sp.q_list = ca.lists[pc]; sp.q_list_count = class_counts + pc;
                sp.q_order = h->d_doc_order; sp.q_order_start = seg + H_seg * nseg; sp.q_order_stride = nseg; sp.q_order_count = n_seg - H_seg;
                sp.doc_counter = h->d_doc_counter + pc;
                sp.wave_bytes = fst[0].wave_bytes; sp.waves_per_block = fst[0].wpb;
                sp.S_cap = std::min(S_cap, 64 * rmax);
                step(mvhdp_launch_sweep_fast(mk, sp, rmax, blocks_for(n_seg, fst[0]), debug, s));
            }
            for (int si = 0; si < MVHDP_N_CLASSES; si++) if (used[si]) step(hipStreamWaitEvent(s, h->ev_join[si], 0));
        } else if (fast) {
            // Optimistic chain.  Each later pass reads its entity list and the list's length from device memory (written
            // by the pass before it, earlier in stream order): no host round trip between passes.  Its grid is sized by
            // what COULD overflow the pass before -- the entities with more tokens than that variant has slots, a static
            // prefix of the longest-first order -- and the pass is not launched at all when nothing can.
            const int32_t* list = nullptr;
            int64_t bound = n_seg;                                   // upper bound of the entities the next pass can receive
            for (int p = 0; p < n_chain && e == hipSuccess && bound > 0; p++) {
                SweepLaunch sp = sl;
                int32_t* out = (p & 1) ? h->d_overflow2 : h->d_overflow;
                if (p > 0) {
                    sp.q_list = list; sp.q_list_count = h->d_ovf_meta + ovf_word[p - 1];
                    sp.q_order = nullptr; sp.q_order_start = 0; sp.q_order_count = 0;
                    sp.doc_counter = h->d_doc_counter + p;
                    sp.slot_hist = nullptr;                 // counted in pass 1 already
                } else {
                    sp.q_order = h->d_doc_order; sp.q_order_start = seg; sp.q_order_stride = nseg; sp.q_order_count = n_seg;
                }
                sp.overflow_list = out; sp.overflow_count = h->d_ovf_meta + ovf_word[p];
                sp.wave_bytes = fst[p].wave_bytes; sp.waves_per_block = fst[p].wpb;
                sp.S_cap = std::min(S_cap, 64 * chain[p]);
                step(mvhdp_launch_sweep_fast(mk, sp, chain[p], blocks_for(bound, fst[p]), debug, s));
                list = out;
                // a wide (16-bit count) variant also hands on the entities with a view beyond 65535 tokens
                const int64_t too_long = (chain[p] >= 8 && mdt > 65535) ? share(longer_than(65535)) : 0;
                bound = std::max<int64_t>((S_cap > 64 * chain[p]) ? share(longer_than((int64_t)64 * chain[p])) : 0, too_long);
            }
            if (e == hipSuccess && bound > 0 && n_chain > 0) {
                // last pass: topic lists (or views) beyond the register variants, generic LDS kernel
                SweepLaunch so = sl;
                so.q_list = list; so.q_list_count = h->d_ovf_meta + ovf_word[n_chain - 1];
                so.q_order = nullptr; so.q_order_start = 0; so.q_order_count = 0;
                so.doc_counter = h->d_doc_counter + 4;
                so.slot_hist = nullptr;
                so.wave_bytes = gen.wave_bytes; so.waves_per_block = gen.wpb; so.S_cap = S_cap;
                step(mvhdp_launch_sweep(mk, so, blocks_for(bound, gen), debug, s));
            }
        } else {
            SweepLaunch sg = sl;
            sg.q_order_start = seg; sg.q_order_stride = nseg; sg.q_order_count = n_seg;
            sg.wave_bytes = gen.wave_bytes; sg.waves_per_block = gen.wpb;
            step(mvhdp_launch_sweep(mk, sg, blocks_for(n_seg, gen), debug, s));
        }
        // segmented deferred sweep: the updater catches up before the next segment (counts += delta, delta = 0)
        if (seg_apply) step(mvhdp_launch_apply_delta(mm, h->d_stats, s));
    }
    if (live) {
        if (flags & MVHDP_SWEEP_NO_APPLY) step(mvhdp_launch_live_helper(mm, 1, h->d_stats, s));   // delta = after - before, counts = snapshot
        else step(mvhdp_launch_live_helper(mm, 2, h->d_stats, s));                                  // UPD:202-215
    }
    step(hipEventRecord(h->ev[2], s));
    unsigned long long hs[ST_COUNT] = {0};
    long long act = LLONG_MAX;
    step(hipMemcpyAsync(hs, h->d_stats, sizeof hs, hipMemcpyDeviceToHost, s));
    step(hipMemcpyAsync(&act, h->d_act_key, sizeof act, hipMemcpyDeviceToHost, s));
    step(hipMemcpyAsync(ovf, h->d_ovf_meta, sizeof ovf, hipMemcpyDeviceToHost, s));
    step(hipStreamSynchronize(s));
    if (e == hipSuccess && sl.walk) {
        double all = 0.0;
        for (int m = 0; m < M; m++) {
            const double n = (double)hs[ST_VIEW_BASE + m * MVHDP_VIEW_STATS];
            all += n;
            if (n >= 64) h->walk_f[m] = (double)hs[ST_VIEW_BASE + m * MVHDP_VIEW_STATS + 1] / n;
            else if (nseg == 1) h->walk_f[m] = 1.0;                  // a view with next to no tokens is never steered
        }
        for (int b = 0; b < MVHDP_WALK_BINS; b++) {
            double c = 0.0;
            for (int m = 0; m < M; m++)
                if (h->walk_f[m] >= 0.0 && h->walk_f[m] < 0.35) c += (double)hs[ST_VIEW_BASE + m * MVHDP_VIEW_STATS + 2 + b];
            h->walk_hist[b] = all > 0 ? c / all : 0.0;
        }
    }
    if (e == hipSuccess && mm.D > 0) {
        // next sweep: the variant that is cheapest for this sweep's topic-list histogram (topic lists change slowly
        // between sweeps); entities counted twice (overflow re-run) only make the choice more conservative
        std::copy(ovf + 1, ovf + 1 + MVHDP_HIST_BINS, h->last_hist);
        h->rmax_hint = rmax_from_hist(ovf + 1);
        if (h->rmax_hint <= 2) {
            // 1 round or 2?  The 1-round variant is proposed as soon as half of the tokens sit in lists of at most 64 topics
            // (the others then run on their own class kernels: classified dispatch below) and kept only if the sweep's clock
            // agrees (the trial at the end of this function); a failed trial bans it for a while.
            double tot = 0, beyond = 0;
            for (int b = 0; b < MVHDP_HIST_BINS; b++) { tot += (double)ovf[1 + b]; if (b >= 1) beyond += (double)ovf[1 + b]; }
            h->rmax_hint = (beyond <= 0.5 * tot && h->sweeps_done >= h->one_round_banned_until) ? 1 : 2;
        }
    }
    if (e != hipSuccess) { cleanup(); HIPC(h, e); }
    if (getenv("MVHDP_DEBUG")) {
        fprintf(stderr, "[mvhdp] walk threshold %.2f (base %.2f, phase %d, dir %+d); tree branch %.4f of tokens, walked on demand %.4f; per view (threshold, tree share):",
                (double)h->walk_probe_i / MVHDP_WALK_BINS, (double)h->walk_i / MVHDP_WALK_BINS, h->walk_phase, h->walk_dir,
                (double)hs[ST_TREE] / std::max<double>(1.0, (double)hs[ST_TOKENS]), (double)hs[ST_ONDEMAND] / std::max<double>(1.0, (double)hs[ST_TOKENS]));
        for (int m = 0; m < M; m++) {
            const double n = std::max<double>(1.0, (double)hs[ST_VIEW_BASE + m * MVHDP_VIEW_STATS]);
            fprintf(stderr, " (%.2f %.3f)", sl.walk_theta[m], hs[ST_VIEW_BASE + m * MVHDP_VIEW_STATS + 1] / n);
        }
        fprintf(stderr, "\n");
    }
    if (getenv("MVHDP_DEBUG") && hs[ST_T_TOTAL])
        fprintf(stderr, "[mvhdp] wave cycles: queue %.1f%% prologue %.1f%% view setup %.1f%% chunk head %.1f%% tokens %.1f%% chunk end %.1f%% | %.0f cycles per token per wave\n",
                100.0 * hs[ST_T_QUEUE] / hs[ST_T_TOTAL], 100.0 * hs[ST_T_PROLOGUE] / hs[ST_T_TOTAL], 100.0 * hs[ST_T_VIEW] / hs[ST_T_TOTAL],
                100.0 * hs[ST_T_CHUNK_HEAD] / hs[ST_T_TOTAL], 100.0 * hs[ST_T_TOKENS] / hs[ST_T_TOTAL], 100.0 * hs[ST_T_CHUNK_END] / hs[ST_T_TOTAL],
                (double)hs[ST_T_TOTAL] / std::max<double>(1.0, (double)hs[ST_TOKENS]));
    if (hs[ST_MISCLASS]) { cleanup(); FAIL(h, MVHDP_ERR_HIP, "internal: an entity reached a sweep kernel variant that cannot hold its topic list"); }

    if (debug) {
        for (int m = 0; m < M; m++)
            if (sl.tok_dbg[m]) step(hipMemcpy(dbg->tok_dbg[m], sl.tok_dbg[m], (size_t)h->N[m] * 4 * sizeof(double), hipMemcpyDeviceToHost));
        if (sl.n_trace > 0) step(hipMemcpy(dbg->trace_out, sl.trace_out, (size_t)sl.n_trace * (K + 1) * sizeof(double), hipMemcpyDeviceToHost));
        cleanup();
        if (e != hipSuccess) HIPC(h, e);
    }

    mvhdp_sweep_stats st{};
    st.tokens = (int64_t)hs[ST_TOKENS]; st.changed = (int64_t)hs[ST_CHANGED];
    st.new_mass_cnt = (int64_t)hs[ST_NEW]; st.topic_doc_mass_cnt = (int64_t)hs[ST_DOC];
    st.word_ftree_mass_cnt = (int64_t)hs[ST_TREE]; st.oov_skipped = (int64_t)hs[ST_OOV];
    st.aborted_docs = (int64_t)hs[ST_ABORT]; st.exact_fallbacks = (int64_t)hs[ST_FALLBACK];
    st.activation_key = act;
    st.activated_topic = -1; st.activated_modality = -1;
    if (act != LLONG_MAX) { st.activated_topic = MVHDP_ACT_KEY_TOPIC(act); st.activated_modality = MVHDP_ACT_KEY_VIEW(act); }

    int ret = MVHDP_OK;
    if (flags & MVHDP_SWEEP_FROZEN) {
        // nothing was queued (WRK:587): the delta buffer is untouched
    } else if (flags & MVHDP_SWEEP_NO_APPLY) {
        h->delta_pending = true;
    } else if (live || seg_apply) {
        // the counts are already updated; what is left of the updater's work is the topic activation of the last segment
        h->have_trees = false;
        if (seg_apply) h->delta_clean = true;                        // apply_delta_kernel zeroed what it added
        ret = apply_activation(h, st.activated_topic, st.activated_modality);
        if (st.activated_topic >= 0) n_activations++;
        if (first_act != LLONG_MAX) {                                // report the sweep's first activation
            st.activation_key = first_act;
            st.activated_topic = MVHDP_ACT_KEY_TOPIC(first_act); st.activated_modality = MVHDP_ACT_KEY_VIEW(first_act);
        }
        if (ret == MVHDP_OK && hs[ST_NEGATIVE]) { h->err = "a topic count went below zero (UPD:202-215)"; ret = MVHDP_ERR_NEGATIVE_COUNT; }
    } else {
        if (st.activated_topic >= 0) n_activations = 1;
        ret = mvhdp_apply_delta(h, st.activated_topic, st.activated_modality);
    }
    HIPC(h, hipEventRecord(h->ev[3], s));
    HIPC(h, hipEventSynchronize(h->ev[3]));
    float ms_k = 0, ms_t = 0;
    hipEventElapsedTime(&ms_k, h->ev[1], h->ev[2]);
    hipEventElapsedTime(&ms_t, h->ev[0], h->ev[3]);
    st.sweep_kernel_ms = ms_k; st.total_ms = ms_t;
    st.activations = n_activations; st.reserved = 0;
    // the 1-round trial (see mvhdp_ctx::last_primary): plain full sweeps only, so that the two times are comparable
    h->sweeps_done++;
    const bool plain = fast && nseg == 1 && !debug && !(flags & (MVHDP_SWEEP_FROZEN | MVHDP_SWEEP_EXACT_CHAIN)) &&
                       !getenv("MVHDP_FORCE_RMAX") && !getenv("MVHDP_FORCE_MODE") && st.tokens > 0;
    if (plain) {
        const double ns = (double)ms_k * 1e6 / (double)st.tokens;    // (a classify pass is inside ms_k)
        if (rmax == 1 && h->last_primary == 2 && h->last_ns_per_token > 0) h->two_round_ns_per_token = h->last_ns_per_token;
        if (rmax == 1 && h->two_round_ns_per_token > 0 && ns > 0.995 * h->two_round_ns_per_token) {
            const int high = MVHDP_WALK_BINS * 4 / 5;
            if (!h->one_round_retry && !theta_env && walk_any && h->walk_cls == 0 && h->walk_i < high) {
                // The 1-round variant is the bandwidth-bound one: it gets its edge from a HIGH walk threshold, and the search
                // has not taken it there yet (it inherits nothing from the wider variants).  One more sweep at 0.8 decides.
                h->one_round_retry = true;
                h->walk_i = high; h->walk_phase = 0; h->walk_idle_until = h->sweeps_done + 1;
            } else {
                h->one_round_retry = false;
                if (h->rmax_hint == 1) h->rmax_hint = 2;
                h->one_round_banned_until = h->sweeps_done + h->one_round_ban;
                h->one_round_ban = std::min<long long>(32, h->one_round_ban * 2);
                h->two_round_ns_per_token = 0;
            }
        } else if (rmax == 1) {                                      // the trial is over: the 1-round variant stays
            if (h->two_round_ns_per_token > 0) h->one_round_ban = 4;
            h->two_round_ns_per_token = 0;
            h->one_round_retry = false;
        }
        h->last_primary = rmax; h->last_ns_per_token = ns;
    } else h->last_primary = 0;
    // the walk-threshold search (see mvhdp_ctx::walk_i): among sweeps of one kernel configuration and update mode only
    const bool comparable = fast && !debug && !theta_env && !(flags & (MVHDP_SWEEP_FROZEN | MVHDP_SWEEP_EXACT_CHAIN)) && st.tokens > 0;
    const int walk_cfg = rmax * 2 + (classified ? 1 : 0) + 64 * nseg + (live ? 1 << 16 : 0) + (seg_apply ? 1 << 17 : 0);
    if (!comparable || h->walk_cfg != walk_cfg) {
        h->walk_phase = 0;
        h->walk_cfg = comparable ? walk_cfg : -1;
        if (comparable) h->walk_ns_a1 = 0.0;
    }
    if (comparable && !walk_any) h->walk_phase = 0;
    if (comparable && walk_any) {
        const double ns = (double)ms_k * 1e6 / (double)st.tokens;
        if (h->walk_phase == 0) {
            h->walk_ns_a1 = ns;
            if (h->sweeps_done >= h->walk_idle_until) h->walk_phase = 1;
        } else if (h->walk_phase == 1) {
            h->walk_ns_b = ns; h->walk_b_i = h->walk_probe_i; h->walk_phase = 2;
        } else {
            const double base = 0.5 * (h->walk_ns_a1 + ns);
            const int step = h->walk_b_i - h->walk_i;
            const bool far = h->walk_far;
            h->walk_far = false;
            // leaving threshold 0 also changes the kernel flavour: ask for more there (no flapping between the two)
            const double need = h->walk_i == 0 ? 0.005 : 0.0025;
            if (std::fabs(h->walk_ns_a1 - ns) > 0.025 * base) {          // the A sweeps disagree (a variant change, a jump of the chain): no verdict
                h->walk_ns_a1 = ns; h->walk_phase = 1; h->walk_far = far;
            } else if (step != 0 && h->walk_ns_b < base * (1.0 - need)) {
                h->walk_i = h->walk_b_i;
                h->walk_ns_a1 = h->walk_ns_b;                            // the B sweep is the first A sweep of the next step
                h->walk_fails = 0; h->walk_wait = 4; h->walk_phase = 1;
                if (step > 0) h->walk_maxj = std::min(6, h->walk_maxj * 2);
                if (step > 0 && h->walk_ns_b < base * (1.0 - 0.008)) h->walk_cap = std::min(0.05, h->walk_cap * 2.0);
                if (far) h->walk_far = true;                             // half again
            } else if (step > 1) {                                       // a long step that did not pay: a shorter one, same direction
                h->walk_ns_a1 = ns;
                h->walk_maxj = std::max(1, step / 2);
                h->walk_cap = std::max(0.004, h->walk_cap * 0.5);
                h->walk_phase = 1;
            } else if (far) {                                            // half the threshold is no better: back to single steps
                h->walk_ns_a1 = ns;
                h->walk_dir = 1;
                h->walk_phase = 1;
            } else {
                h->walk_ns_a1 = ns;
                h->walk_dir = -h->walk_dir;
                h->walk_phase = 1;
                if (++h->walk_fails >= 2) {
                    h->walk_fails = 0;
                    h->walk_idle_until = h->sweeps_done + h->walk_wait;
                    h->walk_wait = std::min(64, h->walk_wait * 2);
                    h->walk_phase = 0;
                    h->walk_far = true;
                }
            }
        }
    }
    if (stats) *stats = st;
    return ret;
}

extern "C" int mvhdp_get_view_weights(mvhdp_handle h, double* p)
{
    CHECK_H(h);
    MvModel& mm = h->mm;
    if (!p) FAIL(h, MVHDP_ERR_INVALID_ARG, "get_view_weights: null");
    if (mm.M == 1) { for (int64_t d = 0; d < mm.D; d++) p[d] = 1.0; return MVHDP_OK; }
    if (!mm.p) FAIL(h, MVHDP_ERR_STATE, "get_view_weights before the first sweep");
    HIPC(h, hipSetDevice(h->device));
    HIPC(h, hipStreamSynchronize(h->stream));
    HIPC(h, hipMemcpy(p, mm.p, (size_t)mm.D * mm.M * mm.M * sizeof(double), hipMemcpyDeviceToHost));
    return MVHDP_OK;
}

extern "C" int mvhdp_device_buffer(mvhdp_handle h, mvhdp_buffer which, void** dev_ptr, size_t* bytes)
{
    CHECK_H(h);
    if (!dev_ptr || !bytes) FAIL(h, MVHDP_ERR_INVALID_ARG, "device_buffer: null");
    size_t b = (size_t)counts_len(h) * sizeof(int32_t);
    if (which == MVHDP_BUF_COUNTS) { *dev_ptr = h->mm.counts; *bytes = b; return MVHDP_OK; }
    if (which == MVHDP_BUF_DELTA) { *dev_ptr = h->mm.delta; *bytes = b; return MVHDP_OK; }
    FAIL(h, MVHDP_ERR_INVALID_ARG, "device_buffer: unknown buffer");
}

extern "C" int mvhdp_counts_written(mvhdp_handle h)
{
    CHECK_H(h);
    h->have_counts = true; h->have_trees = false;
    return MVHDP_OK;
}

// ---------------------------------------------------------------------------
// statistics for optimizeBeta / optimizeP and the log likelihood
// ---------------------------------------------------------------------------
// MALLET 2.0.8 Dirichlet.logGammaStirling (restated from the class file's bytecode)
static double log_gamma_stirling_host(double z)
{
    const double HALF_LOG_TWO_PI = std::log(6.283185307179586) / 2.0;
    int shift = 0;
    while (z < 2.0) { z = z + 1; shift++; }
    double result = HALF_LOG_TWO_PI + (z - 0.5) * std::log(z) - z + 1 / (12.0 * z) - 1 / (360.0 * z * z * z)
                    + 1 / (1260.0 * z * z * z * z * z);
    while (shift > 0) { shift--; z = z - 1; result = result - std::log(z); }
    return result;
}

extern "C" int mvhdp_get_count_histogram(mvhdp_handle h, int32_t m, int32_t* hist, int32_t len)
{
    CHECK_H(h);
    MvModel& mm = h->mm;
    if (m < 0 || m >= mm.M || !hist || len < 1) FAIL(h, MVHDP_ERR_INVALID_ARG, "get_count_histogram: bad argument");
    if (!h->have_counts) FAIL(h, MVHDP_ERR_STATE, "get_count_histogram before build_counts/set_counts");
    HIPC(h, hipSetDevice(h->device));
    int32_t* d = nullptr;
    HIPC(h, hipMalloc(&d, (size_t)len * sizeof(int32_t)));
    hipError_t e = mvhdp_launch_count_hist(mm, m, d, len, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hist, d, (size_t)len * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d);
    HIPC(h, e);
    return MVHDP_OK;
}

extern "C" int mvhdp_view_overlap_sums(mvhdp_handle h, double* sums)
{
    CHECK_H(h);
    MvModel& mm = h->mm;
    if (!sums) FAIL(h, MVHDP_ERR_INVALID_ARG, "view_overlap_sums: null");
    int rc = require_corpus(h); if (rc) return rc;
    const int M = mm.M;
    for (int i = 0; i < M * M; i++) sums[i] = 0.0;
    if (mm.D == 0) return MVHDP_OK;
    HIPC(h, hipSetDevice(h->device));
    double* d = nullptr;
    const size_t n = (size_t)M * M * mm.D;
    HIPC(h, hipMalloc(&d, n * sizeof(double)));
    std::vector<double> host(n);
    hipError_t e = mvhdp_launch_view_overlap(mm, d, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(host.data(), d, n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d);
    HIPC(h, e);
    for (int i = 0; i < M * M; i++) {                     // PTM:2789-2792: sequential, entity order
        double acc = 0;
        const double* col = host.data() + (size_t)i * mm.D;
        for (int64_t doc = 0; doc < mm.D; doc++) acc += col[doc];
        sums[i] = acc;
    }
    return MVHDP_OK;
}

extern "C" int mvhdp_doc_topic_proportions(mvhdp_handle h, const double* view_weights, int64_t d0, int64_t d1, double* out)
{
    CHECK_H(h);
    MvModel& mm = h->mm;
    int rc = require_corpus(h); if (rc) return rc;
    if (!h->have_hyper) FAIL(h, MVHDP_ERR_STATE, "doc_topic_proportions before set_hyper");
    if (!view_weights || !out || d0 < 0 || d1 > mm.D || d0 > d1) FAIL(h, MVHDP_ERR_INVALID_ARG, "doc_topic_proportions: bad range or null buffer");
    if (d1 == d0) return MVHDP_OK;
    HIPC(h, hipSetDevice(h->device));
    DocTopicCarry carry{};
    for (int m = 0; m < mm.M; m++) {
        if (!h->d_carry[m]) {                              // PTM:2873-2886: a missing view keeps the previous entity's counts
            std::vector<int64_t> src((size_t)mm.D);
            int64_t last = -1;
            for (int64_t d = 0; d < mm.D; d++) {
                if (h->h_doc_off[m][d + 1] > h->h_doc_off[m][d]) last = d;
                src[(size_t)d] = last;
            }
            HIPC(h, hipMalloc(&h->d_carry[m], (size_t)mm.D * sizeof(int64_t)));
            HIPC(h, hipMemcpy(h->d_carry[m], src.data(), (size_t)mm.D * sizeof(int64_t), hipMemcpyHostToDevice));
        }
        carry.src[m] = h->d_carry[m];
    }
    double *d_w = nullptr, *d_out = nullptr;
    const size_t n = (size_t)(d1 - d0) * mm.K;
    hipError_t e = hipMalloc(&d_w, (size_t)mm.M * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&d_out, n * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(d_w, view_weights, (size_t)mm.M * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = mvhdp_launch_doc_topic_prop(mm, carry, d_w, d0, d1, d_out, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (d_w) hipFree(d_w);
    if (d_out) hipFree(d_out);
    HIPC(h, e);
    return MVHDP_OK;
}

extern "C" int mvhdp_gamma_doc_statistics(mvhdp_handle h, int32_t m, double gamma_m, uint64_t seed, uint32_t round, double* qs, double* qw)
{
    CHECK_H(h);
    MvModel& mm = h->mm;
    if (m < 0 || m >= mm.M || !qs || !qw || !(gamma_m > 0.0)) FAIL(h, MVHDP_ERR_INVALID_ARG, "gamma_doc_statistics: bad argument");
    int rc = require_corpus(h); if (rc) return rc;
    *qs = 0; *qw = 0;
    if (mm.D == 0) return MVHDP_OK;
    HIPC(h, hipSetDevice(h->device));
    const int NB = 1024;                                   // fixed: the summation order is part of the result
    double* d_part = nullptr;
    std::vector<double> part(2 * NB);
    hipError_t e = hipMalloc(&d_part, 2 * NB * sizeof(double));
    if (e == hipSuccess) e = mvhdp_launch_gamma_doc_stats(mm, m, gamma_m, (uint32_t)seed, (uint32_t)(seed >> 32), round, d_part, NB, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(part.data(), d_part, 2 * NB * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (d_part) hipFree(d_part);
    HIPC(h, e);
    double a = 0, b = 0;
    for (int i = 0; i < NB; i++) { a += part[2 * i]; b += part[2 * i + 1]; }
    *qs = a; *qw = b;
    return MVHDP_OK;
}

extern "C" int mvhdp_model_log_likelihood(mvhdp_handle h, double* out)
{
    CHECK_H(h);
    MvModel& mm = h->mm;
    if (!out) FAIL(h, MVHDP_ERR_INVALID_ARG, "model_log_likelihood: null");
    int rc = require_corpus(h); if (rc) return rc;
    if (!h->have_hyper || !h->have_counts) FAIL(h, MVHDP_ERR_STATE, "model_log_likelihood before set_hyper/build_counts");
    const int M = mm.M, K = mm.K;
    HIPC(h, hipSetDevice(h->device));
    const int NP = 1024;
    double *d_doc = nullptr, *d_part = nullptr;
    unsigned long long* d_nz = nullptr;
    hipError_t e = hipMalloc(&d_doc, (size_t)std::max<int64_t>(mm.D, 1) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&d_part, NP * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&d_nz, sizeof(unsigned long long));
    std::vector<double> hdoc((size_t)std::max<int64_t>(mm.D, 1)), hpart(NP);
    std::vector<int32_t> nk((size_t)K);
    for (int m = 0; m < M && e == hipSuccess; m++) {
        unsigned long long nz = 0;
        e = mvhdp_launch_loglik(mm, m, d_doc, d_part, NP, d_nz, h->stream);
        if (e == hipSuccess && mm.D > 0) e = hipMemcpyAsync(hdoc.data(), d_doc, (size_t)mm.D * sizeof(double), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(hpart.data(), d_part, NP * sizeof(double), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&nz, d_nz, sizeof nz, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(nk.data(), mm.counts + mm.rowbase[M] * K + (int64_t)m * K, (size_t)K * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) break;
        double ll = 0;
        int64_t modalityCnt = 0;
        for (int64_t d = 0; d < mm.D; d++) {
            if (h->h_doc_off[m][d + 1] > h->h_doc_off[m][d]) { ll += hdoc[d]; modalityCnt++; }     // PTM:3348-3367
        }
        ll += modalityCnt * log_gamma_stirling_host((double)mm.gamma[m] * mm.alpha_sum[m]);       // PTM:3373
        if (std::isnan(ll) || std::isinf(ll)) { out[m] = 0; continue; }                           // PTM:3375-3383
        for (int i = 0; i < NP; i++) ll += hpart[i];                                                // PTM:3389-3415
        if (std::isnan(ll) || std::isinf(ll)) ll = 0;
        const double bv = mm.beta[m] * mm.V[m];
        for (int topic = 0; topic < K; topic++) {                                                   // PTM:3417-3435
            ll -= (bv + nk[topic]) == 0 ? 0 : log_gamma_stirling_host(bv + nk[topic]);
            if (std::isnan(ll) || std::isinf(ll)) ll = 0;
        }
        ll += bv == 0 ? 0 : log_gamma_stirling_host(bv) * K;                                        // PTM:3438
        ll -= mm.beta[m] == 0 ? 0 : log_gamma_stirling_host(mm.beta[m]) * (double)nz;               // PTM:3441
        if (std::isinf(ll)) ll = 0;
        out[m] = ll;
    }
    if (d_doc) hipFree(d_doc);
    if (d_part) hipFree(d_part);
    if (d_nz) hipFree(d_nz);
    HIPC(h, e);
    return MVHDP_OK;
}
