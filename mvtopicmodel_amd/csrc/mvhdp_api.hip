// mvhdp_api.hip — host side of the C ABI declared in include/mvhdp.h.
// Owns the device state of one model shard (one HIP device), launches the
// kernels of mvhdp_kernels.hip on one stream and copies results back.
// There is NO CPU fallback: without a gfx950 device mvhdp_create fails.
#include "mvhdp_ctx.h"

static thread_local std::string g_create_error;

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

bool mvhdp_is_live(mvhdp_ctx* h)
{
    std::lock_guard<std::mutex> lk(g_reg_mutex);
    return g_live && g_live->count(h) != 0;
}

// The environment is read ONCE, here (diagnostics; a host uses mvhdp_set_tuning):
//   MVHDP_FORCE_RMAX=1|2|4|8|16   primary variant          MVHDP_NARROW=0        never the 16-bit mirror
//   MVHDP_WALK_THETA=t0,t1,...    fixed walk thresholds    MVHDP_SINGLE_STREAM=1 class kernels one after another
//   MVHDP_DEBUG=1                 the plan of every sweep on stderr
static void read_environment(mvhdp_ctx* h)
{
    if (const char* f = getenv("MVHDP_FORCE_RMAX")) { const int v = atoi(f); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32) h->tu.force_primary = v; }
    if (const char* f = getenv("MVHDP_NARROW")) h->tu.narrow = atoi(f) != 0 ? -1 : 0;
    if (const char* f = getenv("MVHDP_SINGLE_STREAM")) h->tu.single_stream = atoi(f) != 0;
    if (const char* f = getenv("MVHDP_SIDE_PRIORITY")) h->side_priority = atoi(f);               // 0: none, 1: A, B and D, 2 (default): A and B      // 0: every side stream at normal priority (diagnostics)
    if (const char* f = getenv("MVHDP_LIVE16")) h->tu.live16 = atoi(f);
    if (const char* f = getenv("MVHDP_LIVE_OVERLAP")) h->tu.live_overlap = atoi(f);
    if (const char* f = getenv("MVHDP_LIVE_ROWS")) h->tu.live_rows = atoi(f);                     // 0: stored trees rebuilt at every segment border (the round-4 form of a live sweep)
    if (const char* f = getenv("MVHDP_LIVE_ROWS_THETA")) h->tu.live_rows_theta = atof(f);
    if (const char* f = getenv("MVHDP_COEF_LDS_KB")) h->tu.coef_lds_max_bytes = std::max(0, std::min(64, atoi(f))) * 1024;   // (experiment: the live-rows coefficient table in LDS up to this size)
    if (const char* f = getenv("MVHDP_LIVE_ROWS_SEGMENTS")) h->tu.live_rows_segments = std::max(1, std::min(255, atoi(f)));
    if (const char* f = getenv("MVHDP_WIDEST_ON_MAIN")) h->tu.widest_on_main = atoi(f) != 0;
    if (const char* f = getenv("MVHDP_NARROW_WIDE")) h->tu.narrow_wide = atoi(f) != 0;
    if (const char* f = getenv("MVHDP_LIVE_TREE_EVERY")) h->live_tree_every = std::max(1, atoi(f));   // (diagnostics: a live sweep rebuilds its trees at every n-th segment border only)
    if (const char* f = getenv("MVHDP_GATE_PCT")) { const int v = atoi(f); if (v >= 5 && v <= 95) h->gate_pct = v; }   // how far through a live segment the next one is prepared
    if (const char* f = getenv("MVHDP_FOUR_ROUND_ON_C")) h->tu.four_round_on_c = atoi(f);         // -1 by its token share (default), 0 / 1
    if (const char* f = getenv("MVHDP_DELTA16")) h->tu.delta16 = atoi(f) != 0;                   // 0: every n_wk delta in the 32-bit table (diagnostics)
    if (const char* f = getenv("MVHDP_FORK_DELAY_US")) h->tu.fork_delay_us = std::max(0, std::min(1000, atoi(f)));
    if (const char* f = getenv("MVHDP_FORCE_MODE")) { if (!strcmp(f, "serial")) h->tu.single_stream = 1; }   // "streams" (default): class kernels side by side
    if (const char* f = getenv("MVHDP_PRIMARY_MIN_SHARE")) { const double v = atof(f); if (v > 0.0 && v <= 1.0) h->tu.primary_min_share = v; }
    if (const char* t = getenv("MVHDP_WALK_THETA")) {
        int m = 0;
        for (const char* q = t; *q && m < MVHDP_MAXM; m++) {
            h->tu.walk_theta[m] = atof(q);
            while (*q && *q != ',') q++;
            if (*q == ',') q++;
        }
        for (; m < MVHDP_MAXM; m++) h->tu.walk_theta[m] = 0.0;
        h->tu.walk_fixed = 1;
    }
    h->dbg_env = getenv("MVHDP_DEBUG") != nullptr;
}

extern "C" const char* mvhdp_version(void) { return "mvhdp 0.1 (gfx950)"; }

extern "C" const char* mvhdp_last_error(mvhdp_handle h) { return (h && mvhdp_is_live(h)) ? h->err.c_str() : g_create_error.c_str(); }

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
    for (int m = 0; m < M; m++) if (cfg->num_types[m] >= (1 << 29)) { g_create_error = "num_types[m] must be below 2^29"; return MVHDP_ERR_INVALID_ARG; }
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
    // (+ MVHDP_TAIL_WORDS behind the tokensPerTopic part of both buffers: the status word a group of document shards reduces together
    // with that part, so that every rank learns of a failure on any rank inside the collective it has to enter anyway)
    const size_t cbytes = (size_t)(nrows * K + (int64_t)M * K + MVHDP_TAIL_WORDS) * sizeof(int32_t);
    CREATE_HIP(hipMalloc(&mm.counts, cbytes));
    CREATE_HIP(hipMalloc(&mm.delta, cbytes));
    CREATE_HIP(hipMemset(mm.counts, 0, cbytes));
    CREATE_HIP(hipMalloc(&mm.counts16, (size_t)nrows * K * sizeof(uint16_t) + 64));   // (+ 64: the 16-byte row loads of the live-rows form may end beyond the last row)
    CREATE_HIP(hipMemset(mm.counts16, 0, (size_t)nrows * K * sizeof(uint16_t)));
    CREATE_HIP(hipMalloc(&mm.delta16, (size_t)(nrows * K + 2) * sizeof(uint16_t)));
    CREATE_HIP(hipMemsetD16(mm.delta16, (unsigned short)0x8000, (size_t)(nrows * K + 2)));       // (the bias: see SweepLaunch::delta16)
    CREATE_HIP(hipMalloc(&mm.heavy, (size_t)nrows));
    CREATE_HIP(hipMemset(mm.heavy, 1, (size_t)nrows));
    CREATE_HIP(hipMemset(mm.delta, 0, cbytes));
    CREATE_HIP(hipMalloc(&mm.trees, (size_t)nrows * 2 * K * sizeof(double)));
    CREATE_HIP(hipMalloc(&mm.root, (size_t)nrows * sizeof(double)));
    CREATE_HIP(hipMalloc(&mm.mass0, (size_t)nrows * sizeof(float)));
    CREATE_HIP(hipMemset(mm.mass0, 0, (size_t)nrows * sizeof(float)));
    CREATE_HIP(hipMalloc(&mm.coef, (size_t)M * (((K + 7) & ~7) + K + 8) * sizeof(float)));   // coef [M][Kp] (zero-padded rows), then the smoothing running sums [M][K]
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
    CREATE_HIP(hipMalloc(&h->d_births, (size_t)(2 + 2 * K) * sizeof(int32_t)));
    CREATE_HIP(hipMemset(h->d_births, 0, (size_t)(2 + 2 * K) * sizeof(int32_t)));
    CREATE_HIP(hipMalloc(&h->d_birth_keys, (size_t)K * sizeof(long long)));
    CREATE_HIP(hipMalloc(&h->d_ctl, CTL_WORDS * sizeof(unsigned long long)));
    CREATE_HIP(hipMemset(h->d_ctl, 0, CTL_WORDS * sizeof(unsigned long long)));
    h->d_stats = h->d_ctl;
    h->d_act_key = (long long*)(h->d_ctl + ST_COUNT);
    h->d_ovf_meta = (unsigned int*)(h->d_ctl + ST_COUNT + 1);
    h->d_doc_counter = h->d_ctl + ST_COUNT + 1 + META_WORDS64;                       // one work-queue head per kernel class
    CREATE_HIP(hipHostMalloc((void**)&h->h_ctl, CTL_WORDS * sizeof(unsigned long long), hipHostMallocDefault));   // pinned: the read-back never blocks the host
    for (int c = 0; c < MVHDP_N_CLASSES; c++)
        for (int f = 0; f < 3; f++) h->regs.regs[c][f] = mvhdp_sweep_kernel_regs(c, f);
    read_environment(h);
    mm.alpha = h->d_alpha;
    mm.inactive = h->d_inactive;
    h->h_alpha.assign((size_t)M * (K + 1), 0.0);
    h->h_inactive.assign((size_t)K, 0);
    h->wt.init_defaults(K);
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
    for (int m = 0; m < MVHDP_MAXM; m++) { fr(h->d_doc_off[m]); fr(h->d_tok[m]); fr(h->d_z[m]); fr(h->d_carry[m]); fr(h->d_present[m]); }
    fr(h->mm.counts); fr(h->mm.delta16); fr(h->mm.counts16); fr(h->mm.heavy); fr(h->mm.delta); fr(h->mm.trees); fr(h->mm.root); fr(h->mm.coef); fr(h->mm.mass0); fr(h->d_births); fr(h->d_birth_keys); fr(h->mm.dtab); fr(h->mm.p);
    fr(h->d_alpha); fr(h->d_inactive); fr(h->d_ctl);
    if (h->h_ctl) { hipHostFree(h->h_ctl); h->h_ctl = nullptr; }
    h->d_stats = nullptr; h->d_act_key = nullptr; h->d_doc_counter = nullptr; h->d_ovf_meta = nullptr;
    fr(h->d_doc_order); fr(h->d_lists); fr(h->d_nslots); fr(h->d_stats_many); fr(h->d_heavy_list); fr(h->d_heavy_ctl);
    if (h->ev_rf_go) { hipEventDestroy(h->ev_rf_go); h->ev_rf_go = nullptr; }
    if (h->ev_rf_done) { hipEventDestroy(h->ev_rf_done); h->ev_rf_done = nullptr; }
    if (h->rf_stream) { hipStreamDestroy(h->rf_stream); h->rf_stream = nullptr; }
    fr(h->ov.counts2); fr(h->ov.counts16_2); fr(h->ov.dtab2); fr(h->ov.root2); fr(h->ov.trees2); fr(h->ov.delta2); fr(h->ov.delta3); fr(h->ov.ctl2); fr(h->ov.lists2);
    for (auto& e : h->ov.ev_seg) if (e) { hipEventDestroy(e); e = nullptr; }
    if (h->ov.ev_start) { hipEventDestroy(h->ov.ev_start); h->ov.ev_start = nullptr; }
    if (h->ov.x1) { hipStreamDestroy(h->ov.x1); h->ov.x1 = nullptr; }
    if (h->ov.xa) { hipStreamDestroy(h->ov.xa); h->ov.xa = nullptr; }
    for (auto& e : h->ev_many) if (e) { hipEventDestroy(e); e = nullptr; }
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
    if (h->d_present[m]) { hipFree(h->d_present[m]); h->d_present[m] = nullptr; }
    h->h_present[m].clear(); mm.present[m] = nullptr;
    HIPC(h, hipMalloc(&h->d_doc_off[m], (size_t)(D + 1) * sizeof(int64_t)));
    HIPC(h, hipMemcpy(h->d_doc_off[m], doc_off, (size_t)(D + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    const size_t nb = (size_t)std::max<int64_t>(N, 1) * sizeof(int32_t);
    HIPC(h, hipMalloc(&h->d_tok[m], nb));
    HIPC(h, hipMalloc(&h->d_z[m], nb));
    if (N > 0) HIPC(h, hipMemcpy(h->d_tok[m], tokens, (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPC(h, hipMemset(h->d_z[m], 0xff, nb));               // UNASSIGNED_TOPIC (-1), PTM:63
    h->unassigned[m] = N > 0;
    h->h_doc_off[m].assign(doc_off, doc_off + D + 1);
    h->N[m] = N;
    h->have_corpus[m] = true;
    h->max_doc_tokens = -1;
    if (h->d_doc_order) { hipFree(h->d_doc_order); h->d_doc_order = nullptr; }
    if (h->d_lists) { hipFree(h->d_lists); h->d_lists = nullptr; }
    if (h->ov.lists2) { hipFree(h->ov.lists2); h->ov.lists2 = nullptr; }
    if (h->d_nslots) { hipFree(h->d_nslots); h->d_nslots = nullptr; }
    mm.nslots = nullptr;
    for (auto& c : h->d_carry) if (c) { hipFree(c); c = nullptr; }
    h->nslots_valid = false;
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
    bool any_unassigned = false;
    for (int64_t i = 0; i < h->N[m]; i++) {
        if (z[i] < -1 || z[i] >= h->mm.K) FAIL(h, MVHDP_ERR_INVALID_ARG, "set_assignments: topic out of range");
        any_unassigned = any_unassigned || z[i] < 0;
    }
    HIPC(h, hipSetDevice(h->device));
    HIPC(h, hipStreamSynchronize(h->stream));
    if (h->N[m] > 0) HIPC(h, hipMemcpy(h->d_z[m], z, (size_t)h->N[m] * sizeof(int32_t), hipMemcpyHostToDevice));
    h->unassigned[m] = any_unassigned;
    h->nslots_valid = false;
    // the counts no longer describe these assignments: a sampling sweep is refused until build_counts / set_counts /
    // counts_written says they do again (a frozen sweep, whose counts are a trained model's by design, is not)
    if (h->have_counts) h->counts_stale = true;
    return MVHDP_OK;
}

extern "C" int mvhdp_set_view_presence(mvhdp_handle h, int32_t m, const uint8_t* present)
{
    CHECK_H(h);
    MvModel& mm = h->mm;
    if (m < 0 || m >= mm.M) FAIL(h, MVHDP_ERR_INVALID_ARG, "set_view_presence: bad view");
    if (!h->have_corpus[m]) FAIL(h, MVHDP_ERR_STATE, "set_view_presence before set_corpus");
    HIPC(h, hipSetDevice(h->device));
    HIPC(h, hipStreamSynchronize(h->stream));
    if (h->d_present[m]) { hipFree(h->d_present[m]); h->d_present[m] = nullptr; }
    h->h_present[m].clear(); mm.present[m] = nullptr;
    if (h->d_carry[m]) { hipFree(h->d_carry[m]); h->d_carry[m] = nullptr; }         // the carry-over map of doc_topic_proportions depends on it
    if (!present || mm.D == 0) return MVHDP_OK;
    for (int64_t d = 0; d < mm.D; d++)
        if (!present[d] && h->h_doc_off[m][d + 1] > h->h_doc_off[m][d]) FAIL(h, MVHDP_ERR_INVALID_ARG, "set_view_presence: an entity with tokens in the view is marked absent");
    h->h_present[m].assign(present, present + mm.D);
    HIPC(h, hipMalloc(&h->d_present[m], (size_t)mm.D));
    HIPC(h, hipMemcpy(h->d_present[m], present, (size_t)mm.D, hipMemcpyHostToDevice));
    mm.present[m] = h->d_present[m];
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
    if (h->delta16_used) {
        // 16-bit delta cells of a sweep that never reached its apply pass (it failed: the recount is how a host recovers): back to the bias
        HIPC(h, hipMemsetD16Async(h->mm.delta16, (unsigned short)0x8000, (size_t)(h->mm.rowbase[h->mm.M] * h->mm.K), h->stream));
        h->delta16_used = false;
    }
    HIPC(h, hipStreamSynchronize(h->stream));
    h->have_counts = true; h->have_trees = false; h->counts_stale = false;
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
    for (int m = 0; m < h->mm.M; m++) h->unassigned[m] = false;                                              // (every token got a topic, INF:169-199)
    h->nslots_valid = false;
    if (h->have_counts) h->counts_stale = true;
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
    h->have_counts = true; h->have_trees = false; h->counts_stale = false;
    return MVHDP_OK;
}

extern "C" int mvhdp_get_tree(mvhdp_handle h, int32_t m, int32_t type, double* tree)
{
    CHECK_H(h);
    MvModel& mm = h->mm;
    if (m < 0 || m >= mm.M || type < 0 || type >= mm.V[m] || !tree) FAIL(h, MVHDP_ERR_INVALID_ARG, "get_tree: bad argument");
    const bool raw = getenv("MVHDP_REFRESH_FULL") != nullptr;        // (diagnostics: whatever FTree.tree holds, e.g. what heavy_refresh_kernel left there)
    if (!h->have_trees && !raw) FAIL(h, MVHDP_ERR_STATE, "get_tree before build_trees");
    HIPC(h, hipSetDevice(h->device));
    if (!raw) { int rc2 = ensure_full_trees(h); if (rc2) return rc2; }
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

// Births of a live sweep, chunk by chunk (SweepLaunch::births): the reference's updater takes a topic out of inActiveTopicIndex with the
// FIRST delta that reaches it (UPD:263-270) and its samplers then draw the next inactive index (WRK:523-526) -- all 100 inactive topics of
// C5 are active within its first sweep.  A segment starts with the list of the topics that are inactive now (births_begin); its kernels
// move along that list as their chunks' deltas land; births_end activates what was reached, in index order, each topic's alpha[m][K]
// going to the view of its first delta.
static hipError_t births_begin(mvhdp_ctx* h, hipStream_t s)
{
    const int K = h->mm.K;
    h->h_births.assign((size_t)2 + 2 * K, -1);
    int n = 0;
    for (int k = 0; k < K; k++) if (h->h_inactive[k]) { h->h_births[(size_t)2 + K + k] = n; h->h_births[(size_t)2 + n++] = k; }
    h->h_births[0] = 0; h->h_births[1] = n;
    h->h_birth_keys.assign((size_t)K, LLONG_MAX);
    hipError_t e = hipMemcpyAsync(h->d_births, h->h_births.data(), h->h_births.size() * sizeof(int32_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(h->d_birth_keys, h->h_birth_keys.data(), (size_t)K * sizeof(long long), hipMemcpyHostToDevice, s);
    return e;
}
static int births_end(mvhdp_ctx* h, hipStream_t s, SweepOutcome& oc)
{
    MvModel& mm = h->mm;
    const int K = mm.K;
    int32_t head[2] = {0, 0};
    HIPC(h, hipMemcpyAsync(head, h->d_births, sizeof head, hipMemcpyDeviceToHost, s));
    HIPC(h, hipStreamSynchronize(s));
    const int n = std::min(head[0], std::min(head[1], K));
    if (n <= 0) return MVHDP_OK;
    HIPC(h, hipMemcpy(h->h_birth_keys.data(), h->d_birth_keys, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost));
    bool any = false;
    for (int r = 0; r < n; r++) {
        const int t = h->h_births[(size_t)2 + r];
        const long long key = h->h_birth_keys[(size_t)r];
        if (t < 0 || t >= K || key == LLONG_MAX || !h->h_inactive[t]) continue;          // (cannot happen: position r is passed only by a delta that reached it)
        const int mv = MVHDP_ACT_KEY_VIEW(key);
        if (mv < 0 || mv >= mm.M) FAIL(h, MVHDP_ERR_STATE, "births: bad activation key");
        h->h_inactive[t] = 0;
        h->h_alpha[(size_t)mv * (K + 1) + t] = h->h_alpha[(size_t)mv * (K + 1) + K];
        if (oc.n_activations++ == 0) oc.first_act = key;
        any = true;
    }
    if (any) {
        mm.first_inactive = -1;
        for (int k = 0; k < K; k++) if (h->h_inactive[k]) { mm.first_inactive = k; break; }
        HIPC(h, hipMemcpy(h->d_alpha, h->h_alpha.data(), h->h_alpha.size() * sizeof(double), hipMemcpyHostToDevice));
        HIPC(h, hipMemcpy(h->d_inactive, h->h_inactive.data(), (size_t)K, hipMemcpyHostToDevice));
    }
    return MVHDP_OK;
}

extern "C" int mvhdp_apply_delta(mvhdp_handle h, int32_t activated_topic, int32_t activated_modality)
{
    CHECK_H(h);
    MvModel& mm = h->mm;
    HIPC(h, hipSetDevice(h->device));
    HIPC(h, hipMemsetAsync(h->d_stats + ST_NEGATIVE, 0, sizeof(unsigned long long), h->stream));
    HIPC(h, mvhdp_launch_apply_delta(mm, h->d_stats, h->stream, h->delta16_used));
    h->delta16_used = false;
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

// ---------------------------------------------------------------------------------------------------------------
// The sweep: plan (mvhdp_plan.h, pure) -> enqueue (launches only, nothing waits) -> finish (one synchronisation:
// statistics, the next plan's histograms, the walk search).
// ---------------------------------------------------------------------------------------------------------------
static int ensure_slot_counts(mvhdp_ctx* h)
{
    // MvModel::nslots and the histograms come from the sweep kernels themselves; after assignments arrived from the host (or an
    // entity was abandoned, Q11) they are recounted from z: one pass
    MvModel& mm = h->mm;
    if (mm.D > 0 && !h->d_nslots) {
        HIPC(h, hipMalloc(&h->d_nslots, (size_t)mm.D * sizeof(uint16_t)));
        mm.nslots = h->d_nslots;
        h->nslots_valid = false;
    }
    if (h->nslots_valid) return MVHDP_OK;
    unsigned long long hist[MVHDP_HIST_BINS + MVHDP_ENT_BINS] = {0};
    HIPC(h, hipMemsetAsync(h->d_ovf_meta, 0, META_BYTES, h->stream));
    HIPC(h, mvhdp_launch_slot_hist(mm, (unsigned long long*)h->d_ovf_meta + META_HIST, h->stream));
    HIPC(h, hipMemcpyAsync(hist, (unsigned long long*)h->d_ovf_meta + META_HIST, sizeof hist, hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    std::copy(hist, hist + MVHDP_HIST_BINS, h->last_hist);
    std::copy(hist + MVHDP_HIST_BINS, hist + MVHDP_HIST_BINS + MVHDP_ENT_BINS, h->last_ent);
    h->nslots_valid = true;
    return MVHDP_OK;
}

static void fill_plan_in(mvhdp_ctx* h, uint32_t flags, bool debug, bool batch, PlanIn& in)
{
    const MvModel& mm = h->mm;
    in.K = mm.K; in.M = mm.M; in.D = mm.D;
    in.mdt = compute_max_doc_tokens(h);
    in.have_order = h->d_doc_order != nullptr && !h->tokens_desc.empty();
    for (int c = 0; c < 5; c++)
        in.n_longer[c] = in.have_order ? (int64_t)(std::upper_bound(h->tokens_desc.begin(), h->tokens_desc.end(), (int64_t)64 << c, std::greater<int64_t>()) - h->tokens_desc.begin()) : mm.D;
    std::copy(h->last_hist, h->last_hist + MVHDP_HIST_BINS, in.tok_hist);
    std::copy(h->last_ent, h->last_ent + MVHDP_ENT_BINS, in.ent_hist);
    in.flags = flags; in.debug = debug; in.batch = batch;
    in.trees_current = h->have_trees;
    in.unassigned = false;
    for (int m = 0; m < mm.M; m++) in.unassigned = in.unassigned || h->unassigned[m];
    in.first_inactive = mm.first_inactive;
    in.num_cus = h->num_cus; in.max_lds = h->max_lds;
    in.regs = h->regs;
}

static int alloc_debug(mvhdp_ctx* h, const mvhdp_debug* dbg, DebugBufs& db)
{
    const int K = h->mm.K, M = h->mm.M;
    hipStream_t s = h->stream;
    for (int m = 0; m < M; m++) {
        if (dbg->tok_dbg[m] && h->N[m] > 0) {
            void* p = nullptr;
            hipError_t e = hipMalloc(&p, (size_t)h->N[m] * 4 * sizeof(double));
            if (e != hipSuccess) { db.release(); HIPC(h, e); }
            db.to_free.push_back(p);
            hipMemsetAsync(p, 0, (size_t)h->N[m] * 4 * sizeof(double), s);
            db.tok_dbg[m] = (double*)p;
        }
    }
    if (dbg->n_trace > 0) {
        void *a = nullptr, *b = nullptr, *c = nullptr, *o = nullptr;
        const size_t n = (size_t)dbg->n_trace;
        if (hipMalloc(&a, n * 8) != hipSuccess || hipMalloc(&b, n * 4) != hipSuccess ||
            hipMalloc(&c, n * 4) != hipSuccess || hipMalloc(&o, n * (K + 1) * 8) != hipSuccess) {
            for (void* q : {a, b, c, o}) if (q) hipFree(q);
            db.release(); FAIL(h, MVHDP_ERR_HIP, "debug trace allocation failed");
        }
        db.to_free.push_back(a); db.to_free.push_back(b); db.to_free.push_back(c); db.to_free.push_back(o);
        hipMemcpyAsync(a, dbg->trace_doc, n * 8, hipMemcpyHostToDevice, s);
        hipMemcpyAsync(b, dbg->trace_view, n * 4, hipMemcpyHostToDevice, s);
        hipMemcpyAsync(c, dbg->trace_pos, n * 4, hipMemcpyHostToDevice, s);
        hipMemsetAsync(o, 0, n * (K + 1) * 8, s);
        db.n_trace = dbg->n_trace;
        db.trace_doc = (const int64_t*)a; db.trace_view = (const int32_t*)b; db.trace_pos = (const int32_t*)c;
        db.trace_out = (double*)o;
    }
    return MVHDP_OK;
}

// The control words and entity lists one segment's kernels work with (a second set exists for overlapped segments: two segments in flight)
struct SegCtl {
    unsigned int* class_counts;        // [MVHDP_N_CLASSES] lengths of the route pass's lists
    unsigned long long* qheads;        // [8] one work-queue head per kernel class
    int32_t* lists;                    // [MVHDP_N_CLASSES][D] entity lists (nullptr: the handle's own, allocated on first use)
};

// Route pass + every class kernel of segment `seg` on stream s (the wider classes on the side streams behind a fork event, joined
// back into s): positions seg, seg + nseg, ... of the longest-first order.  mk = the model these kernels read and update.
static hipError_t launch_segment_kernels(mvhdp_ctx* h, const SweepPlan& p, const MvModel& mk, const SweepLaunch& sl, int seg, hipStream_t s,
                                         const SegCtl& ctl, unsigned long long* d_stats)
{
    const MvModel& mm = h->mm;
    const int nseg = p.nseg;
    hipError_t e = hipSuccess;
    auto step = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    auto share = [&](int64_t P) -> int64_t { return P > seg ? (P - seg + nseg - 1) / nseg : 0; };
    const int64_t n_seg = share(mm.D);
    unsigned int* class_counts = ctl.class_counts;
    // entities a class kernel of this segment is sized for (an upper bound is enough: the queue is dynamic)
    double tok_tot = 0;
    for (int b = 0; b < MVHDP_HIST_BINS; b++) tok_tot += (double)h->last_hist[b];
    const bool sizes_known = h->last_ent[MVHDP_N_CLASSES] == 0 && tok_tot > 0;
    auto blocks_for = [&](int64_t n, const ClassLaunch& g) {
        const int64_t need = (n + (int64_t)g.wpb * MVHDP_DOC_BATCH - 1) / ((int64_t)g.wpb * MVHDP_DOC_BATCH);
        return (int)std::max<int64_t>(1, std::min<int64_t>(need, g.grid));
    };
    {
        const int64_t H_seg = p.route ? share(p.H) : 0;
        ClassifyArgs ca{};
        int32_t* lists = ctl.lists ? ctl.lists : h->d_lists;
        bool used_stream[PLAN_N_STREAMS] = {};
        if (p.route && H_seg > 0) {
            if (!ctl.lists && !h->d_lists) step(hipMalloc(&h->d_lists, (size_t)MVHDP_N_CLASSES * mm.D * sizeof(int32_t)));
            lists = ctl.lists ? ctl.lists : h->d_lists;
            if (!h->ev_fork) step(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            if (e != hipSuccess) return e;
            ca.order = h->d_doc_order; ca.n = H_seg; ca.start = seg; ca.stride = nseg;
            for (int c = 0; c < MVHDP_N_CLASSES; c++) { ca.class_map[c] = p.class_map[c]; ca.lists[c] = lists + (size_t)c * mm.D; }
            ca.check_views = compute_max_doc_tokens(h) > 65535 ? 1 : 0;
            ca.counts = class_counts;
            // (counted with this sweep's own counters: in a batch -- mvhdp_sweep_many -- every sweep resets the shared control block,
            // its statistics slot survives to the read-back at the end)
            ca.misrouted = d_stats + ST_MISCLASS;
            step(mvhdp_launch_classify(mm, ca, s));
            step(hipEventRecord(h->ev_fork, s));
        }
        // every class kernel of the segment, the widest (longest entities: the sweep's critical path) first; the plan says on which
        // stream (the widest on the handle's own, the primary on a side stream behind the fork event, see mvhdp_plan.h)
        for (int c = MVHDP_N_CLASSES - 1; c >= p.pc && e == hipSuccess; c--) {
            const ClassLaunch& g = p.cls[c];
            if (!g.used) continue;
            if (c != p.pc && !(p.route && H_seg > 0)) continue;   // nothing was routed in this segment: the primary alone
            hipStream_t st = s;
            if (g.stream != PLAN_STREAM_MAIN && p.route && H_seg > 0) {
                const int si = g.stream;
                if (!h->side[si]) {
                    // A and B (the classes of 8 and 16 rounds): high priority = a hardware-queue pool of their own (mvhdp_plan.h)
                    int least = 0, greatest = 0;
                    bool made = false;
                    if (si != PLAN_STREAM_C && h->side_priority && !(h->side_priority == 2 && si == PLAN_STREAM_D) && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest < least) {
                        made = hipStreamCreateWithPriority(&h->side[si], hipStreamNonBlocking, greatest) == hipSuccess;
                        if (!made) { (void)hipGetLastError(); h->side[si] = nullptr; }      // (a runtime without stream priorities: an ordinary stream then)
                    }
                    if (!made) step(hipStreamCreateWithFlags(&h->side[si], hipStreamNonBlocking));
                }
                if (!h->ev_join[si]) step(hipEventCreateWithFlags(&h->ev_join[si], hipEventDisableTiming));
                if (e != hipSuccess) return e;
                if (!used_stream[si]) step(hipStreamWaitEvent(h->side[si], h->ev_fork, 0));
                used_stream[si] = true;
                st = h->side[si];
            }
            // the side streams' kernels first: hold this stream for a moment before the primary takes the chip (mvhdp_launch_delay)
            if (c == p.pc && st == s) {
                bool side = false;
                for (int si = 1; si < PLAN_N_STREAMS; si++) side = side || used_stream[si];
                if (side) step(mvhdp_launch_delay(h->tu.fork_delay_us, s));
            }
            SweepLaunch sc = sl;
            sc.doc_counter = ctl.qheads + c;
            sc.wave_bytes = g.wave_bytes; sc.waves_per_block = g.wpb; sc.S_cap = g.S_cap;
            sc.walk = g.walk; sc.narrow = g.narrow;
            for (int m = 0; m < MVHDP_MAXM; m++) sc.walk_theta[m] = g.theta[m];
            int64_t n_c;
            if (c == p.pc) {
                // the primary: the routed entities that fit it, then everything too short to exceed it
                if (p.route && H_seg > 0) { sc.q_list = ca.lists[c]; sc.q_list_count = class_counts + c; }
                sc.q_order = h->d_doc_order; sc.q_order_start = seg + H_seg * nseg; sc.q_order_stride = nseg; sc.q_order_count = n_seg - H_seg;
                n_c = n_seg;
            } else {
                sc.q_list = ca.lists[c]; sc.q_list_count = class_counts + c;
                sc.q_order = nullptr; sc.q_order_start = 0; sc.q_order_count = 0;
                // entities this class can receive: those the plan's histogram puts there (and into classes mapped onto it), doubled
                // for the unevenness of a segment; everything of the prefix when the sizes are not known
                n_c = H_seg;
                if (sizes_known) {
                    unsigned long long cnt = 0;
                    for (int q = 0; q < MVHDP_N_CLASSES; q++) if (p.class_map[q] == c) cnt += h->last_ent[q];
                    n_c = std::min<int64_t>(H_seg, (int64_t)(2 * cnt / (unsigned)nseg) + 64);
                }
            }
            if (g.fast) step(mvhdp_launch_sweep_fast(mk, sc, g.r, blocks_for(n_c, g), p.debug, st));
            else step(mvhdp_launch_sweep(mk, sc, blocks_for(n_c, g), p.debug, st));
        }
        for (int si = 1; si < PLAN_N_STREAMS; si++) if (used_stream[si]) step(hipEventRecord(h->ev_join[si], h->side[si]));
        for (int si = 1; si < PLAN_N_STREAMS; si++) if (used_stream[si]) step(hipStreamWaitEvent(s, h->ev_join[si], 0));
    }
    return e;
}

// ---------------------------------------------------------------------------------------------------------------
// Overlapped segments: two segments of a sweep in flight, so that no segment border idles the chip (a segment's kernels end with
// a tail of one entity's time per wave, and the updater's pass and the tree rebuild used to run alone between two segments).
//
//   MVHDP_SWEEP_SEGMENT_APPLY | SEGMENT_OVERLAP (deterministic, the oracle follows it)
//     The counts (and their 16-bit mirror) are kept twice (B0/B1) and the deltas in three buffers used in turn.  Segment s samples
//     against the copy that holds the deltas of every segment up to s-2 and writes its own deltas into buffer s mod 3; when its
//     kernels are done the updater's kernel A(s) -- beside the kernels of segment s+1 -- adds the deltas of segments s-1 and s to the
//     OTHER copy; segment s+2 waits for A(s).  At the end the copy that missed the last segment takes it, and both copies are the
//     model again.  The F+trees are those of the sweep start for EVERY segment while n_wk and n_k advance: this DEVIATES from the reference,
//     whose updater refreshes the two touched leaves with every delta (UPD:242-260 -> FT:138-147), so that its trees track the counts --
//     here the tree-branch mass of a token and the count-based branch come from different model states, and the LL curves price that at
//     about half a reference sweep per sweep.  Rebuilding the trees per segment beside the samplers took a whole
//     segment's time in the one block slot per CU the samplers leave (profiles/r04_timeline_c4_oseg8_trees_per_segment.txt), which put
//     the rebuild back on the critical path; the mode is opt-in for a host that knows its tree branch to be small.
//         stream 0:  K(0) A(0) K(2) A(2) K(4) ...          A(s) behind K(s) on its stream and behind A(s-1) on the other one;
//         stream 1:  K(1) A(1) K(3) A(3) ...               K(s+2) behind A(s): while A(s) runs, stream 1 - s mod 2 is sampling
//   MVHDP_SWEEP_LIVE (racy by design, like the reference's updater)
//     One copy of the counts, updated in place; only the descent tables exist twice.  The trees of segment s+1 are rebuilt from the
//     live counts when segment s is about three fifths through (a one-wave gate kernel watches its work-queue head), into the tables
//     segment s-1 has finished with, and segment s+1's kernels follow at once: its first blocks fill in as segment s drains.
// The class kernels' grids leave one block per CU free (SweepPlan::overlap): the updater / tree kernels and the next segment's first
// blocks always find room.  No kernel ever waits for another one on the device (dependencies are stream events; the gate watches a
// kernel that itself waits for nothing), so nothing here can deadlock.
// ---------------------------------------------------------------------------------------------------------------
static int ensure_overlap_buffers(mvhdp_ctx* h, const SweepPlan& p)
{
    MvModel& mm = h->mm;
    const int64_t nrows = mm.rowbase[mm.M];
    const size_t cbytes = (size_t)(counts_len(h) + MVHDP_TAIL_WORDS) * sizeof(int32_t);
    auto& ov = h->ov;
    if (!ov.x1) HIPC(h, hipStreamCreateWithFlags(&ov.x1, hipStreamNonBlocking));

    if (!ov.ev_start) HIPC(h, hipEventCreateWithFlags(&ov.ev_start, hipEventDisableTiming));
    while (ov.ev_seg.size() < (size_t)3 * p.nseg) { hipEvent_t ev; HIPC(h, hipEventCreateWithFlags(&ev, hipEventDisableTiming)); ov.ev_seg.push_back(ev); }
    if (p.live) {                                        // (the trees exist twice for live sweeps only: the segmented sweep keeps the sweep-start trees)
        if (!ov.dtab2) HIPC(h, hipMalloc(&ov.dtab2, (size_t)nrows * mm.dt_nblk * 8 * sizeof(double)));
        if (!ov.root2) HIPC(h, hipMalloc(&ov.root2, (size_t)nrows * sizeof(double)));
        if (p.need_full && !ov.trees2) HIPC(h, hipMalloc(&ov.trees2, (size_t)nrows * 2 * mm.K * sizeof(double)));
    }
    if (!ov.ctl2) { HIPC(h, hipMalloc(&ov.ctl2, 16 * sizeof(unsigned long long))); HIPC(h, hipMemset(ov.ctl2, 0, 16 * sizeof(unsigned long long))); }
    if (p.route && !ov.lists2 && mm.D > 0) HIPC(h, hipMalloc(&ov.lists2, (size_t)MVHDP_N_CLASSES * mm.D * sizeof(int32_t)));
    if (p.seg_apply) {
        if (!ov.counts2) HIPC(h, hipMalloc(&ov.counts2, cbytes));
        if (!ov.counts16_2) HIPC(h, hipMalloc(&ov.counts16_2, (size_t)nrows * mm.K * sizeof(uint16_t)));
        if (!ov.delta2) { HIPC(h, hipMalloc(&ov.delta2, cbytes)); HIPC(h, hipMemset(ov.delta2, 0, cbytes)); }
        if (!ov.delta3) { HIPC(h, hipMalloc(&ov.delta3, cbytes)); HIPC(h, hipMemset(ov.delta3, 0, cbytes)); }
    }
    return MVHDP_OK;
}

static int enqueue_overlapped(mvhdp_ctx* h, const SweepPlan& p, uint32_t sweep_idx, uint64_t seed, const double* p_override,
                              unsigned long long* d_stats, hipEvent_t ev_k0, hipEvent_t ev_k1)
{
    MvModel& mm = h->mm;
    const int M = mm.M, nseg = p.nseg;
    const uint32_t flags = p.flags;
    int rc = ensure_overlap_buffers(h, p); if (rc) return rc;
    auto& ov = h->ov;
    hipStream_t X[2] = {h->stream, ov.x1};
    if (getenv("MVHDP_OVERLAP_SERIAL")) X[1] = h->stream;                          // (diagnostics: the same schedule on one stream)
    hipError_t e = hipSuccess;
    auto step = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    const int64_t clen = counts_len(h), nrows = mm.rowbase[M];
    const size_t cbytes = (size_t)clen * sizeof(int32_t);

    SweepLaunch sl{};
    sl.sweep_idx = sweep_idx; sl.seed_lo = (uint32_t)seed; sl.seed_hi = (uint32_t)(seed >> 32);
    sl.flags = flags & 0x7fffu; sl.S_cap = p.S_cap;
    sl.q_order_stride = 1;
    sl.nk_global = p.nk_global; sl.block_shared_bytes = p.block_shared_bytes;
    sl.live16 = p.live16 ? 1 : 0;
    sl.delta16 = p.delta16 ? 1 : 0;
    sl.stats = d_stats;
    sl.act_key = h->d_act_key;
    sl.slot_hist = (unsigned long long*)h->d_ovf_meta + META_HIST;

    if (M > 1) {
        if (!mm.p && mm.D > 0) step(hipMalloc(&mm.p, (size_t)mm.D * M * M * sizeof(double)));
        if (e == hipSuccess) {
            if (p_override) step(hipMemcpyAsync(mm.p, p_override, (size_t)mm.D * M * M * sizeof(double), hipMemcpyHostToDevice, X[0]));
            else step(mvhdp_launch_draw_p(mm, sweep_idx, sl.seed_lo, sl.seed_hi, X[0]));
        }
    }
    h->last_need_full = p.need_full;
    bool use_mirror = false;
    for (int c = 0; c < MVHDP_N_CLASSES; c++) use_mirror = use_mirror || (p.cls[c].used && p.cls[c].narrow);
    // the trees (and the mirror) of the sweep-start counts: what segments 0 and 1 sample from
    if (!(flags & MVHDP_SWEEP_REUSE_TREES)) {
        step(mvhdp_launch_build_trees(mm, false, p.need_full, X[0]));
        h->have_trees = true; h->full_trees = p.need_full; h->trees_inference = false;
    } else if (p.need_full && !h->full_trees) {
        step(mvhdp_launch_build_trees(mm, h->trees_inference, true, X[0]));
        h->full_trees = true;
    }
    // the two copies of the model
    MvModel B[2] = {mm, mm};
    if (p.live) { B[1].dtab = ov.dtab2; B[1].root = ov.root2; if (ov.trees2) B[1].trees = ov.trees2; }
    int32_t* D3[3] = {mm.delta, ov.delta2, ov.delta3};
    if (p.seg_apply) {
        B[1].counts = ov.counts2; B[1].counts16 = ov.counts16_2;
        step(hipMemcpyAsync(ov.counts2, mm.counts, cbytes, hipMemcpyDeviceToDevice, X[0]));
        step(hipMemcpyAsync(ov.counts16_2, mm.counts16, (size_t)nrows * mm.K * sizeof(uint16_t), hipMemcpyDeviceToDevice, X[0]));
        if (!h->delta_clean) step(hipMemsetAsync(mm.delta, 0, cbytes, X[0]));
        h->delta_clean = false;
        if (ov.deltas_dirty) { step(hipMemsetAsync(ov.delta2, 0, cbytes, X[0])); step(hipMemsetAsync(ov.delta3, 0, cbytes, X[0])); }
        ov.deltas_dirty = true;                                    // (cleared by mvhdp_sweep_finish / the batch's read-back when the sweep is seen to have finished)
    } else if (flags & MVHDP_SWEEP_NO_APPLY) {                     // (live, document shards: the caller wants after - before)
        step(mvhdp_launch_live_helper(mm, 0, d_stats, X[0])); h->delta_clean = false;
    }
    SegCtl ctl[2] = {{h->d_ovf_meta + META_CLASS_COUNTS, h->d_doc_counter, nullptr}, {(unsigned int*)(ov.ctl2 + 8), ov.ctl2, ov.lists2}};
    step(mvhdp_launch_ctl_reset(d_stats, ST_COUNT, h->d_act_key, (unsigned long long*)h->d_ovf_meta, META_WORDS64, nullptr, nullptr, X[0]));
    step(hipEventRecord(ev_k0, X[0]));
    step(hipEventRecord(ov.ev_start, X[0]));
    step(hipStreamWaitEvent(X[1], ov.ev_start, 0));
    auto ev_done = [&](int s) { return ov.ev_seg[(size_t)3 * s]; };
    auto ev_applied = [&](int s) { return ov.ev_seg[(size_t)3 * s + 1]; };
    auto ev_reset = [&](int s) { return ov.ev_seg[(size_t)3 * s + 2]; };
    auto share_of = [&](int seg, int64_t P) -> int64_t { return P > seg ? (P - seg + nseg - 1) / nseg : 0; };

    for (int seg = 0; seg < nseg && e == hipSuccess && mm.D > 0; seg++) {
        hipStream_t xs = X[seg & 1];
        const SegCtl& cs = ctl[seg & 1];
        MvModel mk;
        if (p.seg_apply) {
            // segment s reads the copy updated by A(s-2): copy 0 for segments 0 and 1, then (s-1) mod 2; its deltas go to buffer s mod 3
            // (A(seg - 2) sits in front of this segment on the same stream)
            mk = B[seg == 0 ? 0 : (seg - 1) & 1];
            mk.delta = D3[seg % 3];
        } else {
            // live: the counts are one; the trees of segment s are rebuilt, from the live counts, when segment s-1 is nearly through
            mk = mm;
            mk.delta = mm.counts;
            mk.dtab = B[seg & 1].dtab; mk.root = B[seg & 1].root; mk.trees = B[seg & 1].trees;
            if (seg >= 1 && !(flags & MVHDP_SWEEP_REUSE_TREES)) {
                step(hipStreamWaitEvent(xs, ev_reset(seg - 1), 0));          // (the head the gate watches has been reset for segment s-1)
                const int64_t n_prev = share_of(seg - 1, mm.D), H_prev = p.route ? share_of(seg - 1, p.H) : 0;
                // (three fifths through: the rebuild takes about a millisecond beside the samplers, and the next segment's first blocks
                // should be waiting when the current segment's queue runs dry)
                const unsigned long long thr = (unsigned long long)((n_prev - H_prev) * h->gate_pct / 100);
                step(mvhdp_launch_gate(ctl[(seg - 1) & 1].qheads + p.pc, thr, xs));
                MvModel tm = mm;
                tm.dtab = mk.dtab; tm.root = mk.root; tm.trees = mk.trees;
                if (p.live16) step(mvhdp_launch_build_trees_from_mirror(tm, p.need_full, xs, true));      // (the small-register flavour: beside the samplers)
                else step(mvhdp_launch_build_trees(tm, false, p.need_full, xs, true));
            } else if (seg >= 1) {
                mk.dtab = mm.dtab; mk.root = mm.root; mk.trees = mm.trees;     // REUSE_TREES: the host's trees, for every segment
            }
        }
        step(mvhdp_launch_ctl_reset(nullptr, 0, nullptr, nullptr, 0, cs.class_counts, cs.qheads, xs));
        step(hipEventRecord(ev_reset(seg), xs));
        step(launch_segment_kernels(h, p, mk, sl, seg, xs, cs, d_stats));
        step(hipEventRecord(ev_done(seg), xs));
        if (p.seg_apply) {
            // A(seg), beside the kernels of segment seg + 1: the other copy += deltas of seg - 1 and seg; its trees.  On this segment's
            // own stream, between K(seg) and K(seg + 2) -- which needs it anyway --, behind A(seg - 1) on the other stream.  (A stream
            // of its own shared a hardware queue with the side stream of the wider class kernels: the runtime has four.)
            if (seg >= 1) step(hipStreamWaitEvent(xs, ev_applied(seg - 1), 0));
            const MvModel& dst = B[(seg + 1) & 1];
            step(mvhdp_launch_apply2_counts(dst, D3[seg % 3], seg >= 1 ? D3[(seg - 1) % 3] : nullptr, use_mirror, d_stats + ST_NEGATIVE, xs));
            step(hipEventRecord(ev_applied(seg), xs));
        }
    }
    // everything back onto the handle's stream
    if (mm.D > 0 && e == hipSuccess) {
        if (p.seg_apply) {
            step(hipStreamWaitEvent(X[0], ev_applied(nseg - 1), 0));
            if (nseg >= 2) step(hipStreamWaitEvent(X[0], ev_done(nseg - 2), 0));
            // the copy A(nseg-1) did not write lacks the last segment's deltas
            const MvModel& last = B[(nseg + 1) & 1];
            step(mvhdp_launch_apply_sparse(last, D3[(nseg - 1) % 3], use_mirror, X[0]));
            h->have_trees = false;
            // (both copies equal now; copy 0 = mm is the model, every delta buffer is zero again)
        } else {
            step(hipStreamWaitEvent(X[0], ev_done(nseg - 1), 0));
            if (nseg >= 2) step(hipStreamWaitEvent(X[0], ev_done(nseg - 2), 0));
        }
    }
    if (p.live) {
        if (p.live16 && mm.D > 0) { step(mvhdp_launch_widen_mirror(mm, X[0])); h->have_trees = false; }
        if (flags & MVHDP_SWEEP_NO_APPLY) {
            step(mvhdp_launch_live_helper(mm, 1, d_stats, X[0]));
            if (!(flags & MVHDP_SWEEP_REUSE_TREES)) h->have_trees = false;
        } else step(mvhdp_launch_live_helper(mm, 2, d_stats, X[0]));
        if (!(flags & MVHDP_SWEEP_REUSE_TREES)) h->have_trees = false;     // (the last segments' trees sit in either table set)
    }
    step(hipEventRecord(ev_k1, X[0]));
    if (e != hipSuccess) HIPC(h, e);
    return MVHDP_OK;
}

// Everything one sweep puts on the device, in stream order; returns without waiting (except at the segment borders of a live /
// segmented sweep over a model with inactive topics, where the host performs the activation UPD:263-270).
// d_stats: [ST_COUNT] counters of THIS sweep; ev_k0/ev_k1: recorded around the sweep kernels.
static int enqueue_sweep(mvhdp_ctx* h, const SweepPlan& p, uint32_t sweep_idx, uint64_t seed, const double* p_override,
                         const DebugBufs* db, unsigned long long* d_stats, hipEvent_t ev_k0, hipEvent_t ev_k1, SweepOutcome& oc)
{
    if (p.overlap) return enqueue_overlapped(h, p, sweep_idx, seed, p_override, d_stats, ev_k0, ev_k1);   // (the plan never overlaps segments of a sweep with debug outputs)
    MvModel& mm = h->mm;
    const int M = mm.M;
    const uint32_t flags = p.flags;
    const int nseg = p.nseg;
    hipStream_t s = h->stream;
    hipError_t e = hipSuccess;
    auto step = [&](hipError_t r) { if (e == hipSuccess) e = r; };

    SweepLaunch sl{};
    sl.sweep_idx = sweep_idx; sl.seed_lo = (uint32_t)seed; sl.seed_hi = (uint32_t)(seed >> 32);
    sl.flags = flags & 0x7fffu; sl.S_cap = p.S_cap;
    if (h->tu.single_wave && p.live) sl.flags |= MVHDP_SL_STRICT_LIVE;
    if (p.live_rows && getenv("MVHDP_NO_ROW_SAMPLE")) sl.flags |= 0x2000u;                    // (measurement only: what the tree branch's row scan costs)
#ifdef MVHDP_PROBE
    if (const char* f = getenv("MVHDP_ATOMIC_PROBE")) sl.flags |= ((unsigned)atoi(f) & 7u) << 16;     // measurement build only (mvhdp_sweep_fast.hip)
#endif
    sl.q_order_stride = 1;
    sl.nk_global = p.nk_global; sl.block_shared_bytes = p.block_shared_bytes;
    sl.live16 = p.live16 ? 1 : 0;
    sl.live_rows = p.live_rows ? 1 : 0;
    sl.coef_lds = p.coef_lds ? 1 : 0;
    sl.delta16 = p.delta16 ? 1 : 0;
    sl.stats = d_stats;
    sl.act_key = h->d_act_key;
    // a live sweep in its live-rows form over a truncated HDP: topics are born chunk by chunk (births_begin / births_end), not one per segment
    const bool births = p.live_rows && !(flags & MVHDP_SWEEP_NO_APPLY) && mm.first_inactive >= 0 && p.only_seg < 0;
    if (births) { sl.births = h->d_births; sl.birth_keys = h->d_birth_keys; }
    sl.slot_hist = (unsigned long long*)h->d_ovf_meta + META_HIST;
    if (db) {
        for (int m = 0; m < M; m++) sl.tok_dbg[m] = db->tok_dbg[m];
        sl.n_trace = db->n_trace; sl.trace_doc = db->trace_doc; sl.trace_view = db->trace_view; sl.trace_pos = db->trace_pos; sl.trace_out = db->trace_out;
    }
    unsigned int* class_counts = h->d_ovf_meta + META_CLASS_COUNTS;

    if (M > 1) {
        // (drawing the view weights on a stream of their own beside the tree rebuild was tried in round 4: 0.15 ms of overlap on paper, but the
        // extra stream cost a segmented sweep 0.1-0.2 ms per segment in launch latency and a deferred one nothing gained: gpurun_out/r4t)
        if (!mm.p && mm.D > 0) step(hipMalloc(&mm.p, (size_t)mm.D * M * M * sizeof(double)));
        if (e == hipSuccess) {
            if (p_override) step(hipMemcpyAsync(mm.p, p_override, (size_t)mm.D * M * M * sizeof(double), hipMemcpyHostToDevice, s));
            else step(mvhdp_launch_draw_p(mm, sweep_idx, sl.seed_lo, sl.seed_hi, s));
        }
    }
    h->last_need_full = p.need_full;
    // live-rows form on the mirror: the heavy words keep stored trees, and a small kernel beside the samplers keeps those current
    // (heavy_refresh_kernel); not with one resident wave (mvhdp_tuning.single_wave: the sequential pin has no second kernel)
    const bool dummy_poll = getenv("MVHDP_DUMMY_POLL") != nullptr;        // (diagnostics: the refresher's waves beside ANY sweep, with nothing to rebuild)
    const bool refresher = ((p.live_rows && p.live16) || dummy_poll) && !h->tu.single_wave && !getenv("MVHDP_NO_HEAVY_REFRESH");
    auto live_rows_prepare = [&](bool from_mirror) -> hipError_t {
        if ((p.live16 || dummy_poll) && !h->d_heavy_list) {
            hipError_t e1 = hipMalloc(&h->d_heavy_list, (size_t)MVHDP_HEAVY_CAP * sizeof(int32_t));
            if (e1 == hipSuccess) e1 = hipMalloc(&h->d_heavy_ctl, 2 * sizeof(unsigned int));
            if (e1 != hipSuccess) return e1;
        }
        if (refresher && !h->rf_stream) {
            int least = 0, greatest = 0;
            if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || greatest >= least ||
                hipStreamCreateWithPriority(&h->rf_stream, hipStreamNonBlocking, greatest) != hipSuccess) {
                (void)hipGetLastError(); h->rf_stream = nullptr;
                hipError_t e1 = hipStreamCreateWithFlags(&h->rf_stream, hipStreamNonBlocking); if (e1 != hipSuccess) return e1;
            }
            hipError_t e1 = hipEventCreateWithFlags(&h->ev_rf_go, hipEventDisableTiming);
            if (e1 == hipSuccess) e1 = hipEventCreateWithFlags(&h->ev_rf_done, hipEventDisableTiming);
            if (e1 != hipSuccess) return e1;
        }
        return mvhdp_launch_live_rows_prepare(mm, from_mirror, p.live16, p.live16 ? h->d_heavy_list : nullptr, p.live16 ? h->d_heavy_ctl : nullptr, MVHDP_HEAVY_CAP,
                                              p.live16 ? 512 : 256 /* cells of one register batch of a row: row_sample_live */, s);
    };
    auto rebuild_trees = [&]() {
        step(mvhdp_launch_build_trees(mm, false, p.need_full, s));
        h->have_trees = true; h->full_trees = p.need_full; h->trees_inference = false;
    };
    if (p.live_rows) {
        // no stored trees: the weight classes and the mirror, the coefficients and tree[1] of every word from the sweep-start counts
        step(live_rows_prepare(false));
        h->have_trees = false; h->full_trees = false;
    } else if (!(flags & MVHDP_SWEEP_REUSE_TREES)) rebuild_trees();
    if (dummy_poll && !p.live_rows) {
        if (!h->d_heavy_list) { step(hipMalloc(&h->d_heavy_list, (size_t)MVHDP_HEAVY_CAP * sizeof(int32_t))); step(hipMalloc(&h->d_heavy_ctl, 2 * sizeof(unsigned int))); }
        if (!h->rf_stream) { step(hipStreamCreateWithFlags(&h->rf_stream, hipStreamNonBlocking)); step(hipEventCreateWithFlags(&h->ev_rf_go, hipEventDisableTiming)); step(hipEventCreateWithFlags(&h->ev_rf_done, hipEventDisableTiming)); }
        step(hipMemsetAsync(h->d_heavy_ctl, 0, 2 * sizeof(unsigned int), s));
    }
    else if (p.need_full && !h->full_trees) {
        step(mvhdp_launch_build_trees(mm, h->trees_inference, true, s));
        h->full_trees = true;
    }
    // where the sweep's atomics land: the delta replica (deferred), or the shared counts themselves (live)
    MvModel mk = mm;
    if (p.live) {
        mk.delta = mm.counts;
        if (flags & MVHDP_SWEEP_NO_APPLY) { step(mvhdp_launch_live_helper(mm, 0, d_stats, s)); h->delta_clean = false; }
    } else if (!p.frozen) {                                      // a frozen sweep queues nothing (WRK:587) and leaves the buffer alone
        if (!h->delta_clean) {
            step(hipMemsetAsync(mm.delta, 0, (size_t)counts_len(h) * sizeof(int32_t), s));
            if (h->delta16_used) step(hipMemsetD16Async(mm.delta16, (unsigned short)0x8000, (size_t)(mm.rowbase[M] * mm.K), s));   // (a sweep that failed before its apply pass)
            h->delta16_used = false;
        }
        h->delta_clean = false;
        if (p.delta16) h->delta16_used = true;                   // (cleared by the apply pass, which folds the 16-bit cells in and re-biases them)
    }
    // this sweep's counters, "no activation yet", the histograms for the next plan, the class list lengths, the queue heads: one launch
    step(mvhdp_launch_ctl_reset(d_stats, ST_COUNT, h->d_act_key, (unsigned long long*)h->d_ovf_meta, META_WORDS64, nullptr, h->d_doc_counter, s));
    if (births) step(births_begin(h, s));
    step(hipEventRecord(ev_k0, s));

    const int seg_lo = p.only_seg >= 0 ? p.only_seg : 0, seg_hi = p.only_seg >= 0 ? p.only_seg + 1 : nseg;
    for (int seg = seg_lo; seg < seg_hi && e == hipSuccess && mm.D > 0; seg++) {
        // segment seg = positions seg, seg + nseg, ... of the order
        if (seg > seg_lo) {
            // UPD:263-270 acts as soon as a delta lands on an inactive topic, and the samplers then draw the NEXT inactive
            // index (WRK:523-526): a sweep whose counts are kept current does the same at every segment border -- the
            // segment's first such delta (by entity, view, position) activates its topic before the next segment starts.
            // Not with MVHDP_SWEEP_NO_APPLY: there the caller reduces the key over all document shards first.
            if (births && e == hipSuccess) {
                const int rc = births_end(h, s, oc);                                     // every topic the segment's deltas reached
                if (rc != MVHDP_OK) return rc;
                mk.first_inactive = mm.first_inactive;
                step(births_begin(h, s));                                                // (an empty list once every topic is active)
                step(mvhdp_launch_ctl_reset(nullptr, 0, h->d_act_key, nullptr, 0, nullptr, nullptr, s));
            } else
            if ((p.live || p.seg_apply) && !(flags & MVHDP_SWEEP_NO_APPLY) && mm.first_inactive >= 0 && e == hipSuccess) {
                long long key = LLONG_MAX;
                step(hipMemcpyAsync(&key, h->d_act_key, sizeof key, hipMemcpyDeviceToHost, s));
                step(hipStreamSynchronize(s));
                if (e == hipSuccess && key != LLONG_MAX) {
                    const int rc = apply_activation(h, MVHDP_ACT_KEY_TOPIC(key), MVHDP_ACT_KEY_VIEW(key));
                    if (rc != MVHDP_OK) return rc;
                    if (oc.n_activations++ == 0) oc.first_act = key;
                    mk.first_inactive = mm.first_inactive;
                    step(mvhdp_launch_ctl_reset(nullptr, 0, h->d_act_key, nullptr, 0, nullptr, nullptr, s));
                }
            }
            if (p.seg_apply) {
                // the updater catches up before the next segment (UPD:197-218) and the trees follow: tokensPerTopic first (every
                // leaf needs all of it), then ONE pass per row: counts += delta, delta = 0, the row's tree and 16-bit mirror
                step(mvhdp_launch_apply_nk(mm, d_stats + ST_NEGATIVE, s));
                step(mvhdp_launch_build_trees_rows(mm, false, p.need_full, 0, mm.rowbase[M], true, d_stats + ST_NEGATIVE, s));
                h->have_trees = true; h->full_trees = p.need_full; h->trees_inference = false;
            } else if (p.live_rows) {                                                    // tokensPerTopic has landed: new coefficients, exact roots
                step(live_rows_prepare(p.live16));
            } else if (p.live && !(flags & MVHDP_SWEEP_REUSE_TREES) && (h->live_tree_every <= 1 || seg % h->live_tree_every == 0)) {   // from the live counts
                if (p.live16) {                                                          // (the light rows' live counts are in the mirror)
                    step(mvhdp_launch_build_trees_from_mirror(mm, p.need_full, s));
                    h->have_trees = true; h->full_trees = p.need_full; h->trees_inference = false;
                } else rebuild_trees();
            }
            step(mvhdp_launch_ctl_reset(nullptr, 0, nullptr, nullptr, 0, class_counts, h->d_doc_counter, s));
        }
        if (e != hipSuccess) break;
        if (refresher) {
            // the refresher starts behind the segment's prepare pass and runs beside the samplers; `stop` is set in stream order behind them
            step(hipEventRecord(h->ev_rf_go, s));
            step(hipStreamWaitEvent(h->rf_stream, h->ev_rf_go, 0));
            if (!getenv("MVHDP_REFRESH_NO_KERNEL")) step(mvhdp_launch_heavy_refresh(mm, h->d_heavy_list, h->d_heavy_ctl, MVHDP_HEAVY_CAP, getenv("MVHDP_REFRESH_BLOCKS") ? atoi(getenv("MVHDP_REFRESH_BLOCKS")) : 16, h->rf_stream));
            step(hipEventRecord(h->ev_rf_done, h->rf_stream));
        }
        step(launch_segment_kernels(h, p, mk, sl, seg, s, SegCtl{class_counts, h->d_doc_counter, nullptr}, d_stats));
        if (refresher) {
            // (whatever failed above, the stop word is written and the refresher waited for: it also ends by itself after two seconds)
            hipError_t e2 = mvhdp_launch_set_u32(h->d_heavy_ctl + 1, 1u, s);
            if (e2 == hipSuccess) e2 = hipStreamWaitEvent(s, h->ev_rf_done, 0);
            step(e2);
        }
    }
    if (p.seg_apply && mm.D > 0) {                               // the last segment's deltas (the trees are rebuilt by whoever needs them next)
        step(mvhdp_launch_apply_delta(mm, d_stats, s));
        h->have_trees = false;
    }
    if (p.live) {
        if (p.live16 && mm.D > 0) { step(mvhdp_launch_widen_mirror(mm, s)); h->have_trees = false; }   // the 32-bit table is the model again (the mirror is now ahead of the trees)
        if (flags & MVHDP_SWEEP_NO_APPLY) {
            step(mvhdp_launch_live_helper(mm, 1, d_stats, s));   // delta = after - before, counts = snapshot
            if (nseg > 1 && !(flags & MVHDP_SWEEP_REUSE_TREES)) h->have_trees = false;   // the trees of the last segment belong to the live counts, not to the snapshot
        } else step(mvhdp_launch_live_helper(mm, 2, d_stats, s));                        // UPD:202-215
    }
    step(hipEventRecord(ev_k1, s));
    if (e != hipSuccess) HIPC(h, e);
    return MVHDP_OK;
}

static void stats_from_counters(const unsigned long long* hs, long long act, mvhdp_sweep_stats& st)
{
    st = mvhdp_sweep_stats{};
    st.tokens = (int64_t)hs[ST_TOKENS]; st.changed = (int64_t)hs[ST_CHANGED];
    st.new_mass_cnt = (int64_t)hs[ST_NEW]; st.topic_doc_mass_cnt = (int64_t)hs[ST_DOC];
    st.word_ftree_mass_cnt = (int64_t)hs[ST_TREE]; st.oov_skipped = (int64_t)hs[ST_OOV];
    st.aborted_docs = (int64_t)hs[ST_ABORT]; st.exact_fallbacks = (int64_t)hs[ST_FALLBACK];
    st.activation_key = act;
    st.activated_topic = -1; st.activated_modality = -1;
    if (act != LLONG_MAX) { st.activated_topic = MVHDP_ACT_KEY_TOPIC(act); st.activated_modality = MVHDP_ACT_KEY_VIEW(act); }
}

// after the synchronisation: what the sweep(s) left for the next plan, and the walk search
static void learn_from_sweep(mvhdp_ctx* h, const SweepPlan& p, const unsigned long long* hs, const unsigned long long* meta_hist, double kernel_ms, bool comparable)
{
    const MvModel& mm = h->mm;
    bool any_walk = false;
    for (int c = 0; c < MVHDP_N_CLASSES; c++) any_walk = any_walk || (p.cls[c].used && p.cls[c].walk);
    if (any_walk) h->wt.measured(mm.M, p.nseg, hs + ST_VIEW_BASE);
    if (mm.D > 0 && p.only_seg >= 0) {
        // a single segment was swept: its histograms describe a part of the entities only -- the earlier ones stay (every plan of
        // a single-segment sweep launches whatever is reachable), only an abandoned entity asks for a recount
        if (meta_hist[MVHDP_HIST_BINS + MVHDP_N_CLASSES] != 0) h->nslots_valid = false;
    } else if (mm.D > 0) {
        std::copy(meta_hist, meta_hist + MVHDP_HIST_BINS, h->last_hist);
        std::copy(meta_hist + MVHDP_HIST_BINS, meta_hist + MVHDP_HIST_BINS + MVHDP_ENT_BINS, h->last_ent);
        h->nslots_valid = h->last_ent[MVHDP_N_CLASSES] == 0;        // an abandoned entity (Q11): its list is recounted before the next sweep
    }
    if (h->dbg_env) {
        fprintf(stderr, "[mvhdp] walk threshold %.2f (base %.2f, phase %d, dir %+d, group %d); tree branch %.4f of tokens, walked on demand %.4f; per view tree share:",
                (double)h->wt.walk_probe_i / MVHDP_WALK_BINS, (double)h->wt.walk_i / MVHDP_WALK_BINS, h->wt.walk_phase, h->wt.walk_dir, h->wt.walk_cls,
                (double)hs[ST_TREE] / std::max<double>(1.0, (double)hs[ST_TOKENS]), (double)hs[ST_ONDEMAND] / std::max<double>(1.0, (double)hs[ST_TOKENS]));
        for (int m = 0; m < mm.M; m++) {
            const double n = std::max<double>(1.0, (double)hs[ST_VIEW_BASE + m * MVHDP_VIEW_STATS]);
            fprintf(stderr, " %.3f", hs[ST_VIEW_BASE + m * MVHDP_VIEW_STATS + 1] / n);
        }
        fprintf(stderr, "\n");
        if (hs[ST_T_TOTAL])
            fprintf(stderr, "[mvhdp] wave cycles: queue %.1f%% prologue %.1f%% view setup %.1f%% chunk head %.1f%% tokens %.1f%% chunk end %.1f%% | %.0f cycles per token per wave\n",
                    100.0 * hs[ST_T_QUEUE] / hs[ST_T_TOTAL], 100.0 * hs[ST_T_PROLOGUE] / hs[ST_T_TOTAL], 100.0 * hs[ST_T_VIEW] / hs[ST_T_TOTAL],
                    100.0 * hs[ST_T_CHUNK_HEAD] / hs[ST_T_TOTAL], 100.0 * hs[ST_T_TOKENS] / hs[ST_T_TOTAL], 100.0 * hs[ST_T_CHUNK_END] / hs[ST_T_TOTAL],
                    (double)hs[ST_T_TOTAL] / std::max<double>(1.0, (double)hs[ST_TOKENS]));
        if (hs[ST_T_ROWS])
            fprintf(stderr, "[mvhdp] live rows: %.0f cycles per tree-branch token, %.0f of them until the row is there (%.1f%% of the waves' time; %llu such tokens)\n",
                    (double)hs[ST_T_ROWS] / std::max<double>(1.0, (double)hs[ST_ONDEMAND]), (double)hs[ST_T_ROWS_WAIT] / std::max<double>(1.0, (double)hs[ST_ONDEMAND]),
                    100.0 * hs[ST_T_ROWS] / hs[ST_T_TOTAL], hs[ST_ONDEMAND]);
        if (hs[ST_N_WAVES])
            fprintf(stderr, "[mvhdp] cycles per token in a wave's first / second / later entities: %.0f / %.0f / %.0f (tokens %llu / %llu / %llu); per wave: block init %.0f cycles, wait + flush at the end %.0f, whole wave %.0f\n",
                    (double)hs[ST_T_ENT0] / std::max<double>(1.0, (double)hs[ST_N_ENT0]), (double)hs[ST_T_ENT1] / std::max<double>(1.0, (double)hs[ST_N_ENT1]),
                    (double)hs[ST_T_ENT2] / std::max<double>(1.0, (double)hs[ST_N_ENT2]), hs[ST_N_ENT0], hs[ST_N_ENT1], hs[ST_N_ENT2],
                    (double)hs[ST_T_INIT] / hs[ST_N_WAVES], (double)hs[ST_T_FLUSH] / hs[ST_N_WAVES], (double)hs[ST_T_TOTAL] / hs[ST_N_WAVES]);
    }
    const double tokens = (double)hs[ST_TOKENS];
    h->wt.observe(comparable && tokens > 0, p.walk_cfg, mm.M, tokens > 0 ? kernel_ms * 1e6 / tokens : 0.0);
}

static void debug_print_plan(const mvhdp_ctx* h, const SweepPlan& p, uint32_t sweep_idx)
{
    double tot = 0, b1 = 0, b2 = 0;                           // token share by topic-list size: <= 64, <= 128 slots
    for (int b = 0; b < MVHDP_HIST_BINS; b++) { tot += (double)h->last_hist[b]; if (b < 1) b1 += (double)h->last_hist[b]; if (b < 2) b2 += (double)h->last_hist[b]; }
    fprintf(stderr, "[mvhdp] sweep %u: %s primary class %d, routed prefix %lld, S_cap %d, %d segment(s); tokens in lists <=64: %.4f <=128: %.4f; kernels:",
            sweep_idx, p.fast ? "register-resident" : "generic", p.pc, (long long)p.H, p.S_cap, p.nseg, b1 / std::max(1.0, tot), b2 / std::max(1.0, tot));
    for (int c = 0; c < MVHDP_N_CLASSES; c++)
        if (p.cls[c].used)
            fprintf(stderr, " [class %d %s grid %d wpb %d lds %zu stream %d%s%s theta0 %.2f ents %llu]", c, p.cls[c].fast ? "fast" : "generic", p.cls[c].grid, p.cls[c].wpb, p.cls[c].lds,
                    p.cls[c].stream, p.cls[c].walk ? " walk" : "", p.cls[c].narrow ? " narrow" : "", p.cls[c].theta[0], (unsigned long long)h->last_ent[c]);
    fprintf(stderr, "\n");
}

static int sweep_preconditions(mvhdp_ctx* h, uint32_t flags)
{
    int rc = require_corpus(h); if (rc) return rc;
    if (!h->have_hyper) FAIL(h, MVHDP_ERR_STATE, "sweep before set_hyper");
    if (!h->have_counts) FAIL(h, MVHDP_ERR_STATE, "sweep before build_counts/set_counts");
    if ((flags & (MVHDP_SWEEP_REUSE_TREES | MVHDP_SWEEP_FROZEN)) && !h->have_trees) FAIL(h, MVHDP_ERR_STATE, "REUSE_TREES / FROZEN without trees");
    if (h->rows_applied >= 0) FAIL(h, MVHDP_ERR_STATE, "sweep: an mvhdp_apply_delta_begin bracket is open (call mvhdp_apply_delta_end)");
    if (h->delta_pending && !(flags & MVHDP_SWEEP_FROZEN))
        FAIL(h, MVHDP_ERR_STATE, "sweep: the previous NO_APPLY sweep's deltas have not been applied (mvhdp_apply_delta)");
    if (h->counts_stale && !(flags & MVHDP_SWEEP_FROZEN))
        FAIL(h, MVHDP_ERR_STATE, "sweep: the assignments were replaced after the counts were built (call mvhdp_build_counts, mvhdp_set_counts or mvhdp_counts_written first)");
    return MVHDP_OK;
}

static int set_generic_lds(mvhdp_ctx* h, const SweepPlan& p)
{
    size_t lds = 0;                                          // the generic kernel may need > 64 KiB of dynamic LDS
    for (int c = 0; c < MVHDP_N_CLASSES; c++) if (p.cls[c].used && !p.cls[c].fast) lds = std::max(lds, p.cls[c].lds);
    if (lds > 65536 && lds > h->lds_attr_set) { HIPC(h, mvhdp_sweep_set_max_lds(lds)); h->lds_attr_set = lds; }
    return MVHDP_OK;
}

int mvhdp_sweep_begin(mvhdp_ctx* h, uint32_t sweep_idx, uint64_t seed, uint32_t flags, const double* p_override, const mvhdp_debug* dbg, PendingSweep& ps)
{
    int rc = sweep_preconditions(h, flags); if (rc) return rc;
    HIPC(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    rc = ensure_slot_counts(h); if (rc) return rc;
    ps.flags = flags; ps.dbg = dbg; ps.debug = dbg != nullptr; ps.oc = SweepOutcome();
    PlanIn in; fill_plan_in(h, flags, ps.debug, false, in);
    plan_sweep(in, h->tu, h->wt, ps.p);
    if (ps.p.err) FAIL(h, ps.p.err, ps.p.msg);
    rc = set_generic_lds(h, ps.p); if (rc) return rc;
    if (h->dbg_env) debug_print_plan(h, ps.p, sweep_idx);
    if (ps.debug) { rc = alloc_debug(h, dbg, ps.db); if (rc) return rc; }
    HIPC(h, hipEventRecord(h->ev[0], s));
    ps.births = ps.p.live_rows && !(flags & MVHDP_SWEEP_NO_APPLY) && h->mm.first_inactive >= 0 && ps.p.only_seg < 0;   // (as enqueue_sweep decides it)
    rc = enqueue_sweep(h, ps.p, sweep_idx, seed, p_override, ps.debug ? &ps.db : nullptr, h->d_stats, h->ev[1], h->ev[2], ps.oc);
    if (rc) { ps.db.release(); return rc; }
    // counters | activation key | histograms: one copy into the handle's pinned buffer, in stream order behind the kernels
    hipError_t e = hipMemcpyAsync(h->h_ctl, h->d_ctl, (ST_COUNT + 1 + META_WORDS64) * sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) { ps.db.release(); HIPC(h, e); }
    ps.open = true;
    return MVHDP_OK;
}

int mvhdp_sweep_finish(mvhdp_ctx* h, PendingSweep& ps, mvhdp_sweep_stats* stats)
{
    if (!ps.open) FAIL(h, MVHDP_ERR_STATE, "sweep_finish without sweep_begin");
    ps.open = false;
    MvModel& mm = h->mm;
    const SweepPlan& p = ps.p;
    const uint32_t flags = ps.flags;
    HIPC(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) { ps.db.release(); HIPC(h, e); }
    const unsigned long long* hs = h->h_ctl;
    const long long act = (long long)h->h_ctl[ST_COUNT];
    const unsigned long long* meta = h->h_ctl + ST_COUNT + 1;
    if (hs[ST_MISCLASS] || meta[META_MISROUTED]) {
        ps.db.release();
        h->nslots_valid = false;
        FAIL(h, MVHDP_ERR_HIP, "internal: an entity reached a sweep kernel variant that cannot hold its topic list");
    }
    if (ps.debug) {
        const mvhdp_debug* dbg = ps.dbg;
        for (int m = 0; m < mm.M; m++)
            if (ps.db.tok_dbg[m] && e == hipSuccess) e = hipMemcpy(dbg->tok_dbg[m], ps.db.tok_dbg[m], (size_t)h->N[m] * 4 * sizeof(double), hipMemcpyDeviceToHost);
        if (ps.db.n_trace > 0 && e == hipSuccess) e = hipMemcpy(dbg->trace_out, ps.db.trace_out, (size_t)ps.db.n_trace * (mm.K + 1) * sizeof(double), hipMemcpyDeviceToHost);
        ps.db.release();
        if (e != hipSuccess) HIPC(h, e);
    }

    mvhdp_sweep_stats st;
    stats_from_counters(hs, act, st);
    // every entity was visited and none abandoned: no token of a known type is unassigned any more (WRK:557)
    if (!p.frozen && p.only_seg < 0 && st.aborted_docs == 0) for (int m = 0; m < mm.M; m++) h->unassigned[m] = false;
    const unsigned long long negatives = hs[ST_NEGATIVE];
    // (the pinned buffer belongs to the handle: a sweep begun on it before this one returns would overwrite it -- what the
    // planner learns from is copied out first)
    unsigned long long hist_keep[MVHDP_HIST_BINS + MVHDP_ENT_BINS];
    std::copy(meta + META_HIST, meta + META_HIST + MVHDP_HIST_BINS + MVHDP_ENT_BINS, hist_keep);
    unsigned long long hs_keep[ST_COUNT];
    std::copy(hs, hs + ST_COUNT, hs_keep);
    int n_activations = ps.oc.n_activations;
    int ret = MVHDP_OK;
    if (p.frozen) {
        // nothing was queued (WRK:587): the delta buffer is untouched
    } else if (flags & MVHDP_SWEEP_NO_APPLY) {
        h->delta_pending = true;
    } else if (p.live || p.seg_apply) {
        // the counts are already updated; what is left of the updater's work is the topic activation of the last segment
        h->have_trees = false;
        if (p.seg_apply) h->delta_clean = true;                      // apply_delta_kernel zeroed what it added
        if (p.seg_apply && p.overlap) h->ov.deltas_dirty = false;
        if (ps.births) {                                             // the last segment's births (the earlier segments' were applied at their borders)
            ret = births_end(h, s, ps.oc);
            n_activations = ps.oc.n_activations;
        } else {
            ret = apply_activation(h, st.activated_topic, st.activated_modality);
            if (st.activated_topic >= 0) n_activations++;
        }
        if (ps.oc.first_act != LLONG_MAX) {                          // report the sweep's first activation
            st.activation_key = ps.oc.first_act;
            st.activated_topic = MVHDP_ACT_KEY_TOPIC(ps.oc.first_act); st.activated_modality = MVHDP_ACT_KEY_VIEW(ps.oc.first_act);
        }
        if (ret == MVHDP_OK && negatives) { h->err = "a topic count went below zero (UPD:202-215)"; ret = MVHDP_ERR_NEGATIVE_COUNT; }
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
    const bool comparable = p.fast && !ps.debug && !h->tu.walk_fixed && !(flags & (MVHDP_SWEEP_FROZEN | MVHDP_SWEEP_EXACT_CHAIN));
    learn_from_sweep(h, p, hs_keep, hist_keep, ms_k, comparable);
    if (stats) *stats = st;
    return ret;
}

extern "C" int mvhdp_sweep(mvhdp_handle h, uint32_t sweep_idx, uint64_t seed, uint32_t flags,
                           const double* p_override, const mvhdp_debug* dbg, mvhdp_sweep_stats* stats)
{
    CHECK_H(h);
    PendingSweep ps;
    const int rc = mvhdp_sweep_begin(h, sweep_idx, seed, flags, p_override, dbg, ps);
    if (rc) return rc;
    return mvhdp_sweep_finish(h, ps, stats);
}

// n sweeps (indices first_idx .. first_idx + n - 1) put on the device back to back: ONE plan, no host round trip between the
// sweeps (the iteration loop PTM:1146-1239 without its per-iteration barrier on the host), the statistics of every sweep
// collected on the device and read once.  Same integers as n calls of mvhdp_sweep: the plan decides which kernel variant visits
// an entity and when a word tree is walked, never what is sampled.  Falls back to n single calls where a sweep needs the host in
// between (inactive topics waiting for activation, view weights or debug output from the host, NO_APPLY).
extern "C" int mvhdp_sweep_many(mvhdp_handle h, uint32_t first_idx, int32_t n, uint64_t seed, uint32_t flags, mvhdp_sweep_stats* stats)
{
    CHECK_H(h);
    if (n < 0) FAIL(h, MVHDP_ERR_INVALID_ARG, "sweep_many: n < 0");
    if (n == 0) return MVHDP_OK;
    MvModel& mm = h->mm;
    const bool frozen = (flags & MVHDP_SWEEP_FROZEN) != 0;
    // (REUSE_TREES without FROZEN: the first sweep's update makes the trees stale, and a single call then says so -- MVHDP_ERR_STATE at
    // the second sweep; the batch must not sample on from stale trees and a stale mirror instead)
    if (n == 1 || mm.first_inactive >= 0 || ((flags & MVHDP_SWEEP_NO_APPLY) && !frozen) || ((flags & MVHDP_SWEEP_REUSE_TREES) && !frozen) || n > 4096) {
        for (int i = 0; i < n; i++) {
            const int rc = mvhdp_sweep(h, first_idx + (uint32_t)i, seed, flags, nullptr, nullptr, stats ? stats + i : nullptr);
            if (rc) return rc;
        }
        return MVHDP_OK;
    }
    int rc = sweep_preconditions(h, flags); if (rc) return rc;
    HIPC(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    rc = ensure_slot_counts(h); if (rc) return rc;
    PlanIn in; fill_plan_in(h, flags, false, true, in);
    SweepPlan p;
    plan_sweep(in, h->tu, h->wt, p);
    if (p.err) FAIL(h, p.err, p.msg);
    rc = set_generic_lds(h, p); if (rc) return rc;
    if (h->dbg_env) debug_print_plan(h, p, first_idx);
    if (h->stats_many_cap < n) {
        if (h->d_stats_many) { hipFree(h->d_stats_many); h->d_stats_many = nullptr; h->stats_many_cap = 0; }
        HIPC(h, hipMalloc(&h->d_stats_many, (size_t)n * ST_COUNT * sizeof(unsigned long long)));
        h->stats_many_cap = n;
    }
    while ((int)h->ev_many.size() < 2 * n) { hipEvent_t ev; HIPC(h, hipEventCreate(&ev)); h->ev_many.push_back(ev); }
    HIPC(h, hipEventRecord(h->ev[0], s));
    for (int i = 0; i < n; i++) {
        SweepOutcome oc;
        unsigned long long* d_st = h->d_stats_many + (size_t)i * ST_COUNT;
        rc = enqueue_sweep(h, p, first_idx + (uint32_t)i, seed, nullptr, nullptr, d_st, h->ev_many[2 * i], h->ev_many[2 * i + 1], oc);
        if (rc) return rc;
        if (!p.frozen && !p.live && !p.seg_apply) {
            // the updater's pass (UPD:197-218) in stream order; negative counts are counted into this sweep's counters
            HIPC(h, mvhdp_launch_apply_delta(mm, d_st, s, h->delta16_used));
            h->delta16_used = false;
            h->have_trees = false; h->delta_clean = true;
        } else if (p.live || p.seg_apply) {
            h->have_trees = false;
            if (p.seg_apply) h->delta_clean = true;
        }
    }
    HIPC(h, hipEventRecord(h->ev[3], s));
    std::vector<unsigned long long> hs((size_t)n * ST_COUNT);
    unsigned long long meta[META_WORDS64] = {0};
    HIPC(h, hipMemcpyAsync(hs.data(), h->d_stats_many, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIPC(h, hipMemcpyAsync(meta, h->d_ovf_meta, sizeof meta, hipMemcpyDeviceToHost, s));
    HIPC(h, hipStreamSynchronize(s));
    if (p.seg_apply && p.overlap) h->ov.deltas_dirty = false;
    float ms_t = 0;
    hipEventElapsedTime(&ms_t, h->ev[0], h->ev[3]);
    int ret = MVHDP_OK;
    double ms_sum = 0;
    for (int i = 0; i < n; i++) {
        const unsigned long long* hi = hs.data() + (size_t)i * ST_COUNT;
        mvhdp_sweep_stats st;
        stats_from_counters(hi, LLONG_MAX, st);
        float ms_k = 0;
        hipEventElapsedTime(&ms_k, h->ev_many[2 * i], h->ev_many[2 * i + 1]);
        st.sweep_kernel_ms = ms_k; st.total_ms = ms_t / n;
        ms_sum += ms_k;
        if (stats) stats[i] = st;
        if (hi[ST_MISCLASS]) { h->nslots_valid = false; FAIL(h, MVHDP_ERR_HIP, "internal: an entity reached a sweep kernel variant that cannot hold its topic list"); }
        if (hi[ST_NEGATIVE] && ret == MVHDP_OK) { h->err = "a topic count went below zero (UPD:202-215)"; ret = MVHDP_ERR_NEGATIVE_COUNT; }
    }
    if (meta[META_MISROUTED]) { h->nslots_valid = false; FAIL(h, MVHDP_ERR_HIP, "internal: an entity could not be routed to a sweep kernel"); }
    // the batch counts as one observation of the walk search (mean kernel time per token)
    std::vector<unsigned long long> acc(ST_COUNT, 0);
    for (int i = 0; i < n; i++) for (int k = 0; k < ST_COUNT; k++) acc[k] += hs[(size_t)i * ST_COUNT + k];
    const bool comparable = p.fast && !h->tu.walk_fixed && !(flags & (MVHDP_SWEEP_FROZEN | MVHDP_SWEEP_EXACT_CHAIN));
    learn_from_sweep(h, p, acc.data(), meta + META_HIST, ms_sum, comparable);
    if (!p.frozen && acc[ST_ABORT] == 0) for (int m = 0; m < mm.M; m++) h->unassigned[m] = false;
    return ret;
}

// ---- tuning: what a host may pin, what the library has learnt (so that a document shard, a resumed chain or another handle on
// the same corpus does not search again) ----
extern "C" int mvhdp_get_tuning(mvhdp_handle h, mvhdp_tuning* t)
{
    CHECK_H(h);
    if (!t) FAIL(h, MVHDP_ERR_INVALID_ARG, "get_tuning: null");
    memset(t, 0, sizeof *t);
    t->force_primary = h->tu.force_primary; t->narrow = h->tu.narrow; t->walk_fixed = h->tu.walk_fixed;
    t->single_stream = h->tu.single_stream; t->primary_min_share = h->tu.primary_min_share; t->live16 = h->tu.live16;
    t->single_wave = h->tu.single_wave; t->live_overlap = h->tu.live_overlap; t->live_rows = h->tu.live_rows;
    for (int m = 0; m < MVHDP_MAX_MODALITIES; m++) { t->walk_theta[m] = h->tu.walk_theta[m]; t->tree_branch_share[m] = h->wt.walk_f[m]; }
    for (int g = 0; g < WALK_GROUPS; g++) t->learnt_walk_step[g] = g == h->wt.walk_cls ? h->wt.walk_i : h->wt.walk_i_by[g];
    t->learnt_walk_step[3] = -1;
    return MVHDP_OK;
}

extern "C" int mvhdp_set_tuning(mvhdp_handle h, const mvhdp_tuning* t)
{
    CHECK_H(h);
    if (!t) FAIL(h, MVHDP_ERR_INVALID_ARG, "set_tuning: null");
    const int fp = t->force_primary;
    if (!(fp == 0 || fp == 1 || fp == 2 || fp == 4 || fp == 8 || fp == 16 || fp == 32)) FAIL(h, MVHDP_ERR_INVALID_ARG, "set_tuning: force_primary must be 0, 1, 2, 4, 8, 16 or 32");
    if (t->primary_min_share < 0.0 || t->primary_min_share > 1.0) FAIL(h, MVHDP_ERR_INVALID_ARG, "set_tuning: primary_min_share outside [0, 1]");
    for (int g = 0; g < WALK_GROUPS; g++) if (t->learnt_walk_step[g] < -1 || t->learnt_walk_step[g] > MVHDP_WALK_BINS) FAIL(h, MVHDP_ERR_INVALID_ARG, "set_tuning: learnt_walk_step outside [-1, 20]");
    h->tu.force_primary = fp; h->tu.narrow = t->narrow; h->tu.walk_fixed = t->walk_fixed ? 1 : 0;
    h->tu.single_stream = t->single_stream ? 1 : 0;
    h->tu.live16 = t->live16 < 0 ? -1 : (t->live16 ? 1 : 0);
    h->tu.single_wave = t->single_wave ? 1 : 0;
    h->tu.live_overlap = t->live_overlap < 0 ? -1 : (t->live_overlap ? 1 : 0);
    h->tu.live_rows = t->live_rows < 0 ? -1 : (t->live_rows ? 1 : 0);
    h->tu.primary_min_share = t->primary_min_share > 0.0 ? t->primary_min_share : 0.10;
    for (int m = 0; m < MVHDP_MAX_MODALITIES; m++) h->tu.walk_theta[m] = t->walk_theta[m];
    if (t->learnt_walk_step[0] >= 0 || t->learnt_walk_step[1] >= 0 || t->learnt_walk_step[2] >= 0) h->wt.restore(t->learnt_walk_step, t->tree_branch_share, h->mm.M);
    return MVHDP_OK;
}

// The planner and the walk search without a device (tests/test_plan.py): pure functions of their arguments.
extern "C" int mvhdp_plan_probe(const mvhdp_plan_input* pi, const mvhdp_tuning* t, mvhdp_plan_output* po)
{
    if (!pi || !po) return MVHDP_ERR_INVALID_ARG;
    PlanIn in;
    in.K = pi->num_topics; in.M = pi->num_modalities; in.D = pi->num_entities; in.mdt = pi->max_entity_tokens;
    if (in.K < 1 || in.K > MVHDP_MAX_TOPICS || in.M < 1 || in.M > MVHDP_MAX_MODALITIES || in.D < 0) return MVHDP_ERR_INVALID_ARG;
    in.have_order = true;
    for (int c = 0; c < 5; c++) in.n_longer[c] = pi->entities_longer_than[c];
    for (int b = 0; b < MVHDP_HIST_BINS; b++) in.tok_hist[b] = pi->tokens_by_list_rounds[b];
    for (int b = 0; b < MVHDP_ENT_BINS; b++) in.ent_hist[b] = pi->entities_by_class[b];
    in.flags = pi->flags; in.debug = pi->debug != 0; in.batch = pi->batch != 0; in.trees_current = pi->trees_current != 0;
    in.first_inactive = pi->inactive_topics ? 0 : -1;
    in.num_cus = pi->num_cus > 0 ? pi->num_cus : 256;
    for (int c = 0; c < MVHDP_N_CLASSES; c++) for (int f = 0; f < 3; f++) in.regs.regs[c][f] = pi->kernel_registers[c][f];
    PlanTuning tu;
    WalkTuner wt;
    wt.init_defaults(in.K);
    if (t) {
        tu.force_primary = t->force_primary; tu.narrow = t->narrow; tu.walk_fixed = t->walk_fixed; tu.single_stream = t->single_stream;
        tu.live16 = t->live16; tu.single_wave = t->single_wave ? 1 : 0; tu.live_overlap = t->live_overlap; tu.live_rows = t->live_rows;
        if (t->primary_min_share > 0) tu.primary_min_share = t->primary_min_share;
        for (int m = 0; m < MVHDP_MAX_MODALITIES; m++) tu.walk_theta[m] = t->walk_theta[m];
        if (t->learnt_walk_step[0] >= 0 || t->learnt_walk_step[1] >= 0 || t->learnt_walk_step[2] >= 0) wt.restore(t->learnt_walk_step, t->tree_branch_share, in.M);
    }
    SweepPlan p;
    plan_sweep(in, tu, wt, p);
    memset(po, 0, sizeof *po);
    po->status = p.err;
    if (p.err) return MVHDP_OK;
    po->segments = p.nseg; po->primary_class = p.pc; po->routed_prefix = p.H; po->register_resident = p.fast ? 1 : 0;
    po->need_full_trees = p.need_full ? 1 : 0; po->dominant_class = p.dominant;
    for (int c = 0; c < MVHDP_N_CLASSES; c++) {
        po->class_used[c] = p.cls[c].used ? 1 : 0; po->class_map[c] = p.class_map[c];
        po->class_stream[c] = p.cls[c].stream; po->class_grid[c] = p.cls[c].grid; po->class_lds_bytes[c] = (int64_t)p.cls[c].lds;
        po->class_walk[c] = p.cls[c].walk; po->class_narrow[c] = p.cls[c].narrow; po->class_register_resident[c] = p.cls[c].fast ? 1 : 0;
        po->class_theta0[c] = p.cls[c].theta[0];
    }
    po->delta16 = p.delta16 ? 1 : 0;
    po->live_rows = p.live_rows ? 1 : 0;
    return MVHDP_OK;
}

extern "C" int mvhdp_tuner_probe(int32_t num_modalities, const double* tree_branch_share, const double* u1_hist, const double* ns_by_step /*[21]*/,
                                 int32_t n_sweeps, int32_t group, int32_t* steps_out /*[n_sweeps]*/)
{
    if (!tree_branch_share || !ns_by_step || !steps_out || num_modalities < 1 || num_modalities > MVHDP_MAX_MODALITIES || n_sweeps < 0) return MVHDP_ERR_INVALID_ARG;
    WalkTuner wt;
    if (group < 0 || group >= WALK_GROUPS) return MVHDP_ERR_INVALID_ARG;
    for (int m = 0; m < num_modalities; m++) wt.walk_f[m] = tree_branch_share[m];
    for (int b = 0; b < MVHDP_WALK_BINS; b++) wt.walk_hist[b] = u1_hist ? u1_hist[b] : 0.0;
    for (int i = 0; i < n_sweeps; i++) {
        double theta[WALK_GROUPS][MVHDP_MAXM]; bool measure;
        wt.propose(group, num_modalities, theta, &measure);
        steps_out[i] = wt.walk_probe_i;
        wt.observe(true, 1, num_modalities, ns_by_step[std::max(0, std::min(MVHDP_WALK_BINS, wt.walk_probe_i))]);
    }
    return MVHDP_OK;
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
    h->have_counts = true; h->have_trees = false; h->counts_stale = false;
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

// optimizeP PTM:2706-2792: acc[m*M+i] += pDistr_Mean[m][i][doc] over this handle's entities, one after the other in entity order
// (PTM:2789-2792).  A group of document shards in ONE process hands the accumulators from member to member (ascending doc_id_base): the
// additions are then the single handle's, in the same order, bit for bit.
int mvhdp_view_overlap_accumulate(mvhdp_ctx* h, double* acc)
{
    MvModel& mm = h->mm;
    int rc = require_corpus(h); if (rc) return rc;
    const int M = mm.M;
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
        double a = acc[i];
        const double* col = host.data() + (size_t)i * mm.D;
        for (int64_t doc = 0; doc < mm.D; doc++) a += col[doc];
        acc[i] = a;
    }
    return MVHDP_OK;
}

extern "C" int mvhdp_view_overlap_sums(mvhdp_handle h, double* sums)
{
    CHECK_H(h);
    if (!sums) FAIL(h, MVHDP_ERR_INVALID_ARG, "view_overlap_sums: null");
    for (int i = 0; i < h->mm.M * h->mm.M; i++) sums[i] = 0.0;
    return mvhdp_view_overlap_accumulate(h, sums);
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
                const bool has = h->h_present[m].empty() ? h->h_doc_off[m][d + 1] > h->h_doc_off[m][d] : h->h_present[m][(size_t)d] != 0;
                if (has) last = d;                             // (a present view without tokens is refreshed to zeros, PTM:2873-2886)
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

// optimizeDP's view-table simulation PTM:2454-2488 on the device (opt-in: mvhdp_stats.hip dp_tables_kernel).  The histogram comes from
// the host -- a single handle's mvhdp_get_doc_topic_hist or a sharded model's mvhdp_group_doc_topic_hist: the draw of a cell belongs to
// the WHOLE model's cell, so the statistic is the same however the entities are sharded.
extern "C" int mvhdp_dp_table_statistics(mvhdp_handle h, int32_t m, const int32_t* hist, int32_t hist_len, const double* conc, uint64_t seed, uint32_t round,
                                         double* mk, uint8_t* active)
{
    CHECK_H(h);
    MvModel& mm = h->mm;
    const int K = mm.K;
    if (m < 0 || m >= mm.M || !hist || hist_len < 1 || !conc || !mk || !active) FAIL(h, MVHDP_ERR_INVALID_ARG, "dp_table_statistics: bad argument");
    HIPC(h, hipSetDevice(h->device));
    const size_t hb = (size_t)K * hist_len * sizeof(int32_t);
    void* d = nullptr;
    hipError_t e = hipMalloc(&d, hb + (size_t)K * (2 * sizeof(double) + 8));
    if (e != hipSuccess) HIPC(h, e);
    int32_t* d_hist = (int32_t*)d;
    double* d_conc = (double*)((char*)d + ((hb + 7) & ~(size_t)7));
    double* d_mk = d_conc + K;
    uint8_t* d_act = (uint8_t*)(d_mk + K);
    e = hipMemcpyAsync(d_hist, hist, hb, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_conc, conc, (size_t)K * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = mvhdp_launch_dp_tables(d_hist, hist_len, K, m, d_conc, (uint32_t)seed, (uint32_t)(seed >> 32), round, d_mk, d_act, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(mk, d_mk, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(active, d_act, (size_t)K, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d);
    HIPC(h, e);
    return MVHDP_OK;
}

// n independent Antoniak draws (optimizeDP's root level PTM:2491-2517 with the device option): see antoniak_draws_kernel
extern "C" int mvhdp_antoniak_draws(mvhdp_handle h, int32_t n, const int32_t* items, const double* conc, uint64_t seed, uint32_t round, int32_t* tables)
{
    CHECK_H(h);
    if (n < 0 || (n > 0 && (!items || !conc || !tables))) FAIL(h, MVHDP_ERR_INVALID_ARG, "antoniak_draws: bad argument");
    if (n == 0) return MVHDP_OK;
    HIPC(h, hipSetDevice(h->device));
    void* d = nullptr;
    hipError_t e = hipMalloc(&d, (size_t)n * (sizeof(double) + 2 * sizeof(int32_t)));
    if (e != hipSuccess) HIPC(h, e);
    double* d_conc = (double*)d; int32_t* d_items = (int32_t*)(d_conc + n); int32_t* d_tab = d_items + n;
    e = hipMemcpyAsync(d_conc, conc, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_items, items, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = mvhdp_launch_antoniak_draws(n, d_items, d_conc, (uint32_t)seed, (uint32_t)(seed >> 32), round, d_tab, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(tables, d_tab, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d);
    HIPC(h, e);
    return MVHDP_OK;
}

// modelLogLikelihood PTM:3322-3452 in two parts, so that a group of document shards can put them together (mvhdp_group_log_likelihood):
// the DOCUMENT part of view m (PTM:3341-3367) belongs to the entities of a handle -- *ll and *cnt (modalityCnt) continue a sequential
// sum in entity order --, the MODEL part (PTM:3373-3441: the modalityCnt term, the topic-word term over n_wk, the n_k terms) to the
// replicated counts: once per model.
int mvhdp_ll_doc_accumulate(mvhdp_ctx* h, int m, double* ll, int64_t* cnt)
{
    MvModel& mm = h->mm;
    int rc = require_corpus(h); if (rc) return rc;
    if (!h->have_hyper) FAIL(h, MVHDP_ERR_STATE, "model_log_likelihood before set_hyper");
    if (m < 0 || m >= mm.M) FAIL(h, MVHDP_ERR_INVALID_ARG, "model_log_likelihood: bad view");
    if (mm.D == 0) return MVHDP_OK;
    HIPC(h, hipSetDevice(h->device));
    double* d_doc = nullptr;
    HIPC(h, hipMalloc(&d_doc, (size_t)mm.D * sizeof(double)));
    std::vector<double> hdoc((size_t)mm.D);
    hipError_t e = mvhdp_launch_loglik_doc(mm, m, d_doc, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hdoc.data(), d_doc, (size_t)mm.D * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d_doc);
    HIPC(h, e);
    double a = *ll;
    int64_t c = *cnt;
    for (int64_t d = 0; d < mm.D; d++) {
        const bool has = h->h_present[m].empty() ? h->h_doc_off[m][d + 1] > h->h_doc_off[m][d] : h->h_present[m][(size_t)d] != 0;
        if (has) { a += hdoc[d]; c++; }                                                              // PTM:3348-3367
    }
    *ll = a; *cnt = c;
    return MVHDP_OK;
}

int mvhdp_ll_model_finish(mvhdp_ctx* h, int m, double ll, int64_t modalityCnt, double* out)
{
    MvModel& mm = h->mm;
    if (!h->have_hyper || !h->have_counts) FAIL(h, MVHDP_ERR_STATE, "model_log_likelihood before set_hyper/build_counts");
    if (m < 0 || m >= mm.M || !out) FAIL(h, MVHDP_ERR_INVALID_ARG, "model_log_likelihood: bad view");
    const int M = mm.M, K = mm.K;
    ll += modalityCnt * log_gamma_stirling_host((double)mm.gamma[m] * mm.alpha_sum[m]);           // PTM:3373
    if (std::isnan(ll) || std::isinf(ll)) { *out = 0; return MVHDP_OK; }                           // PTM:3375-3383
    HIPC(h, hipSetDevice(h->device));
    const int NP = 1024;
    double* d_part = nullptr;
    unsigned long long* d_nz = nullptr;
    hipError_t e = hipMalloc(&d_part, NP * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&d_nz, sizeof(unsigned long long));
    std::vector<double> hpart(NP);
    std::vector<int32_t> nk((size_t)K);
    unsigned long long nz = 0;
    if (e == hipSuccess) e = mvhdp_launch_loglik_topic(mm, m, d_part, NP, d_nz, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hpart.data(), d_part, NP * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&nz, d_nz, sizeof nz, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(nk.data(), mm.counts + mm.rowbase[M] * K + (int64_t)m * K, (size_t)K * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (d_part) hipFree(d_part);
    if (d_nz) hipFree(d_nz);
    HIPC(h, e);
    for (int i = 0; i < NP; i++) ll += hpart[i];                                                    // PTM:3389-3415
    if (std::isnan(ll) || std::isinf(ll)) ll = 0;
    const double bv = mm.beta[m] * mm.V[m];
    for (int topic = 0; topic < K; topic++) {                                                       // PTM:3417-3435
        ll -= (bv + nk[topic]) == 0 ? 0 : log_gamma_stirling_host(bv + nk[topic]);
        if (std::isnan(ll) || std::isinf(ll)) ll = 0;
    }
    ll += bv == 0 ? 0 : log_gamma_stirling_host(bv) * K;                                            // PTM:3438
    ll -= mm.beta[m] == 0 ? 0 : log_gamma_stirling_host(mm.beta[m]) * (double)nz;                   // PTM:3441
    if (std::isinf(ll)) ll = 0;
    *out = ll;
    return MVHDP_OK;
}

extern "C" int mvhdp_model_log_likelihood(mvhdp_handle h, double* out)
{
    CHECK_H(h);
    if (!out) FAIL(h, MVHDP_ERR_INVALID_ARG, "model_log_likelihood: null");
    int rc = require_corpus(h); if (rc) return rc;
    if (!h->have_hyper || !h->have_counts) FAIL(h, MVHDP_ERR_STATE, "model_log_likelihood before set_hyper/build_counts");
    for (int m = 0; m < h->mm.M; m++) {
        double ll = 0;
        int64_t cnt = 0;
        rc = mvhdp_ll_doc_accumulate(h, m, &ll, &cnt); if (rc) return rc;
        rc = mvhdp_ll_model_finish(h, m, ll, cnt, &out[m]); if (rc) return rc;
    }
    return MVHDP_OK;
}
