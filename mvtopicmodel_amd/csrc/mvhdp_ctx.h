// mvhdp_ctx.h — the state behind an mvhdp_handle and the internal interfaces shared by the host-side sources of libmvhdp.so
// (mvhdp_api.hip: the C ABI of one handle; mvhdp_group.hip: document shards on several GPUs).  Not installed: include/mvhdp.h is the ABI.
#pragma once
#include "mvhdp_device.h"
#include "../../include/mvhdp.h"
#include "mvhdp_plan.h"

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
    uint8_t* d_present[MVHDP_MAXM]{};        // mvhdp_set_view_presence (nullptr: inferred from the spans)
    std::vector<uint8_t> h_present[MVHDP_MAXM];
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

    unsigned long long* d_ctl = nullptr;     // ONE block: [ST_COUNT] counters | activation key | META_WORDS64 | 8 work-queue heads (one reset launch, one read-back)
    unsigned long long* h_ctl = nullptr;     // pinned host copy of the first three parts (a sweep's read-back)
    unsigned long long* d_stats = nullptr;   //   = d_ctl
    long long* d_act_key = nullptr;          //   = d_ctl + ST_COUNT
    int32_t* d_births = nullptr;             // [2 + 2K] births of a live sweep (SweepLaunch::births)
    long long* d_birth_keys = nullptr;       // [K]
    std::vector<int32_t> h_births;           // host copies of the two (the list a segment starts with; what it ended with)
    std::vector<long long> h_birth_keys;
    unsigned long long* d_doc_counter = nullptr;
    int32_t* d_doc_order = nullptr;          // entities by decreasing token count (work-queue order)
    unsigned int* d_ovf_meta = nullptr;      // META_*: the next sweep's histograms (tokens by list size, entities by kernel class), per-class list lengths, misroutes
    int32_t* d_lists = nullptr;              // [MVHDP_N_CLASSES][D] entity lists written by route_kernel
    uint16_t* d_nslots = nullptr;            // [D] MvModel::nslots
    // live-rows form of a live sweep on the mirror: the heavy rows (listed by the segment's prepare pass) and the kernel that keeps their
    // stored trees current beside the samplers (mvhdp_kernels.hip heavy_refresh_kernel)
    int32_t* d_heavy_list = nullptr; unsigned int* d_heavy_ctl = nullptr;     // [HEAVY_CAP] rows; [0] rows listed, [1] stop
    hipStream_t rf_stream = nullptr; hipEvent_t ev_rf_go = nullptr, ev_rf_done = nullptr;
    int live_tree_every = 1;                 // diagnostics (MVHDP_LIVE_TREE_EVERY): tree rebuilds of a live sweep at every n-th segment border only
    int gate_pct = 60;                       // overlapped live segments: the next segment's trees and kernels are enqueued when this share of the current one's queue is taken
    bool delta16_used = false;               // MvModel::delta16 holds deltas of the last sweep (until the apply pass)
    int side_priority = 2;                   // side streams A and B at high priority (a hardware-queue pool of their own)
    hipStream_t side[PLAN_N_STREAMS]{};      // side streams of the wider kernel classes (created on first use; [0] unused: the handle's stream)
    hipEvent_t ev_fork = nullptr, ev_join[PLAN_N_STREAMS]{};
    std::vector<hipEvent_t> ev_many;         // mvhdp_sweep_many: two events per sweep of the batch
    unsigned long long* d_stats_many = nullptr;   // mvhdp_sweep_many: [n][ST_COUNT]
    int stats_many_cap = 0;
    std::vector<int64_t> tokens_desc;        // entity token counts, descending (the order of d_doc_order)
    int64_t* d_carry[MVHDP_MAXM]{};          // doc_topic_proportions: per view, the entity whose view-m counts score entity d (lazily built)
    // what the last sweep (or the recount after new assignments) left behind for the next plan
    unsigned long long last_hist[MVHDP_HIST_BINS]{};   // tokens by topic-list size class
    unsigned long long last_ent[MVHDP_ENT_BINS]{};     // entities by kernel class
    bool nslots_valid = false;               // MvModel::nslots and the two histograms describe the current assignments
    // Some token of view m may still carry UNASSIGNED_TOPIC (-1, PTM:63): set by set_corpus (which fills z with -1) and by set_assignments
    // when the host's array holds one, cleared when a full sweep has visited every entity without abandoning one.  A live sweep on the
    // 16-bit mirror needs every row's total to be constant (a LIGHT row can then never reach 65535 in a cell); a first visit of an
    // unassigned token only adds to its row, so while this is set live sweeps stay on the 32-bit table.
    bool unassigned[MVHDP_MAXM]{};
    bool counts_stale = false;               // assignments were replaced (set_assignments / init_from_trees) and the counts not rebuilt since
    // overlapped segments (MVHDP_SWEEP_SEGMENT_OVERLAP, live sweeps with live_overlap): the second copy of what two segments in flight
    // must not share, allocated on first use
    struct Overlap {
        int32_t* counts2 = nullptr; uint16_t* counts16_2 = nullptr; double* dtab2 = nullptr; double* root2 = nullptr; double* trees2 = nullptr;
        int32_t* delta2 = nullptr; int32_t* delta3 = nullptr;
        unsigned long long* ctl2 = nullptr;  // [8] queue heads, then [8 x u32] class list lengths
        int32_t* lists2 = nullptr;
        hipStream_t x1 = nullptr, xa = nullptr;
        hipEvent_t ev_start = nullptr;
        bool deltas_dirty = false;           // an overlapped segmented sweep was enqueued and has not been seen to finish: delta2 / delta3 may hold leftovers
        std::vector<hipEvent_t> ev_seg;      // per segment: [3 * s] kernels done, [3 * s + 1] update done, [3 * s + 2] queue heads reset
    } ov;
    PlanRegs regs{};                         // register counts of the compiled kernels (occupancy)
    PlanTuning tu;                           // what the host pinned (mvhdp_set_tuning; environment read once at create)
    WalkTuner wt;                            // the walk-threshold search
    bool dbg_env = false;                    // MVHDP_DEBUG was set at create
    size_t lds_attr_set = 0;
};


bool mvhdp_is_live(mvhdp_ctx* h);
#define CHECK_H(h) do { if (!(h) || !mvhdp_is_live(h)) return MVHDP_ERR_INVALID_ARG; \
                        if ((h)->device_released) return MVHDP_ERR_STATE; /* the process is exiting */ } while (0)
#define HIPC(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    (h)->err = std::string(#call) + ": " + hipGetErrorString(e_); return MVHDP_ERR_HIP; } } while (0)
#define FAIL(h, code, msg) do { (h)->err = (msg); return (code); } while (0)

enum { MVHDP_HEAVY_CAP = 8192 };             // heavy rows a live sweep keeps trees for (each holds 65535 tokens or more: beyond any corpus that fits a GPU)
enum { MVHDP_TAIL_WORDS = 4 };               // int32 words allocated behind counts_len() in the counts and delta buffers ([0]: a group's status word)
static int64_t counts_len(const mvhdp_ctx* h) { return h->mm.rowbase[h->mm.M] * h->mm.K + (int64_t)h->mm.M * h->mm.K; }

// d_ovf_meta (u64 words unless said otherwise): [META_HIST .. +MVHDP_HIST_BINS+MVHDP_ENT_BINS) what the sweep kernels leave for the next
// plan (SweepLaunch::slot_hist); u32 words [META_CLASS_COUNTS .. +MVHDP_N_CLASSES) lengths of the route pass's class lists;
// [META_MISROUTED] entities the route pass could not place
enum { META_HIST = 0, META_MISROUTED = 40, META_WORDS64 = 48, META_BYTES = META_WORDS64 * 8, META_CLASS_COUNTS = 64 /* u32 index = byte 256 */ };
static_assert(MVHDP_HIST_BINS + MVHDP_ENT_BINS <= 32, "histograms overlap the class counts");
enum { CTL_WORDS = ST_COUNT + 1 + META_WORDS64 + 8 };


// device-side buffers of the parity tests' debug outputs
struct DebugBufs {
    std::vector<void*> to_free;
    double* tok_dbg[MVHDP_MAXM] = {};
    int n_trace = 0;
    const int64_t* trace_doc = nullptr; const int32_t* trace_view = nullptr; const int32_t* trace_pos = nullptr;
    double* trace_out = nullptr;
    void release() { for (void* p : to_free) hipFree(p); to_free.clear(); }
};

struct SweepOutcome {                        // what enqueue_sweep learnt on the way (segment-border activations need the host)
    int n_activations = 0;
    long long first_act = LLONG_MAX;
};

// A sweep between its two halves: mvhdp_sweep_begin plans it and puts everything on the device without waiting (kernels, the
// read-back of the counters into the handle's pinned buffer); mvhdp_sweep_finish waits, applies (unless NO_APPLY / FROZEN / live),
// reports and learns.  mvhdp_sweep = begin + finish; a group begins the sweeps of all its members before it finishes any.
struct PendingSweep {
    SweepPlan p;
    SweepOutcome oc;
    uint32_t flags = 0;
    bool debug = false;
    const mvhdp_debug* dbg = nullptr;
    DebugBufs db;
    bool open = false;
    bool births = false;                     // a live sweep whose topics are born chunk by chunk (SweepLaunch::births): the last segment's are applied by the finish
};
int mvhdp_sweep_begin(mvhdp_ctx* h, uint32_t sweep_idx, uint64_t seed, uint32_t flags, const double* p_override, const mvhdp_debug* dbg, PendingSweep& ps);
int mvhdp_sweep_finish(mvhdp_ctx* h, PendingSweep& ps, mvhdp_sweep_stats* stats);
// pieces of the statistics either side of the sweep that a group of document shards composes (mvhdp_api.hip; see there)
int mvhdp_view_overlap_accumulate(mvhdp_ctx* h, double* acc /*[M*M], continued*/);
int mvhdp_ll_doc_accumulate(mvhdp_ctx* h, int m, double* ll, int64_t* cnt);
int mvhdp_ll_model_finish(mvhdp_ctx* h, int m, double ll_doc, int64_t modalityCnt, double* out);
