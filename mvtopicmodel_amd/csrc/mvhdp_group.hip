// mvhdp_group.hip — document shards on several GPUs inside the library (include/mvhdp.h "document shards").
//
// Replaces, for a host in ANY language, what the reference keeps inside its own process: the nst x nut queue mesh between
// sampler and updater threads and the barrier at the end of an iteration (PTM:1042-1049, PTM:1232).  Here the samplers are
// GPUs: every member handle holds a contiguous range of entities and a full replica of n_wk / n_k; one sweep of the group =
//   every member samples its entities against the same snapshot (MVHDP_SWEEP_NO_APPLY)            WRK:186-233 x members
//   the int32 deltas of all members are summed: device-side over members that share a GPU, RCCL all-reduce over xGMI between GPUs
//   every replica applies the same sum row range by row range, rebuilding the F+trees of a range while the next is on the wire
//   (when inActiveTopicIndex is non-empty) the activation key is MIN-reduced so that every replica activates the same topic UPD:263-270
// Two ways to form a group: one process driving all its GPUs (mvhdp_group_create: ncclCommInitAll over the members' devices), or one
// process per GPU (mvhdp_group_create_rank: ncclCommInitRank with an id the host's launcher passes around).
//
// RCCL is opened at run time (dlopen by soname; MVHDP_RCCL_LIB names another file): libmvhdp.so itself does not need it, a
// single-GPU host never loads it, and a process that already has a copy mapped (PyTorch) shares that copy.
#include "mvhdp_ctx.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <map>

namespace {

struct Rccl {
    void* lib = nullptr;
    std::string path, why;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

std::mutex g_rccl_mutex;
Rccl g_rccl;

// opens RCCL once per process; returns nullptr (and says why in g_rccl.why) when it cannot
Rccl* rccl()
{
    std::lock_guard<std::mutex> lk(g_rccl_mutex);
    if (g_rccl.lib) return &g_rccl;
    if (!g_rccl.why.empty()) return nullptr;
    // Order: a copy that is ALREADY mapped into the process (PyTorch's: same soname, RTLD_NOLOAD maps nothing new), then the file the
    // host names (MVHDP_RCCL_LIB: mvtopicmodel_amd/_lib.py points it at the torch wheel's copy so that a later `import torch` shares
    // it), then the installation's own.  (The plain soname first would always succeed through this library's RUNPATH and the
    // override would never be looked at.)
    // MVHDP_RCCL_LIB_FIRST=1 (tests, diagnostics): the named file even when a copy is mapped -- a host process that has imported PyTorch
    // (bench.py) holds the real RCCL, which refuses two ranks on one device; the rank tests name their stand-in this way.
    std::string errs;
    const char* named = getenv("MVHDP_RCCL_LIB");
    const char* first = getenv("MVHDP_RCCL_LIB_FIRST");
    const bool named_first = named && first && atoi(first) != 0;
    if (!named_first)
        if (void* l = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD)) { g_rccl.lib = l; g_rccl.path = "librccl.so.1 (already mapped)"; }
    std::vector<std::string> names;
    if (const char* e = getenv("MVHDP_RCCL_LIB")) names.push_back(e);
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    names.push_back("/opt/rocm/lib/librccl.so.1");
    for (const std::string& n : names) {
        if (g_rccl.lib) break;
        void* l = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!l) { errs += std::string(dlerror() ? dlerror() : "?") + "; "; continue; }
        g_rccl.lib = l; g_rccl.path = n;
    }
    if (!g_rccl.lib) { g_rccl.why = "RCCL could not be opened: " + errs; return nullptr; }
#define SYM(field, name) do { *(void**)(&g_rccl.field) = dlsym(g_rccl.lib, name); \
        if (!g_rccl.field) { g_rccl.why = std::string("RCCL lacks ") + name; dlclose(g_rccl.lib); g_rccl.lib = nullptr; return nullptr; } } while (0)
    SYM(GetVersion, "ncclGetVersion"); SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommInitAll, "ncclCommInitAll"); SYM(CommDestroy, "ncclCommDestroy"); SYM(AllReduce, "ncclAllReduce");
    SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    return &g_rccl;
}

__global__ __launch_bounds__(256) void add_into_kernel(int32_t* __restrict__ dst, const int32_t* __restrict__ src, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] += src[i];
}

hipError_t launch_add_into(int32_t* dst, const int32_t* src, int64_t n, hipStream_t s)
{
    int grid = (int)std::min<int64_t>((n + 255) / 256, 8192);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(add_into_kernel, dim3(grid), dim3(256), 0, s, dst, src, n);
    return hipGetLastError();
}

__global__ void set_word_kernel(int32_t* p, int32_t v) { *p = v; }

// ---- the exchange at half width ----
// A row whose type holds at most 32767 tokens in the whole corpus (MvModel::heavy == 0: decided from the replicated counts whenever the
// row's tree is built) cannot collect a delta beyond +-32767 in one sweep, summed over ALL ranks: a delta of a cell is bounded by the
// tokens of the type that moved.  Two such deltas travel in one 32-bit word, a + 65536 b as a plain integer: the sum of the words over
// the ranks is sum(a) + 65536 sum(b) (no wrap: both sums stay within 16 bits), from which sum(a) is the sign-extended low half and sum(b)
// what is left.  The other rows (0.4 % of them at C4, half the tokens) travel as they are.  woff[r]: first word of row r in the packed
// buffer (prefix over the rows of (K + 1) / 2 or K words), the same on every rank because the row classes are.
__global__ __launch_bounds__(256) void pack_rows_kernel(const int32_t* __restrict__ delta, const uint8_t* __restrict__ heavy, const int64_t* __restrict__ woff,
                                                        int32_t* __restrict__ xp, int64_t r0, int64_t r1, int K)
{
    const int lane = threadIdx.x & 63;
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t row = r0 + (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); row < r1; row += wstride) {
        const int32_t* d = delta + row * K;
        int32_t* o = xp + woff[row];
        if (heavy[row] == 0) {
            const int nw = (K + 1) >> 1;
            for (int j = lane; j < nw; j += 64) { const int a = d[2 * j], b = (2 * j + 1 < K) ? d[2 * j + 1] : 0; o[j] = a + b * 65536; }
        } else for (int k = lane; k < K; k += 64) o[k] = d[k];
    }
}

__global__ __launch_bounds__(256) void unpack_rows_kernel(int32_t* __restrict__ delta, const uint8_t* __restrict__ heavy, const int64_t* __restrict__ woff,
                                                          const int32_t* __restrict__ xp, int64_t r0, int64_t r1, int K)
{
    const int lane = threadIdx.x & 63;
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t row = r0 + (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); row < r1; row += wstride) {
        int32_t* d = delta + row * K;
        const int32_t* o = xp + woff[row];
        if (heavy[row] == 0) {
            const int nw = (K + 1) >> 1;
            for (int j = lane; j < nw; j += 64) {
                const int w = o[j];
                const int a = (int)(short)(w & 0xffff);
                d[2 * j] = a;
                if (2 * j + 1 < K) d[2 * j + 1] = (w - a) >> 16;
            }
        } else for (int k = lane; k < K; k += 64) d[k] = o[k];
    }
}

// counts += sum - own  (the other shards' share of an exchanged delta buffer)
__global__ __launch_bounds__(256) void add_remote_kernel(int32_t* __restrict__ counts, const int32_t* __restrict__ sum, const int32_t* __restrict__ own, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { const int d = sum[i] - own[i]; if (d) counts[i] += d; }
}

// the caller's current device, put back when a group call returns (a group call visits the devices of all its members)
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
};

}  // namespace

enum { XSCRATCH_BYTES = 1 << 20 };            // device scratch of the cross-rank statistics (one process per GPU)

struct mvhdp_group_ctx {
    std::vector<mvhdp_ctx*> members;          // local members, in the caller's order
    std::vector<int> leader_of;               // per member: index of the first member on the same device (its delta buffer takes the device's sum)
    std::vector<int> leaders;                 // member indices that are leaders, one per distinct device: the RCCL ranks of this process
    std::vector<ncclComm_t> comms;            // one communicator per leader (empty: no RCCL, a single device)
    std::vector<hipEvent_t> ev_swept;         // per member: its sweep (or count build) is on its stream
    std::vector<hipEvent_t> ev_reduced;       // per leader: the reduced buffer is complete
    std::vector<long long*> d_key;            // per leader: 8 bytes for the MIN-reduce of the activation key
    hipEvent_t ev_x0 = nullptr, ev_x1 = nullptr;   // around the exchange on leader 0's stream
    int nranks = 1, rank0 = 0;                // ranks over all processes, rank of this process's first leader
    bool multi_process = false;
    bool rccl_used = false;
    int rccl_version = 0;
    int chunks = 4;                           // row ranges of one exchange: the apply + tree rebuild of range i runs while range i+1 is on the wire
    double last_exchange_ms = 0.0;
    long long sweeps = 0;
    // MVHDP_SWEEP_ASYNC_EXCHANGE (live sweeps): the all-reduce of sweep t's deltas runs beside sweep t+1 on a stream of its own
    std::vector<int32_t*> xbuf, sbuf;         // per member: the deltas on the wire (summed in place) and this member's own share of them
    std::vector<hipStream_t> comm;            // per leader: the stream of the collective
    std::vector<hipEvent_t> ev_xfer;          // per leader: the sum has arrived (and been handed to the co-located members)
    bool async_pending = false;               // an exchange is in flight: the replicas lack each other's last sweep
    bool abort_raised = false;                // mvhdp_group_abort: the next sweep of this rank contributes nothing and fails on every rank
    std::vector<int> by_entity;               // member indices by ascending doc_id_base: the order of the "in entity order" statistics
    // the exchange at half width (pack_rows_kernel): the layout is fixed by the first completed sweep behind a (re)count -- from then on every
    // token is assigned and a row's total, hence its class, cannot move -- and confirmed by all ranks in the sweep after it
    bool pack_on = false;                      // the exchange travels packed (agreed on by every rank, see group_step)
    bool pack_ready = false;                   // this process has computed the layout and proposes it
    bool pack_allowed = true;                  // MVHDP_EXCHANGE16=0 switches it off (diagnostics)
    int32_t pack_hash = 0;                     // 1 .. 4092: what this process proposes (the layout's word count, folded)
    std::vector<int64_t> woff;                 // [rows + 1] first packed word of every row
    std::vector<int64_t*> d_woff;              // per leader: the same on the device
    std::vector<uint8_t*> d_cls;               // per leader: the row classes the layout was made from (MvModel::heavy of that moment)
    std::vector<int32_t*> xpack;               // per leader: the packed buffer
    long long sweeps_since_counts = 0;
    long long last_exchange_bytes = 0;         // bytes this rank handed to the collective in the last sweep
    void* d_scratch = nullptr;                // one process per GPU: device buffer of the cross-rank statistics (allocated with the group: no allocation stands between a rank and a collective)
    std::string err;
};

static std::mutex g_group_mutex;
static std::set<mvhdp_group_ctx*>* g_groups = nullptr;
static thread_local std::string g_group_create_error;

static bool group_live(mvhdp_group_ctx* g)
{
    std::lock_guard<std::mutex> lk(g_group_mutex);
    return g_groups && g_groups->count(g) != 0;
}

// a member destroyed before its group (the header asks for the other order) is noticed, not dereferenced
static bool members_live(mvhdp_group_ctx* g)
{
    for (mvhdp_ctx* h : g->members) if (!mvhdp_is_live(h) || h->device_released) return false;
    return true;
}
#define CHECK_G(g) do { if (!(g) || !group_live(g)) return MVHDP_ERR_INVALID_ARG; \
                        if (!members_live(g)) { (g)->err = "a member handle of the group has been destroyed"; return MVHDP_ERR_STATE; } } while (0)
#define GFAIL(g, code, msg) do { (g)->err = (msg); return (code); } while (0)
#define GHIP(g, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (g)->err = std::string(#call) + ": " + hipGetErrorString(e_); return MVHDP_ERR_HIP; } } while (0)
#define GNCCL(g, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { \
    (g)->err = std::string(#call) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "RCCL error"); return MVHDP_ERR_HIP; } } while (0)
#define GMEM(g, i, call) do { int rc_ = (call); if (rc_ != MVHDP_OK) { \
    (g)->err = "member " + std::to_string(i) + ": " + (g)->members[i]->err; return rc_; } } while (0)

extern "C" const char* mvhdp_group_last_error(mvhdp_group g)
{
    return (g && group_live(g)) ? g->err.c_str() : g_group_create_error.c_str();
}

static int64_t counts_len_of(const mvhdp_ctx* h) { return h->mm.rowbase[h->mm.M] * h->mm.K + (int64_t)h->mm.M * h->mm.K; }

static int validate_members(int32_t n, const mvhdp_handle* members)
{
    if (n < 1 || !members) { g_group_create_error = "group_create: no members"; return MVHDP_ERR_INVALID_ARG; }
    for (int i = 0; i < n; i++) {
        mvhdp_ctx* h = members[i];
        if (!h || !mvhdp_is_live(h) || h->device_released) { g_group_create_error = "group_create: member " + std::to_string(i) + " is not a live handle"; return MVHDP_ERR_INVALID_ARG; }
        for (int j = 0; j < i; j++) if (members[j] == h) { g_group_create_error = "group_create: a handle is listed twice"; return MVHDP_ERR_INVALID_ARG; }
        const MvModel &a = members[0]->mm, &b = h->mm;
        bool same = a.K == b.K && a.M == b.M;
        for (int m = 0; same && m < a.M; m++) same = a.V[m] == b.V[m];
        if (!same) { g_group_create_error = "group_create: every member must hold the same model shape (topics, views, alphabet sizes)"; return MVHDP_ERR_INVALID_ARG; }
    }
    return MVHDP_OK;
}

static int group_common_init(mvhdp_group_ctx* g, int32_t n, const mvhdp_handle* members)
{
    if (const char* e = getenv("MVHDP_EXCHANGE16")) g->pack_allowed = atoi(e) != 0;      // (every rank of a group must say the same)
    std::map<int, int> first_on_device;
    for (int i = 0; i < n; i++) {
        mvhdp_ctx* h = members[i];
        g->members.push_back(h);
        auto it = first_on_device.find(h->device);
        if (it == first_on_device.end()) { first_on_device[h->device] = i; g->leader_of.push_back(i); g->leaders.push_back(i); }
        else g->leader_of.push_back(it->second);
    }
    for (int i = 0; i < n; i++) g->by_entity.push_back(i);
    std::stable_sort(g->by_entity.begin(), g->by_entity.end(), [&](int a, int b) { return g->members[a]->mm.doc_id_base < g->members[b]->mm.doc_id_base; });
    for (int i = 0; i < n; i++) {
        hipEvent_t ev;
        GHIP(g, hipSetDevice(g->members[i]->device));
        GHIP(g, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        g->ev_swept.push_back(ev);
    }
    for (size_t l = 0; l < g->leaders.size(); l++) {
        hipEvent_t ev; long long* k = nullptr;
        GHIP(g, hipSetDevice(g->members[g->leaders[l]]->device));
        GHIP(g, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        g->ev_reduced.push_back(ev);
        GHIP(g, hipMalloc(&k, sizeof(long long)));
        g->d_key.push_back(k);
    }
    GHIP(g, hipSetDevice(g->members[g->leaders[0]]->device));
    GHIP(g, hipEventCreate(&g->ev_x0));
    GHIP(g, hipEventCreate(&g->ev_x1));
    return MVHDP_OK;
}

static void group_register(mvhdp_group_ctx* g)
{
    std::lock_guard<std::mutex> lk(g_group_mutex);
    if (!g_groups) g_groups = new std::set<mvhdp_group_ctx*>();
    g_groups->insert(g);
}

static void group_release(mvhdp_group_ctx* g)
{
    Rccl* r = g_rccl.lib ? &g_rccl : nullptr;
    for (size_t l = 0; l < g->comms.size(); l++) if (g->comms[l] && r) r->CommDestroy(g->comms[l]);
    g->comms.clear();
    for (size_t i = 0; i < g->ev_swept.size(); i++) if (g->ev_swept[i]) hipEventDestroy(g->ev_swept[i]);
    for (size_t l = 0; l < g->ev_reduced.size(); l++) if (g->ev_reduced[l]) hipEventDestroy(g->ev_reduced[l]);
    for (size_t l = 0; l < g->d_key.size(); l++) if (g->d_key[l]) hipFree(g->d_key[l]);
    for (int32_t* b : g->xbuf) if (b) hipFree(b);
    for (int32_t* b : g->sbuf) if (b) hipFree(b);
    for (hipStream_t st : g->comm) if (st) hipStreamDestroy(st);
    for (hipEvent_t ev : g->ev_xfer) if (ev) hipEventDestroy(ev);
    g->xbuf.clear(); g->sbuf.clear(); g->comm.clear(); g->ev_xfer.clear();
    if (g->d_scratch) hipFree(g->d_scratch);
    for (int64_t* q : g->d_woff) if (q) hipFree(q);
    for (uint8_t* q : g->d_cls) if (q) hipFree(q);
    for (int32_t* q : g->xpack) if (q) hipFree(q);
    if (g->ev_x0) hipEventDestroy(g->ev_x0);
    if (g->ev_x1) hipEventDestroy(g->ev_x1);
}

// One process, n handles: ncclCommInitAll over the distinct devices of the members.  Members that share a device (a test
// arrangement: several document shards on one GPU) are summed on that device and enter the collective as one rank.
extern "C" int mvhdp_group_create(int32_t n, const mvhdp_handle* members, mvhdp_group* out)
{
    if (!out) { g_group_create_error = "group_create: null"; return MVHDP_ERR_INVALID_ARG; }
    *out = nullptr;
    int rc = validate_members(n, members); if (rc) return rc;
    DeviceGuard dg;
    mvhdp_group_ctx* g = new mvhdp_group_ctx();
    rc = group_common_init(g, n, members);
    if (rc) { g_group_create_error = g->err; group_release(g); delete g; return rc; }
    g->nranks = (int)g->leaders.size(); g->rank0 = 0; g->multi_process = false;
    Rccl* r = rccl();
    if (!r) {
        if (g->leaders.size() > 1) { g_group_create_error = "group_create: " + g_rccl.why; group_release(g); delete g; return MVHDP_ERR_UNSUPPORTED; }
        // one device and no RCCL in this installation: the exchange is the device-side sum alone (mvhdp_group_info says so)
    } else {
        std::vector<int> devs;
        for (int l : g->leaders) devs.push_back(g->members[l]->device);
        g->comms.assign(devs.size(), nullptr);
        ncclResult_t nr = r->CommInitAll(g->comms.data(), (int)devs.size(), devs.data());
        if (nr != ncclSuccess) {
            g_group_create_error = std::string("ncclCommInitAll: ") + r->GetErrorString(nr);
            g->comms.clear(); group_release(g); delete g; return MVHDP_ERR_HIP;
        }
        g->rccl_used = true;
        r->GetVersion(&g->rccl_version);
    }
    group_register(g);
    *out = g;
    return MVHDP_OK;
}

extern "C" int mvhdp_group_unique_id(uint8_t* id)
{
    if (!id) { g_group_create_error = "group_unique_id: null"; return MVHDP_ERR_INVALID_ARG; }
    Rccl* r = rccl();
    if (!r) { g_group_create_error = "group_unique_id: " + g_rccl.why; return MVHDP_ERR_UNSUPPORTED; }
    static_assert(MVHDP_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId u;
    ncclResult_t nr = r->GetUniqueId(&u);
    if (nr != ncclSuccess) { g_group_create_error = std::string("ncclGetUniqueId: ") + r->GetErrorString(nr); return MVHDP_ERR_HIP; }
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return MVHDP_OK;
}

// One process per GPU: this process's handle is rank `rank` of `nranks`; `id` comes from mvhdp_group_unique_id on one rank and
// reaches the others through the host's launcher (a file, a socket, MPI, torch.distributed's store: 128 bytes).  Collective: every
// rank must call it.
extern "C" int mvhdp_group_create_rank(mvhdp_handle member, const uint8_t* id, int32_t rank, int32_t nranks, mvhdp_group* out)
{
    if (!out || !id) { g_group_create_error = "group_create_rank: null"; return MVHDP_ERR_INVALID_ARG; }
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) { g_group_create_error = "group_create_rank: bad rank"; return MVHDP_ERR_INVALID_ARG; }
    int rc = validate_members(1, &member); if (rc) return rc;
    Rccl* r = rccl();
    if (!r) { g_group_create_error = "group_create_rank: " + g_rccl.why; return MVHDP_ERR_UNSUPPORTED; }
    DeviceGuard dg;
    mvhdp_group_ctx* g = new mvhdp_group_ctx();
    rc = group_common_init(g, 1, &member);
    if (rc) { g_group_create_error = g->err; group_release(g); delete g; return rc; }
    g->nranks = nranks; g->rank0 = rank; g->multi_process = true;
    if (hipSetDevice(member->device) != hipSuccess || hipMalloc(&g->d_scratch, XSCRATCH_BYTES) != hipSuccess) {
        g_group_create_error = "group_create_rank: the scratch buffer of the cross-rank statistics could not be allocated"; group_release(g); delete g; return MVHDP_ERR_HIP;
    }
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    g->comms.assign(1, nullptr);
    hipSetDevice(member->device);
    ncclResult_t nr = r->CommInitRank(&g->comms[0], nranks, u, rank);
    if (nr != ncclSuccess) {
        g_group_create_error = std::string("ncclCommInitRank: ") + r->GetErrorString(nr);
        g->comms.clear(); group_release(g); delete g; return MVHDP_ERR_HIP;
    }
    g->rccl_used = true;
    r->GetVersion(&g->rccl_version);
    group_register(g);
    *out = g;
    return MVHDP_OK;
}

extern "C" int mvhdp_group_destroy(mvhdp_group g)
{
    if (!g) return MVHDP_OK;
    {
        std::lock_guard<std::mutex> lk(g_group_mutex);
        if (!g_groups || g_groups->erase(g) == 0) return MVHDP_ERR_INVALID_ARG;
    }
    DeviceGuard dg;
    for (mvhdp_ctx* h : g->members) if (mvhdp_is_live(h) && !h->device_released) { hipSetDevice(h->device); hipStreamSynchronize(h->stream); }
    group_release(g);
    delete g;
    return MVHDP_OK;
}

extern "C" int mvhdp_group_get_info(mvhdp_group g, mvhdp_group_info* info)
{
    CHECK_G(g);
    if (!info) GFAIL(g, MVHDP_ERR_INVALID_ARG, "group_get_info: null");
    memset(info, 0, sizeof *info);
    info->local_members = (int32_t)g->members.size();
    info->local_devices = (int32_t)g->leaders.size();
    info->ranks = g->nranks; info->first_rank = g->rank0;
    info->rccl = g->rccl_used ? 1 : 0; info->rccl_version = g->rccl_version;
    info->exchange_chunks = g->chunks;
    info->last_exchange_ms = g->last_exchange_ms;
    info->last_exchange_bytes = g->last_exchange_bytes;
    info->exchange_packed = g->pack_on ? 1 : 0;
    return MVHDP_OK;
}

extern "C" int mvhdp_group_set_exchange_chunks(mvhdp_group g, int32_t chunks)
{
    CHECK_G(g);
    if (chunks < 1 || chunks > 64) GFAIL(g, MVHDP_ERR_INVALID_ARG, "group_set_exchange_chunks: 1..64");
    g->chunks = chunks;
    return MVHDP_OK;
}

// ---- the exchange: buffer `which` of every member becomes the sum over ALL members of all ranks ----
// Stream order throughout, no host wait: co-located members are added into their device's leader, the leaders all-reduce the
// element range [e0, e1) (RCCL, in place, on the leader's stream), and ev_reduced[l] marks the range complete on leader l.
// None of the three returns before it has issued everything it was asked to issue: a local HIP failure is remembered (first error
// wins, g->err says what) and the calls go on -- a rank that stops half way through an exchange leaves its peers inside a collective.
struct XErr {
    mvhdp_group_ctx* g;
    int rc = MVHDP_OK;
    void hip(hipError_t e, const char* what) { if (e != hipSuccess && rc == MVHDP_OK) { rc = MVHDP_ERR_HIP; g->err = std::string(what) + ": " + hipGetErrorString(e); } }
    void nccl(ncclResult_t r, const char* what) { if (r != ncclSuccess && rc == MVHDP_OK) { rc = MVHDP_ERR_HIP; g->err = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error"); } }
};

static int reduce_local(mvhdp_group_ctx* g, bool counts)
{
    // members that share a device: delta_leader += delta_member, in stream order behind both sweeps
    XErr x{g};
    const int n = (int)g->members.size();
    for (int i = 0; i < n; i++) {
        const int li = g->leader_of[i];
        if (li == i) continue;
        mvhdp_ctx *h = g->members[i], *L = g->members[li];
        x.hip(hipSetDevice(L->device), "hipSetDevice");
        x.hip(hipStreamWaitEvent(L->stream, g->ev_swept[i], 0), "hipStreamWaitEvent");
        x.hip(launch_add_into(counts ? L->mm.counts : L->mm.delta, counts ? h->mm.counts : h->mm.delta, counts_len_of(L), L->stream), "add_into_kernel");
    }
    return x.rc;
}

static int allreduce_range(mvhdp_group_ctx* g, bool counts, int64_t e0, int64_t e1)
{
    if (e1 <= e0 || g->comms.empty()) return MVHDP_OK;
    Rccl* r = &g_rccl;
    XErr x{g};
    if (g->comms.size() > 1) x.nccl(r->GroupStart(), "ncclGroupStart");
    for (size_t l = 0; l < g->leaders.size(); l++) {
        mvhdp_ctx* L = g->members[g->leaders[l]];
        x.hip(hipSetDevice(L->device), "hipSetDevice");
        int32_t* buf = (counts ? L->mm.counts : L->mm.delta) + e0;
        x.nccl(r->AllReduce(buf, buf, (size_t)(e1 - e0), ncclInt32, ncclSum, g->comms[l], L->stream), "ncclAllReduce");
    }
    if (g->comms.size() > 1) x.nccl(r->GroupEnd(), "ncclGroupEnd");
    return x.rc;
}

// the co-located members of a device take the reduced range from their leader
static int fan_out_range(mvhdp_group_ctx* g, bool counts, int64_t e0, int64_t e1)
{
    if (e1 <= e0 || g->members.size() == g->leaders.size()) return MVHDP_OK;
    XErr x{g};
    for (size_t l = 0; l < g->leaders.size(); l++) {
        mvhdp_ctx* L = g->members[g->leaders[l]];
        x.hip(hipSetDevice(L->device), "hipSetDevice");
        x.hip(hipEventRecord(g->ev_reduced[l], L->stream), "hipEventRecord");
        for (size_t i = 0; i < g->members.size(); i++) {
            if (g->leader_of[i] != g->leaders[l] || (int)i == g->leaders[l]) continue;
            mvhdp_ctx* h = g->members[i];
            x.hip(hipStreamWaitEvent(h->stream, g->ev_reduced[l], 0), "hipStreamWaitEvent");
            const int32_t* src = (counts ? L->mm.counts : L->mm.delta) + e0;
            int32_t* dst = (counts ? h->mm.counts : h->mm.delta) + e0;
            x.hip(hipMemcpyAsync(dst, src, (size_t)(e1 - e0) * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream), "hipMemcpyAsync");
            // the leader's own update of this range zeroes its delta rows: not before this copy has read them
            x.hip(hipEventRecord(g->ev_swept[i], h->stream), "hipEventRecord");
            x.hip(hipStreamWaitEvent(L->stream, g->ev_swept[i], 0), "hipStreamWaitEvent");
        }
    }
    return x.rc;
}

// The layout of the packed exchange, from the row classes as they stand (called behind a completed sweep of fully assigned tokens: a
// row's total cannot move any more).  Local work only; what it yields is PROPOSED to the other ranks by the next sweep's status words.
static int pack_prepare(mvhdp_group_ctx* g)
{
    if (g->pack_ready || !g->pack_allowed) return MVHDP_OK;
    for (mvhdp_ctx* h : g->members) for (int m = 0; m < h->mm.M; m++) if (h->unassigned[m]) return MVHDP_OK;   // (a first visit only adds to its row: not yet)
    mvhdp_ctx* L0 = g->members[g->leaders[0]];
    const int64_t rows = L0->mm.rowbase[L0->mm.M];
    const int K = L0->mm.K;
    if (rows <= 0) return MVHDP_OK;
    std::vector<uint8_t> cls((size_t)rows);
    GHIP(g, hipSetDevice(L0->device));
    GHIP(g, hipMemcpy(cls.data(), L0->mm.heavy, (size_t)rows, hipMemcpyDeviceToHost));
    g->woff.assign((size_t)rows + 1, 0);
    for (int64_t r = 0; r < rows; r++) g->woff[(size_t)r + 1] = g->woff[(size_t)r] + (cls[(size_t)r] == 0 ? (K + 1) / 2 : K);
    if (g->d_woff.empty()) { g->d_woff.assign(g->leaders.size(), nullptr); g->d_cls.assign(g->leaders.size(), nullptr); g->xpack.assign(g->leaders.size(), nullptr); }
    for (size_t l = 0; l < g->leaders.size(); l++) {
        mvhdp_ctx* L = g->members[g->leaders[l]];
        GHIP(g, hipSetDevice(L->device));
        if (!g->d_woff[l]) GHIP(g, hipMalloc(&g->d_woff[l], (size_t)(rows + 1) * sizeof(int64_t)));
        if (!g->d_cls[l]) GHIP(g, hipMalloc(&g->d_cls[l], (size_t)rows));
        if (!g->xpack[l]) GHIP(g, hipMalloc(&g->xpack[l], (size_t)rows * K * sizeof(int32_t)));
        GHIP(g, hipMemcpy(g->d_woff[l], g->woff.data(), (size_t)(rows + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
        GHIP(g, hipMemcpy(g->d_cls[l], cls.data(), (size_t)rows, hipMemcpyHostToDevice));
    }
    g->pack_hash = (int32_t)(g->woff[(size_t)rows] % 4091) + 1;
    g->pack_ready = true;
    return MVHDP_OK;
}

static void pack_reset(mvhdp_group_ctx* g) { g->pack_on = false; g->pack_ready = false; g->pack_hash = 0; g->sweeps_since_counts = 0; }

// rows [r0, r1) of every leader's delta buffer, packed, all-reduced and unpacked again: in stream order, nothing returns early
static int allreduce_rows_packed(mvhdp_group_ctx* g, int64_t r0, int64_t r1)
{
    XErr x{g};
    Rccl* r = &g_rccl;
    const int64_t w0 = g->woff[(size_t)r0], w1 = g->woff[(size_t)r1];
    const int K = g->members[0]->mm.K;
    const int grid = (int)std::min<int64_t>((r1 - r0 + 3) / 4, 8192);
    for (size_t l = 0; l < g->leaders.size(); l++) {
        mvhdp_ctx* L = g->members[g->leaders[l]];
        x.hip(hipSetDevice(L->device), "hipSetDevice");
        hipLaunchKernelGGL(pack_rows_kernel, dim3(std::max(grid, 1)), dim3(256), 0, L->stream, L->mm.delta, g->d_cls[l], g->d_woff[l], g->xpack[l], r0, r1, K);
        x.hip(hipGetLastError(), "pack_rows_kernel");
    }
    if (!g->comms.empty() && w1 > w0) {
        if (g->comms.size() > 1) x.nccl(r->GroupStart(), "ncclGroupStart");
        for (size_t l = 0; l < g->leaders.size(); l++) {
            mvhdp_ctx* L = g->members[g->leaders[l]];
            x.hip(hipSetDevice(L->device), "hipSetDevice");
            x.nccl(r->AllReduce(g->xpack[l] + w0, g->xpack[l] + w0, (size_t)(w1 - w0), ncclInt32, ncclSum, g->comms[l], L->stream), "ncclAllReduce");
        }
        if (g->comms.size() > 1) x.nccl(r->GroupEnd(), "ncclGroupEnd");
    }
    for (size_t l = 0; l < g->leaders.size(); l++) {
        mvhdp_ctx* L = g->members[g->leaders[l]];
        x.hip(hipSetDevice(L->device), "hipSetDevice");
        hipLaunchKernelGGL(unpack_rows_kernel, dim3(std::max(grid, 1)), dim3(256), 0, L->stream, L->mm.delta, g->d_cls[l], g->d_woff[l], g->xpack[l], r0, r1, K);
        x.hip(hipGetLastError(), "unpack_rows_kernel");
    }
    g->last_exchange_bytes += (w1 - w0) * (long long)sizeof(int32_t);
    return x.rc;
}

// buildInitialTypeTopicCounts PTM:600-652 over every shard: local counts, then the sum over all members of all ranks.
// Collective; the failure protocol of group_step below: a rank whose recount failed contributes zeros and a 1 in the status word behind
// the tokensPerTopic part, every rank enters the same all-reduce and all return an error from this call.
extern "C" int mvhdp_group_build_counts(mvhdp_group g)
{
    CHECK_G(g);
    DeviceGuard dg;
    if (g->async_pending) {                               // (what is on the wire is dropped: the recount below starts from the assignments)
        for (size_t l = 0; l < g->leaders.size(); l++) { hipSetDevice(g->members[g->leaders[l]]->device); hipStreamSynchronize(g->comm[l]); }
        g->async_pending = false;
    }
    const int n = (int)g->members.size();
    const int64_t len = counts_len_of(g->members[0]);
    int local_err = MVHDP_OK;
    auto note = [&](int rc, const std::string& what) { if (rc != MVHDP_OK && local_err == MVHDP_OK) { local_err = rc; g->err = what; } };
    XErr x{g};
    pack_reset(g);                                        // (new counts: the rows' classes are decided again)
    for (int i = 0; i < n; i++) {
        mvhdp_ctx* h = g->members[i];
        const int rc = mvhdp_build_counts(h);
        if (rc != MVHDP_OK) note(rc, "member " + std::to_string(i) + ": " + h->err);
        // a stale delta buffer (a failed exchange left the peers' sums in it) must not survive the recount
        if (!h->delta_clean) { x.hip(hipSetDevice(h->device), "hipSetDevice"); x.hip(hipMemsetAsync(h->mm.delta, 0, (size_t)(len + MVHDP_TAIL_WORDS) * sizeof(int32_t), h->stream), "hipMemsetAsync"); h->delta_clean = true; h->delta_pending = false; }
    }
    if (local_err != MVHDP_OK)
        for (int i = 0; i < n; i++) { mvhdp_ctx* h = g->members[i]; x.hip(hipSetDevice(h->device), "hipSetDevice"); x.hip(hipMemsetAsync(h->mm.counts, 0, (size_t)len * sizeof(int32_t), h->stream), "hipMemsetAsync"); }
    for (int i = 0; i < n; i++) { x.hip(hipSetDevice(g->members[i]->device), "hipSetDevice"); x.hip(hipEventRecord(g->ev_swept[i], g->members[i]->stream), "hipEventRecord"); }
    note(reduce_local(g, true), g->err);
    note(x.rc, g->err);
    for (size_t l = 0; l < g->leaders.size(); l++) {
        mvhdp_ctx* L = g->members[g->leaders[l]];
        x.hip(hipSetDevice(L->device), "hipSetDevice");
        hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, L->stream, L->mm.counts + len, local_err != MVHDP_OK ? 1 : 0);
    }
    note(allreduce_range(g, true, 0, len + 1), g->err);
    note(fan_out_range(g, true, 0, len + 1), g->err);
    int32_t failed_ranks = 0;
    mvhdp_ctx* L0 = g->members[g->leaders[0]];
    x.hip(hipSetDevice(L0->device), "hipSetDevice");
    x.hip(hipMemcpyAsync(&failed_ranks, L0->mm.counts + len, sizeof failed_ranks, hipMemcpyDeviceToHost, L0->stream), "hipMemcpyAsync");
    for (int i = 0; i < n; i++) {
        x.hip(hipSetDevice(g->members[i]->device), "hipSetDevice");
        x.hip(hipStreamSynchronize(g->members[i]->stream), "hipStreamSynchronize");
    }
    note(x.rc, g->err);
    if (local_err == MVHDP_OK && failed_ranks == 0)
        for (int i = 0; i < n; i++) { const int rc = mvhdp_counts_written(g->members[i]); if (rc != MVHDP_OK) note(rc, "member " + std::to_string(i) + ": " + g->members[i]->err); }
    else
        for (int i = 0; i < n; i++) { g->members[i]->counts_stale = true; g->members[i]->have_trees = false; }
    if (local_err != MVHDP_OK) return local_err;
    if (failed_ranks != 0)
        GFAIL(g, MVHDP_ERR_STATE, "the recount failed on " + std::to_string(failed_ranks) + " other rank(s) of the group (or the collective broke): the counts are not valid");
    return MVHDP_OK;
}

// One exchange-terminated step: every member sweeps (all of its entities, or one segment of them) with NO_APPLY, the deltas are
// summed over the group and applied by every replica.  st: per-member statistics of this step.
//
// Failure protocol (one process per GPU: a rank must never leave the others inside a collective).  Whatever happens locally, every
// rank enters the SAME collectives -- their number depends only on replicated facts: the chunk count, the segment count taken from the
// flags, whether the hyper-parameters hold inactive topics -- and none of the calls in between returns early (XErr).  A rank whose sweep
// or local sum failed (or whose host called mvhdp_group_abort) contributes zero deltas and a 1 in the status word behind the
// tokensPerTopic part, which is summed with that part: every rank reads the sum at the end of the step and all return an error together.
// After such an error the replicas agree with each other but not with the failed rank's assignments: mvhdp_group_build_counts (a recount
// from z, collective) makes the model consistent again.  A rank that fails LATER in the step (a HIP call of its own exchange) returns
// its error alone, having entered every collective: its host raises mvhdp_group_abort and calls the next sweep, which then fails on
// every rank together.  Whoever does not apply the sum leaves a delta buffer that is marked dirty: the next sweep (or the recount)
// clears it before anything is added to it again.
static int group_step(mvhdp_group_ctx* g, uint32_t sweep_idx, uint64_t seed, uint32_t flags, std::vector<mvhdp_sweep_stats>& st)
{
    const int n = (int)g->members.size();
    const bool live = (flags & MVHDP_SWEEP_LIVE) != 0;
    std::vector<PendingSweep> ps((size_t)n);
    int local_err = MVHDP_OK;
    auto note = [&](int rc, const std::string& what) { if (rc != MVHDP_OK && local_err == MVHDP_OK) { local_err = rc; g->err = what; } };
    if (g->abort_raised) { note(MVHDP_ERR_STATE, "the host raised mvhdp_group_abort on this rank"); g->abort_raised = false; }
    // 1. every member's sweep goes on its device before any is waited for; every sweep that was begun is finished, whatever happened
    for (int i = 0; i < n && local_err == MVHDP_OK; i++) {
        mvhdp_ctx* h = g->members[i];
        // the trees a pipelined apply left behind are those of the counts this sweep starts from (a live sweep rebuilds per segment itself)
        const uint32_t reuse = (h->have_trees && !live) ? MVHDP_SWEEP_REUSE_TREES : 0u;
        const int rc = mvhdp_sweep_begin(h, sweep_idx, seed, flags | MVHDP_SWEEP_NO_APPLY | reuse, nullptr, nullptr, ps[i]);
        if (rc != MVHDP_OK) note(rc, "member " + std::to_string(i) + ": " + h->err);
    }
    for (int i = 0; i < n; i++) {
        if (!ps[i].open) continue;
        const int rc = mvhdp_sweep_finish(g->members[i], ps[i], &st[i]);
        if (rc != MVHDP_OK) note(rc, "member " + std::to_string(i) + ": " + g->members[i]->err);
    }
    XErr x{g};
    const MvModel& mm = g->members[0]->mm;
    const int64_t rows = mm.rowbase[mm.M], K = mm.K, nk_off = rows * K, len = counts_len_of(g->members[0]);
    auto drop_local = [&]() {
        // nothing of this process enters the sum; its members' assignments may have moved without their counts: a recount is due
        for (int i = 0; i < n; i++) {
            mvhdp_ctx* h = g->members[i];
            hipSetDevice(h->device);
            hipMemsetAsync(h->mm.delta, 0, (size_t)len * sizeof(int32_t), h->stream);
            h->delta_pending = false; h->counts_stale = true; h->have_trees = false; h->rows_applied = -1;
        }
    };
    if (local_err != MVHDP_OK) drop_local();
    for (int i = 0; i < n; i++) {
        mvhdp_ctx* h = g->members[i];
        x.hip(hipSetDevice(h->device), "hipSetDevice");
        x.hip(hipEventRecord(g->ev_swept[i], h->stream), "hipEventRecord");
    }
    // 2. the exchange, as a pipeline in stream order: the tokensPerTopic part (and the status word) first -- every tree needs all of
    //    it --, then the n_wk rows in `chunks` ranges: while range i+1 is on the wire, range i is applied and its F+trees rebuilt
    //    (mvhdp_apply_delta_rows).  The collectives are issued whatever the local status; what touches a member's model is not.
    mvhdp_ctx* L0 = g->members[g->leaders[0]];
    x.hip(hipSetDevice(L0->device), "hipSetDevice");
    x.hip(hipEventRecord(g->ev_x0, L0->stream), "hipEventRecord");
    {
        const bool was_ok = local_err == MVHDP_OK;
        note(reduce_local(g, false), g->err);
        note(x.rc, g->err);
        if (was_ok && local_err != MVHDP_OK) {
            // the local sum itself failed: what the leaders hold is not this process's share -- zeros and a raised status word instead
            for (int i = 0; i < n; i++) { hipSetDevice(g->members[i]->device); hipStreamSynchronize(g->members[i]->stream); }
            drop_local();
        }
    }
    // the words behind the tokensPerTopic part, summed with it: [0] ranks whose sweep failed; [1], [2] this process's proposal for the
    // packed exchange and its square (0: none yet) -- every rank proposes the same layout iff n * sum(h^2) == (sum h)^2 with sum h > 0,
    // a test every rank makes on the same two sums, so all switch to the packed exchange behind the same sweep
    const int32_t prop = (!g->pack_on && g->pack_ready && local_err == MVHDP_OK) ? g->pack_hash : 0;
    for (size_t l = 0; l < g->leaders.size(); l++) {
        mvhdp_ctx* L = g->members[g->leaders[l]];
        x.hip(hipSetDevice(L->device), "hipSetDevice");
        hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, L->stream, L->mm.delta + len, local_err != MVHDP_OK ? 1 : 0);
        hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, L->stream, L->mm.delta + len + 1, prop);
        hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, L->stream, L->mm.delta + len + 2, prop * prop);
    }
    g->last_exchange_bytes += (len + 3 - nk_off) * (long long)sizeof(int32_t);
    int xrc = allreduce_range(g, false, nk_off, len + 3);
    { const int rc = fan_out_range(g, false, nk_off, len + 3); if (xrc == MVHDP_OK) xrc = rc; }
    bool applying = local_err == MVHDP_OK && xrc == MVHDP_OK;
    if (applying) for (int i = 0; i < n && applying; i++) { const int rc = mvhdp_apply_delta_begin(g->members[i]); if (rc) { note(rc, "member " + std::to_string(i) + ": " + g->members[i]->err); applying = false; } }
    const int nch = (int)std::max<int64_t>(1, std::min<int64_t>(g->chunks, rows));
    for (int c = 0; c < nch; c++) {
        const int64_t r0 = rows * c / nch, r1 = rows * (c + 1) / nch;
        if (r1 <= r0) continue;
        if (g->pack_on) { const int rc = allreduce_rows_packed(g, r0, r1); if (xrc == MVHDP_OK) xrc = rc; }
        else { const int rc = allreduce_range(g, false, r0 * K, r1 * K); if (xrc == MVHDP_OK) xrc = rc; g->last_exchange_bytes += (r1 - r0) * K * (long long)sizeof(int32_t); }     // (issued whatever happened before)
        { const int rc = fan_out_range(g, false, r0 * K, r1 * K); if (xrc == MVHDP_OK) xrc = rc; }
        if (applying && xrc == MVHDP_OK)
            for (int i = 0; i < n && applying; i++) { const int rc = mvhdp_apply_delta_rows(g->members[i], r0, r1); if (rc) { note(rc, "member " + std::to_string(i) + ": " + g->members[i]->err); applying = false; } }
    }
    // 3. UPD:263-270 across shards: the first activating delta in (entity, view, position) order wins on every replica alike
    long long key = LLONG_MAX;
    const bool has_inactive = g->members[0]->mm.first_inactive >= 0;       // the hyper-parameters are replicated: every rank answers alike
    if (has_inactive) {
        if (local_err == MVHDP_OK) for (int i = 0; i < n; i++) key = std::min<long long>(key, (long long)st[i].activation_key);
        if (!g->comms.empty() && g->nranks > 1) {
            Rccl* r = &g_rccl;
            for (size_t l = 0; l < g->leaders.size(); l++) {
                mvhdp_ctx* L = g->members[g->leaders[l]];
                x.hip(hipSetDevice(L->device), "hipSetDevice");
                x.hip(hipMemcpyAsync(g->d_key[l], &key, sizeof key, hipMemcpyHostToDevice, L->stream), "hipMemcpyAsync");
            }
            if (g->comms.size() > 1) x.nccl(r->GroupStart(), "ncclGroupStart");
            for (size_t l = 0; l < g->leaders.size(); l++) {
                mvhdp_ctx* L = g->members[g->leaders[l]];
                x.hip(hipSetDevice(L->device), "hipSetDevice");
                x.nccl(r->AllReduce(g->d_key[l], g->d_key[l], 1, ncclInt64, ncclMin, g->comms[l], L->stream), "activation key all-reduce");
            }
            if (g->comms.size() > 1) x.nccl(r->GroupEnd(), "ncclGroupEnd");
            x.hip(hipSetDevice(L0->device), "hipSetDevice");
            x.hip(hipMemcpyAsync(&key, g->d_key[0], sizeof key, hipMemcpyDeviceToHost, L0->stream), "hipMemcpyAsync");
            x.hip(hipStreamSynchronize(L0->stream), "hipStreamSynchronize");
        }
    }
    if (xrc != MVHDP_OK && local_err == MVHDP_OK) local_err = xrc;         // (g->err was set where the collective failed)
    // the status word of the whole group, read behind everything this step put on the first device's stream
    int32_t tail[3] = {0, 0, 0};
    x.hip(hipSetDevice(L0->device), "hipSetDevice");
    x.hip(hipEventRecord(g->ev_x1, L0->stream), "hipEventRecord");
    x.hip(hipMemcpyAsync(tail, L0->mm.delta + len, sizeof tail, hipMemcpyDeviceToHost, L0->stream), "hipMemcpyAsync");
    x.hip(hipStreamSynchronize(L0->stream), "hipStreamSynchronize");
    note(x.rc, g->err);
    const int32_t failed_ranks = tail[0];
    // (the test of the proposals: made from the reduced words alone, so every rank decides alike; a failed sweep decides nothing)
    if (!g->pack_on && failed_ranks == 0 && tail[1] > 0 && (long long)g->nranks * tail[2] == (long long)tail[1] * tail[1]) g->pack_on = true;
    const int32_t topic = key == LLONG_MAX ? -1 : MVHDP_ACT_KEY_TOPIC(key), view = key == LLONG_MAX ? -1 : MVHDP_ACT_KEY_VIEW(key);
    for (int i = 0; i < n; i++) {
        mvhdp_ctx* h = g->members[i];
        if (applying) {
            const int rc2 = mvhdp_apply_delta_end(h, topic, view);
            if (rc2 != MVHDP_OK) note(rc2, "member " + std::to_string(i) + ": " + h->err);
        } else {
            hipSetDevice(h->device); hipStreamSynchronize(h->stream);
            if (h->rows_applied >= 0) { h->rows_applied = -1; h->have_trees = false; }     // a bracket this step opened and could not close
            // The collectives ran in place on this member's delta buffer: it now holds the OTHER ranks' sums (or half an exchange), which
            // nothing will apply.  Mark it dirty: the next sweep's enqueue and mvhdp_group_build_counts clear a dirty buffer first.
            h->counts_stale = true; h->delta_pending = false; h->delta_clean = false;
        }
        st[i].activation_key = key; st[i].activated_topic = topic; st[i].activated_modality = view;
        st[i].activations = topic >= 0 ? 1 : 0;
    }
    float ms = 0;
    hipSetDevice(L0->device);
    if (hipEventElapsedTime(&ms, g->ev_x0, g->ev_x1) == hipSuccess) g->last_exchange_ms += ms;
    if (local_err != MVHDP_OK) return local_err;
    if (failed_ranks != 0) {
        // the other replicas applied the same (partial) sum and agree with each other; the model as a whole needs the recount
        for (int i = 0; i < n; i++) g->members[i]->counts_stale = true;
        GFAIL(g, MVHDP_ERR_STATE, "the sweep failed on " + std::to_string(failed_ranks) + " other rank(s) of the group (or the collective broke): call mvhdp_group_build_counts on every rank");
    }
    return MVHDP_OK;
}

// ---- MVHDP_SWEEP_ASYNC_EXCHANGE: the collective off the critical path (live sweeps only) ----
// A live sweep across shards is AD-LDA anyway: every replica is live for its own entities and stale for the others'.  Here the
// other shards' deltas arrive one sweep later still: sweep t keeps its own changes in place, its deltas go on the wire at once --
// on a stream of their own, beside sweep t+1 -- and are added to the replica when sweep t+1 has been sampled.  The step is then
// max(sampling, collective) instead of their sum; the chain sees the other shards' tokens one to two sweeps late instead of zero
// to one (measured from members on one device: profiles/r04_ll_curves.md).  mvhdp_group_drain waits for what is in flight and makes
// every replica the global model again; the statistics and mvhdp_group_build_counts drain by themselves.
static int async_buffers(mvhdp_group_ctx* g)
{
    if (!g->xbuf.empty()) return MVHDP_OK;
    const size_t bytes = (size_t)(counts_len_of(g->members[0]) + MVHDP_TAIL_WORDS) * sizeof(int32_t);
    for (mvhdp_ctx* h : g->members) {
        int32_t *x = nullptr, *o = nullptr;
        GHIP(g, hipSetDevice(h->device));
        GHIP(g, hipMalloc(&x, bytes)); g->xbuf.push_back(x);
        GHIP(g, hipMalloc(&o, bytes)); g->sbuf.push_back(o);
        GHIP(g, hipMemset(x, 0, bytes)); GHIP(g, hipMemset(o, 0, bytes));
    }
    for (int l : g->leaders) {
        hipStream_t st; hipEvent_t ev;
        GHIP(g, hipSetDevice(g->members[l]->device));
        // the collective of the asynchronous exchange runs BESIDE the next sweep: a high-priority stream has a hardware queue outside the
        // pool the sweep's streams share (mvhdp_plan.h) -- on a normal one it could land in the sweep kernels' queue and wait for them
        int least = 0, greatest = 0;
        st = nullptr;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest < least &&
            hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest) != hipSuccess) { (void)hipGetLastError(); st = nullptr; }
        if (!st) GHIP(g, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        g->comm.push_back(st);
        GHIP(g, hipEventCreateWithFlags(&ev, hipEventDisableTiming)); g->ev_xfer.push_back(ev);
    }
    return MVHDP_OK;
}

// what is in flight lands: every member adds the other shards' share of the exchanged deltas; *failed: ranks whose sweep had failed.
// Issues everything it has to issue whatever fails on the way (a step of the asynchronous exchange calls it in front of its own
// collective, which the peers are about to enter).
static int async_land(mvhdp_group_ctx* g, int32_t* failed)
{
    *failed = 0;
    if (!g->async_pending) return MVHDP_OK;
    XErr x{g};
    const int64_t len = counts_len_of(g->members[0]);
    for (size_t i = 0; i < g->members.size(); i++) {
        mvhdp_ctx* h = g->members[i];
        size_t l = 0; while (g->leaders[l] != g->leader_of[i]) l++;
        x.hip(hipSetDevice(h->device), "hipSetDevice");
        x.hip(hipStreamWaitEvent(h->stream, g->ev_xfer[l], 0), "hipStreamWaitEvent");
        const int grid = (int)std::min<int64_t>((len + 255) / 256, 8192);
        hipLaunchKernelGGL(add_remote_kernel, dim3(grid), dim3(256), 0, h->stream, h->mm.counts, g->xbuf[i], g->sbuf[i], len);
        x.hip(hipGetLastError(), "add_remote_kernel");
        h->have_trees = false;
    }
    mvhdp_ctx* L0 = g->members[g->leaders[0]];
    x.hip(hipSetDevice(L0->device), "hipSetDevice");
    x.hip(hipMemcpyAsync(failed, g->xbuf[g->leaders[0]] + len, sizeof(int32_t), hipMemcpyDeviceToHost, L0->stream), "hipMemcpyAsync");
    for (mvhdp_ctx* h : g->members) { x.hip(hipSetDevice(h->device), "hipSetDevice"); x.hip(hipStreamSynchronize(h->stream), "hipStreamSynchronize"); }
    g->async_pending = false;
    return x.rc;
}

extern "C" int mvhdp_group_drain(mvhdp_group g)
{
    CHECK_G(g);
    DeviceGuard dg;
    int32_t failed = 0;
    int rc = async_land(g, &failed);
    if (rc) { for (mvhdp_ctx* h : g->members) h->counts_stale = true; return rc; }
    if (failed != 0) {
        for (mvhdp_ctx* h : g->members) h->counts_stale = true;
        GFAIL(g, MVHDP_ERR_STATE, "a sweep whose deltas were still on the wire had failed on " + std::to_string(failed) + " rank(s): call mvhdp_group_build_counts on every rank");
    }
    return MVHDP_OK;
}

static int group_step_async(mvhdp_group_ctx* g, uint32_t sweep_idx, uint64_t seed, uint32_t flags, std::vector<mvhdp_sweep_stats>& st)
{
    const int n = (int)g->members.size();
    // (decided from replicated facts, before anything is on a device: every rank refuses alike)
    if (g->members[0]->mm.first_inactive >= 0) GFAIL(g, MVHDP_ERR_UNSUPPORTED, "group_sweep: ASYNC_EXCHANGE with inactive topics (their activation has to be agreed on before the next sweep)");
    // the exchange buffers are allocated by the FIRST such sweep, before any peer can be inside this step's collective: a rank that
    // cannot allocate fails here alone, and its host follows the protocol of a rank that cannot go on (mvhdp_group_abort)
    int rc = async_buffers(g); if (rc) return rc;
    const int64_t len = counts_len_of(g->members[0]);
    const size_t bytes = (size_t)len * sizeof(int32_t);
    std::vector<PendingSweep> ps((size_t)n);
    int local_err = MVHDP_OK;
    auto note = [&](int r, const std::string& what) { if (r != MVHDP_OK && local_err == MVHDP_OK) { local_err = r; g->err = what; } };
    if (g->abort_raised) { note(MVHDP_ERR_STATE, "the host raised mvhdp_group_abort on this rank"); g->abort_raised = false; }
    // 1. this sweep, live on every member's own replica (which lacks the other shards' previous sweep: still on the wire)
    for (int i = 0; i < n && local_err == MVHDP_OK; i++) {
        const int r = mvhdp_sweep_begin(g->members[i], sweep_idx, seed, (flags & ~MVHDP_SWEEP_ASYNC_EXCHANGE) | MVHDP_SWEEP_NO_APPLY, nullptr, nullptr, ps[i]);
        if (r != MVHDP_OK) note(r, "member " + std::to_string(i) + ": " + g->members[i]->err);
    }
    for (int i = 0; i < n; i++) {
        if (!ps[i].open) continue;
        const int r = mvhdp_sweep_finish(g->members[i], ps[i], &st[i]);
        if (r != MVHDP_OK) note(r, "member " + std::to_string(i) + ": " + g->members[i]->err);
    }
    // From here to the all-reduce below nothing returns: the peers enter that collective whatever happens on this rank (XErr).
    XErr x{g};
    // 2. the previous sweep's exchange has had this sweep's time: it lands now (counts were restored to the sweep-start snapshot by
    //    the NO_APPLY form: first this sweep's own deltas go back in, then the other shards' share of the previous one)
    for (int i = 0; i < n; i++) {
        mvhdp_ctx* h = g->members[i];
        x.hip(hipSetDevice(h->device), "hipSetDevice");
        if (local_err != MVHDP_OK) { x.hip(hipMemsetAsync(h->mm.delta, 0, bytes, h->stream), "hipMemsetAsync"); h->counts_stale = true; }
        else x.hip(launch_add_into(h->mm.counts, h->mm.delta, len, h->stream), "add_into_kernel");
    }
    int32_t failed = 0;
    note(async_land(g, &failed), g->err);
    note(x.rc, g->err);
    if (local_err != MVHDP_OK)                             // (a failure found after the deltas went back in: this rank still ships zeros and a raised word)
        for (int i = 0; i < n; i++) { mvhdp_ctx* h = g->members[i]; hipSetDevice(h->device); hipMemsetAsync(h->mm.delta, 0, bytes, h->stream); h->counts_stale = true; }
    // 3. this sweep's deltas go on the wire (a copy: the delta buffer is the next sweep's), the collective on its own stream
    for (int i = 0; i < n; i++) {
        mvhdp_ctx* h = g->members[i];
        x.hip(hipSetDevice(h->device), "hipSetDevice");
        x.hip(hipMemcpyAsync(g->xbuf[i], h->mm.delta, bytes, hipMemcpyDeviceToDevice, h->stream), "hipMemcpyAsync");
        x.hip(hipMemcpyAsync(g->sbuf[i], h->mm.delta, bytes, hipMemcpyDeviceToDevice, h->stream), "hipMemcpyAsync");
        x.hip(hipMemsetAsync(h->mm.delta, 0, bytes, h->stream), "hipMemsetAsync");
        hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, h->stream, g->xbuf[i] + len, (g->leader_of[i] == i && local_err != MVHDP_OK) ? 1 : 0);
        x.hip(hipEventRecord(g->ev_swept[i], h->stream), "hipEventRecord");
        h->delta_pending = false; h->delta_clean = true; h->have_trees = false;
    }
    for (size_t l = 0; l < g->leaders.size(); l++) {
        const int li = g->leaders[l];
        mvhdp_ctx* L = g->members[li];
        x.hip(hipSetDevice(L->device), "hipSetDevice");
        for (int i = 0; i < n; i++) {
            if (g->leader_of[i] != li) continue;
            x.hip(hipStreamWaitEvent(g->comm[l], g->ev_swept[i], 0), "hipStreamWaitEvent");
            if (i != li) x.hip(launch_add_into(g->xbuf[li], g->xbuf[i], len, g->comm[l]), "add_into_kernel");
        }
    }
    if (!g->comms.empty()) {
        Rccl* r = &g_rccl;
        if (g->comms.size() > 1) x.nccl(r->GroupStart(), "ncclGroupStart");
        for (size_t l = 0; l < g->leaders.size(); l++) {
            mvhdp_ctx* L = g->members[g->leaders[l]];
            x.hip(hipSetDevice(L->device), "hipSetDevice");
            int32_t* buf = g->xbuf[g->leaders[l]];
            x.nccl(r->AllReduce(buf, buf, (size_t)(len + 1), ncclInt32, ncclSum, g->comms[l], g->comm[l]), "ncclAllReduce");
        }
        if (g->comms.size() > 1) x.nccl(r->GroupEnd(), "ncclGroupEnd");
    }
    for (size_t l = 0; l < g->leaders.size(); l++) {
        const int li = g->leaders[l];
        x.hip(hipSetDevice(g->members[li]->device), "hipSetDevice");
        for (int i = 0; i < n; i++)
            if (g->leader_of[i] == li && i != li) x.hip(hipMemcpyAsync(g->xbuf[i], g->xbuf[li], (size_t)(len + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, g->comm[l]), "hipMemcpyAsync");
        x.hip(hipEventRecord(g->ev_xfer[l], g->comm[l]), "hipEventRecord");
    }
    g->async_pending = true;
    note(x.rc, g->err);
    for (int i = 0; i < n; i++) { st[i].activation_key = LLONG_MAX; st[i].activated_topic = -1; st[i].activated_modality = -1; st[i].activations = 0; }
    if (local_err != MVHDP_OK) { for (mvhdp_ctx* h : g->members) h->counts_stale = true; return local_err; }
    if (failed != 0) {
        for (mvhdp_ctx* h : g->members) h->counts_stale = true;
        GFAIL(g, MVHDP_ERR_STATE, "the previous sweep failed on " + std::to_string(failed) + " other rank(s) of the group: call mvhdp_group_build_counts on every rank");
    }
    return MVHDP_OK;
}

// A host whose rank cannot go on (an exception outside the library, a signal it handles) calls this BEFORE its next mvhdp_group_sweep:
// that sweep then samples nothing here, enters the collectives with zero deltas and fails on every rank together -- nobody is left
// waiting inside an all-reduce for a rank that will never arrive.
extern "C" int mvhdp_group_abort(mvhdp_group g)
{
    CHECK_G(g);
    g->abort_raised = true;
    return MVHDP_OK;
}

// One Gibbs sweep of the whole model (see the head of this file).  flags: MVHDP_SWEEP_EXACT_CHAIN, MVHDP_SWEEP_GENERIC_KERNEL,
// MVHDP_SWEEP_LIVE (+ LIVE_SEGMENTS: each replica is live for its own entities and one sweep stale for the others', AD-LDA), or
// MVHDP_SWEEP_SEGMENT_APPLY (+ LIVE_SEGMENTS(n)): the deterministic segmented sweep ACROSS the shards -- every member sweeps its
// segment s, the deltas are exchanged and applied, then segment s+1: n exchanges per sweep, a token sees counts at most one segment
// old on every replica.  NO_APPLY and REUSE_TREES are the group's own business.  stats: one entry per local member, or NULL.
// Deferred and segmented modes are bit-identical to the same sweep of one handle holding all entities when the segments are cut the
// same way; a group cuts them per member (each member's own longest-first order), so the segmented sweep of a group is its own
// deterministic chain, reproducible for a given sharding.
extern "C" int mvhdp_group_sweep(mvhdp_group g, uint32_t sweep_idx, uint64_t seed, uint32_t flags, mvhdp_sweep_stats* stats)
{
    CHECK_G(g);
    if (flags & (MVHDP_SWEEP_NO_APPLY | MVHDP_SWEEP_REUSE_TREES | MVHDP_SWEEP_FROZEN | MVHDP_SWEEP_SEGMENT_OVERLAP | 0xff000000u))
        GFAIL(g, MVHDP_ERR_INVALID_ARG, "group_sweep: NO_APPLY, REUSE_TREES and ONLY_SEGMENT are set by the group itself; FROZEN is a single-handle mode");
    if ((flags & MVHDP_SWEEP_SEGMENT_APPLY) && (flags & MVHDP_SWEEP_LIVE)) GFAIL(g, MVHDP_ERR_INVALID_ARG, "group_sweep: SEGMENT_APPLY excludes LIVE");
    if ((flags & MVHDP_SWEEP_ASYNC_EXCHANGE) && !(flags & MVHDP_SWEEP_LIVE)) GFAIL(g, MVHDP_ERR_INVALID_ARG, "group_sweep: ASYNC_EXCHANGE goes with LIVE (a deferred sweep is the parity contract: every replica must hold the global counts)");
    DeviceGuard dg;
    const int n = (int)g->members.size();
    std::vector<mvhdp_sweep_stats> total((size_t)n), st((size_t)n);
    g->last_exchange_ms = 0.0;
    g->last_exchange_bytes = 0;
    int ret = MVHDP_OK;
    if (!(flags & MVHDP_SWEEP_ASYNC_EXCHANGE) && g->async_pending) { ret = mvhdp_group_drain(g); if (ret) return ret; }
    if (flags & MVHDP_SWEEP_ASYNC_EXCHANGE) {
        ret = group_step_async(g, sweep_idx, seed, flags, total);
    } else if (flags & MVHDP_SWEEP_SEGMENT_APPLY) {
        int nseg = (int)((flags >> 16) & 0xffu);
        if (nseg == 0) nseg = 4;
        // (the segment count comes from the flags alone -- every rank issues the same number of exchanges; a member with fewer entities
        // than segments simply has empty ones)
        const uint32_t base = (flags & ~(MVHDP_SWEEP_SEGMENT_APPLY | MVHDP_SWEEP_LIVE_SEGMENTS(0xff))) | MVHDP_SWEEP_LIVE_SEGMENTS(nseg);
        for (int s = 0; s < nseg && ret == MVHDP_OK; s++) {
            ret = group_step(g, sweep_idx, seed, base | MVHDP_SWEEP_ONLY_SEGMENT(s), st);
            for (int i = 0; i < n; i++) {
                if (s == 0) { total[i] = st[i]; continue; }
                total[i].tokens += st[i].tokens; total[i].changed += st[i].changed; total[i].new_mass_cnt += st[i].new_mass_cnt;
                total[i].topic_doc_mass_cnt += st[i].topic_doc_mass_cnt; total[i].word_ftree_mass_cnt += st[i].word_ftree_mass_cnt;
                total[i].oov_skipped += st[i].oov_skipped; total[i].aborted_docs += st[i].aborted_docs; total[i].exact_fallbacks += st[i].exact_fallbacks;
                total[i].sweep_kernel_ms += st[i].sweep_kernel_ms; total[i].total_ms += st[i].total_ms;
                if (total[i].activated_topic < 0 && st[i].activated_topic >= 0) {
                    total[i].activated_topic = st[i].activated_topic; total[i].activated_modality = st[i].activated_modality; total[i].activation_key = st[i].activation_key;
                }
                total[i].activations += st[i].activations;
            }
        }
    } else {
        ret = group_step(g, sweep_idx, seed, flags, total);
    }
    g->sweeps++;
    if (ret == MVHDP_OK && !(flags & MVHDP_SWEEP_ASYNC_EXCHANGE)) {
        g->sweeps_since_counts++;
        if (!g->pack_ready) (void)pack_prepare(g);           // (a failure here only means: no proposal from this rank, the exchange stays at full width)
    }
    if (stats) for (int i = 0; i < n; i++) stats[i] = total[i];
    return ret;
}

// ---------------------------------------------------------------------------------------------------------------
// The steps either side of the sweep for a sharded model (what estimate() does every optimizeInterval and
// every tenth iteration, PTM:1173-1210, PTM:1296-1320).  The counts are replicated, the entities are not: a statistic
// over n_wk / n_k is any member's, a statistic over the entities is the members' put together --
//   in ONE process, in ascending doc_id_base, carrying the running sum from member to member: the additions are those
//   of a single handle holding every entity, in the same order, bit for bit;
//   across processes (one member each), every rank's partial result is made known to all ranks (an all-reduce of a buffer
//   that is zero outside the rank's own slot: exact) and added in rank order: deterministic for a given sharding, equal to
//   the single handle's to rounding.  Collective there: every rank must call.
// ---------------------------------------------------------------------------------------------------------------
namespace {

// The cross-rank pieces of the statistics (one process per GPU).  Same rule as the sweep: a rank whose own part failed still enters
// the collective -- with zeros and a raised status slot -- so that every rank learns of it from the result and all return an error from
// the same call; nothing is allocated on the way (the scratch buffer exists since mvhdp_group_create_rank) and no HIP failure returns
// before the all-reduce has been issued.  local_status: MVHDP_OK or this rank's error; *failed_ranks: ranks that raised their slot.

// all[r*n + i] = rank r's vals[i]; single process: all = vals
int xrank_gather_f64(mvhdp_group_ctx* g, const double* vals, int n, int local_status, std::vector<double>& all, int* failed_ranks)
{
    *failed_ranks = 0;
    if (!g->multi_process || g->nranks <= 1) { all.assign(vals, vals + n); return MVHDP_OK; }
    mvhdp_ctx* L = g->members[g->leaders[0]];
    const size_t per = (size_t)n + 1, total = (size_t)g->nranks * per;
    if (total * sizeof(double) > XSCRATCH_BYTES || !g->d_scratch) GFAIL(g, MVHDP_ERR_INVALID_ARG, "xrank_gather_f64: message larger than the scratch buffer");   // (replicated sizes: every rank refuses alike)
    std::vector<double> host(total, 0.0);
    if (local_status == MVHDP_OK) std::copy(vals, vals + n, host.begin() + (size_t)g->rank0 * per);
    host[(size_t)g->rank0 * per + n] = local_status == MVHDP_OK ? 0.0 : 1.0;
    XErr x{g};
    double* d = (double*)g->d_scratch;
    x.hip(hipSetDevice(L->device), "hipSetDevice");
    x.hip(hipMemcpyAsync(d, host.data(), total * sizeof(double), hipMemcpyHostToDevice, L->stream), "hipMemcpyAsync");
    x.nccl(g_rccl.AllReduce(d, d, total, ncclDouble, ncclSum, g->comms[0], L->stream), "ncclAllReduce");
    x.hip(hipMemcpyAsync(host.data(), d, total * sizeof(double), hipMemcpyDeviceToHost, L->stream), "hipMemcpyAsync");
    x.hip(hipStreamSynchronize(L->stream), "hipStreamSynchronize");
    if (x.rc != MVHDP_OK) return x.rc;
    all.resize((size_t)g->nranks * n);
    for (int r = 0; r < g->nranks; r++) {
        std::copy(host.begin() + (size_t)r * per, host.begin() + (size_t)r * per + n, all.begin() + (size_t)r * n);
        if (host[(size_t)r * per + n] != 0.0) (*failed_ranks)++;
    }
    return MVHDP_OK;
}

// vals[i] <- sum over ranks (integers: any order); in pieces of the scratch buffer, the status word behind the last one
int xrank_sum_i32(mvhdp_group_ctx* g, int32_t* vals, size_t n, int local_status, int* failed_ranks)
{
    *failed_ranks = 0;
    if (!g->multi_process || g->nranks <= 1) return MVHDP_OK;
    mvhdp_ctx* L = g->members[g->leaders[0]];
    if (!g->d_scratch) GFAIL(g, MVHDP_ERR_STATE, "xrank_sum_i32: no scratch buffer");
    const size_t cap = XSCRATCH_BYTES / sizeof(int32_t) - 1;
    std::vector<int32_t> piece(cap + 1);
    XErr x{g};
    int32_t* d = (int32_t*)g->d_scratch;
    x.hip(hipSetDevice(L->device), "hipSetDevice");
    for (size_t off = 0; off < n || off == 0; off += cap) {
        const size_t k = std::min(cap, n - off);
        const bool last = off + k >= n;
        if (local_status == MVHDP_OK) std::copy(vals + off, vals + off + k, piece.begin()); else std::fill(piece.begin(), piece.begin() + k, 0);
        piece[k] = local_status == MVHDP_OK ? 0 : 1;
        const size_t cnt = k + (last ? 1 : 0);
        x.hip(hipMemcpyAsync(d, piece.data(), cnt * sizeof(int32_t), hipMemcpyHostToDevice, L->stream), "hipMemcpyAsync");
        x.nccl(g_rccl.AllReduce(d, d, cnt, ncclInt32, ncclSum, g->comms[0], L->stream), "ncclAllReduce");
        x.hip(hipMemcpyAsync(piece.data(), d, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, L->stream), "hipMemcpyAsync");
        x.hip(hipStreamSynchronize(L->stream), "hipStreamSynchronize");
        if (x.rc == MVHDP_OK) { std::copy(piece.begin(), piece.begin() + k, vals + off); if (last) *failed_ranks = piece[k]; }
        if (last) break;
    }
    return x.rc;
}

// A statistic of a group whose asynchronous exchange is still in flight lands it first.  Returns non-zero only for what EVERY rank sees
// alike (a peer's failed sweep, read from the exchanged status word); a failure of this rank alone goes to *local_status, and the
// caller still enters its collectives.
int land_before_statistics(mvhdp_group_ctx* g, int* local_status)
{
    if (!g->async_pending) return MVHDP_OK;
    int32_t failed = 0;
    const int rc = async_land(g, &failed);
    if (rc != MVHDP_OK) { if (*local_status == MVHDP_OK) *local_status = rc; for (mvhdp_ctx* h : g->members) h->counts_stale = true; if (!g->multi_process) return rc; return MVHDP_OK; }
    if (failed != 0) {
        for (mvhdp_ctx* h : g->members) h->counts_stale = true;
        GFAIL(g, MVHDP_ERR_STATE, "a sweep whose deltas were still on the wire had failed on " + std::to_string(failed) + " rank(s): call mvhdp_group_build_counts on every rank");
    }
    return MVHDP_OK;
}

#define XFAILED(g, failed) GFAIL(g, MVHDP_ERR_STATE, "the statistic failed on " + std::to_string(failed) + " other rank(s) of the group")

}  // namespace

// the hyper-parameters are replicated: every local member takes them (every rank calls with the same values)
extern "C" int mvhdp_group_set_hyper(mvhdp_group g, const mvhdp_hyper* hy)
{
    CHECK_G(g);
    DeviceGuard dg;
    for (size_t i = 0; i < g->members.size(); i++) GMEM(g, i, mvhdp_set_hyper(g->members[i], hy));
    return MVHDP_OK;
}

// modelLogLikelihood PTM:3322-3452 of the whole model: the document part summed over every entity of every member in entity order,
// the model part (modalityCnt term, topic-word term, tokensPerTopic terms) once, from the replicated counts.
extern "C" int mvhdp_group_log_likelihood(mvhdp_group g, double* out)
{
    CHECK_G(g);
    if (!out) GFAIL(g, MVHDP_ERR_INVALID_ARG, "group_log_likelihood: null");
    DeviceGuard dg;
    int lst = MVHDP_OK;
    { const int rcd = land_before_statistics(g, &lst); if (rcd) return rcd; }
    const bool multi = g->multi_process && g->nranks > 1;
    const int M = g->members[0]->mm.M;
    for (int m = 0; m < M; m++) {
        double ll = 0;
        int64_t cnt = 0;
        for (int i : g->by_entity) {
            if (lst != MVHDP_OK) break;
            const int rc = mvhdp_ll_doc_accumulate(g->members[i], m, &ll, &cnt);
            if (rc != MVHDP_OK) { lst = rc; g->err = "member " + std::to_string(i) + ": " + g->members[i]->err; if (!multi) return rc; }
        }
        if (multi) {
            const double mine[2] = {ll, (double)cnt};
            std::vector<double> all;
            int failed = 0;
            const std::string keep = g->err;
            const int rc = xrank_gather_f64(g, mine, 2, lst, all, &failed);
            if (lst != MVHDP_OK) { g->err = keep; return lst; }
            if (rc) return rc;
            if (failed) XFAILED(g, failed);
            ll = 0; double c = 0;
            for (int r = 0; r < g->nranks; r++) { ll += all[(size_t)2 * r]; c += all[(size_t)2 * r + 1]; }
            cnt = (int64_t)c;
        }
        GMEM(g, 0, mvhdp_ll_model_finish(g->members[0], m, ll, cnt, &out[m]));
    }
    return MVHDP_OK;
}

// topicDocCounts[m][k][c] and docLengthCounts[m][len] (PTM:620-651, UPD:220-232) over every entity of every member
extern "C" int mvhdp_group_doc_topic_hist(mvhdp_group g, int32_t m, int32_t* hist, int32_t hist_len, int32_t* doc_len_counts, int32_t len_len)
{
    CHECK_G(g);
    DeviceGuard dg;
    int lst = MVHDP_OK;
    { const int rcd = land_before_statistics(g, &lst); if (rcd) return rcd; }
    const bool multi = g->multi_process && g->nranks > 1;
    const int K = g->members[0]->mm.K;
    const size_t nh = hist ? (size_t)K * (size_t)std::max(hist_len, 0) : 0, nl = doc_len_counts ? (size_t)std::max(len_len, 0) : 0;
    std::vector<int32_t> th(nh), tl(nl);
    if (hist) std::fill(hist, hist + nh, 0);
    if (doc_len_counts) std::fill(doc_len_counts, doc_len_counts + nl, 0);
    std::string keep;
    for (size_t i = 0; i < g->members.size() && lst == MVHDP_OK; i++) {
        const int rc = mvhdp_get_doc_topic_hist(g->members[i], m, hist ? th.data() : nullptr, hist_len, doc_len_counts ? tl.data() : nullptr, len_len);
        if (rc != MVHDP_OK) { lst = rc; g->err = keep = "member " + std::to_string(i) + ": " + g->members[i]->err; if (!multi) return rc; break; }
        for (size_t q = 0; q < nh; q++) hist[q] += th[q];
        for (size_t q = 0; q < nl; q++) doc_len_counts[q] += tl[q];
    }
    // (both collectives are entered whatever the first one said: their number depends on the arguments alone, which every rank passes alike)
    int f1 = 0, f2 = 0;
    const int rc1 = hist ? xrank_sum_i32(g, hist, nh, lst, &f1) : MVHDP_OK;
    const int rc2 = doc_len_counts ? xrank_sum_i32(g, doc_len_counts, nl, lst, &f2) : MVHDP_OK;
    if (lst != MVHDP_OK) { if (!keep.empty()) g->err = keep; return lst; }
    if (rc1) return rc1;
    if (rc2) return rc2;
    if (f1 || f2) XFAILED(g, std::max(f1, f2));
    return MVHDP_OK;
}

// countHistogram of optimizeBeta PTM:2295-2309: a statistic of the replicated n_wk -- any member's
extern "C" int mvhdp_group_count_histogram(mvhdp_group g, int32_t m, int32_t* hist, int32_t len)
{
    CHECK_G(g);
    DeviceGuard dg;
    // (no collective of its own; a drain is the landing of something every rank has already issued)
    int lst = MVHDP_OK;
    { const int rcd = land_before_statistics(g, &lst); if (rcd) return rcd; }
    if (lst != MVHDP_OK) return lst;
    GMEM(g, 0, mvhdp_get_count_histogram(g->members[0], m, hist, len));
    return MVHDP_OK;
}

// optimizeP PTM:2706-2792: sums[m][i] over every entity, in entity order
extern "C" int mvhdp_group_view_overlap_sums(mvhdp_group g, double* sums)
{
    CHECK_G(g);
    if (!sums) GFAIL(g, MVHDP_ERR_INVALID_ARG, "group_view_overlap_sums: null");
    DeviceGuard dg;
    int lst = MVHDP_OK;
    { const int rcd = land_before_statistics(g, &lst); if (rcd) return rcd; }
    const bool multi = g->multi_process && g->nranks > 1;
    const int M = g->members[0]->mm.M;
    for (int i = 0; i < M * M; i++) sums[i] = 0.0;
    std::string keep;
    for (int i : g->by_entity) {
        if (lst != MVHDP_OK) break;
        const int rc = mvhdp_view_overlap_accumulate(g->members[i], sums);
        if (rc != MVHDP_OK) { lst = rc; g->err = keep = "member " + std::to_string(i) + ": " + g->members[i]->err; if (!multi) return rc; }
    }
    if (multi) {
        std::vector<double> all;
        int failed = 0;
        const int rc = xrank_gather_f64(g, sums, M * M, lst, all, &failed);
        if (lst != MVHDP_OK) { if (!keep.empty()) g->err = keep; return lst; }
        if (rc) return rc;
        if (failed) XFAILED(g, failed);
        for (int i = 0; i < M * M; i++) { double a = 0; for (int r = 0; r < g->nranks; r++) a += all[(size_t)r * M * M + i]; sums[i] = a; }
    }
    return MVHDP_OK;
}

// optimizeGamma's document level PTM:2415-2433 (mvhdp_gamma_doc_statistics): every entity draws from its own stream (global entity
// id), so the two sums over a sharded model are the same random variables; the members' sums are added in entity order
extern "C" int mvhdp_group_gamma_doc_statistics(mvhdp_group g, int32_t m, double gamma_m, uint64_t seed, uint32_t round, double* qs, double* qw)
{
    CHECK_G(g);
    if (!qs || !qw) GFAIL(g, MVHDP_ERR_INVALID_ARG, "group_gamma_doc_statistics: null");
    DeviceGuard dg;
    int lst = MVHDP_OK;
    { const int rcd = land_before_statistics(g, &lst); if (rcd) return rcd; }
    const bool multi = g->multi_process && g->nranks > 1;
    double a = 0, b = 0;
    std::string keep;
    for (int i : g->by_entity) {
        if (lst != MVHDP_OK) break;
        double x = 0, y = 0;
        const int rc = mvhdp_gamma_doc_statistics(g->members[i], m, gamma_m, seed, round, &x, &y);
        if (rc != MVHDP_OK) { lst = rc; g->err = keep = "member " + std::to_string(i) + ": " + g->members[i]->err; if (!multi) return rc; break; }
        a += x; b += y;
    }
    if (multi) {
        const double mine[2] = {a, b};
        std::vector<double> all;
        int failed = 0;
        const int rc = xrank_gather_f64(g, mine, 2, lst, all, &failed);
        if (lst != MVHDP_OK) { if (!keep.empty()) g->err = keep; return lst; }
        if (rc) return rc;
        if (failed) XFAILED(g, failed);
        a = 0; b = 0;
        for (int r = 0; r < g->nranks; r++) { a += all[(size_t)2 * r]; b += all[(size_t)2 * r + 1]; }
    }
    *qs = a; *qw = b;
    return MVHDP_OK;
}
