/*
 * fake_rccl.c -- TEST INFRASTRUCTURE, never shipped and never a fallback of the product.
 *
 * A stand-in for librccl with exactly the nine entry points libmvhdp.so resolves at run time
 * (mvtopicmodel_amd/csrc/mvhdp_group.hip: ncclGetVersion, ncclGetUniqueId, ncclCommInitRank, ncclCommInitAll,
 * ncclCommDestroy, ncclAllReduce, ncclGroupStart, ncclGroupEnd, ncclGetErrorString), so that the one-process-per-GPU
 * path of the library (mvhdp_group_create_rank with nranks > 1) can run with SEVERAL PROCESSES ON ONE GPU -- the real
 * RCCL refuses two ranks on one device, and no multi-GPU node is available to the tests.  Selected by the test through
 * MVHDP_RCCL_LIB; tests/test_gpu_group_ranks.py builds it with gcc.
 *
 * The collective runs through POSIX shared memory and is ordered on the HIP stream it is given, like the real one:
 *     hipMemcpyAsync(device -> this rank's slot)   |  in stream order: nothing of the host waits here
 *     hipLaunchHostFunc: barrier, every rank reduces all slots in RANK ORDER into its own result buffer, barrier
 *     hipMemcpyAsync(result -> device)
 * so a missing event dependency in the library shows as wrong numbers here exactly as it would on hardware.
 * What it checks on top of the arithmetic, because these are the ways a multi-rank path goes wrong:
 *   - every rank must enter the SAME collective (count, type, operation) in the same order: a mismatch breaks the
 *     communicator and poisons the result (0x7f bytes: a summed status word then reads "failed");
 *   - the collectives of one communicator must execute in the order they were enqueued (RCCL wants them serialised);
 *   - a rank that never arrives (it died, or it left the protocol) is noticed after FAKE_RCCL_TIMEOUT_MS (default
 *     20000): the communicator breaks, the result is poisoned, every later call returns ncclSystemError -- the peers
 *     get an error instead of a hang.
 * Environment: FAKE_RCCL_SLOT_BYTES (default 4 MiB: larger messages go in pieces), FAKE_RCCL_TIMEOUT_MS,
 * FAKE_RCCL_SYNC=1 (no host function: the host waits for the stream and reduces in place; diagnostics),
 * FAKE_RCCL_LOG=1 (one line per collective on stderr).
 */
#define _GNU_SOURCE
#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>

#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3, ncclAvg = 4 } ncclRedOp_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;

#define FAKE_MAGIC 0x46524343u /* "FRCC" */
#define MAX_RANKS 64

typedef struct {
    _Atomic uint32_t magic;
    uint32_t nranks;
    uint64_t slot_bytes;
    _Atomic uint32_t arrive, generation;        /* barrier */
    _Atomic uint32_t broken;                    /* a rank timed out or the ranks disagreed: the communicator is dead */
    _Atomic uint32_t attached;
    /* what each rank believes the current collective is (written before the first barrier, compared after it) */
    struct { uint64_t seq, count; uint32_t dtype, op; } what[MAX_RANKS];
} shm_hdr;

struct ncclComm {
    int rank, nranks;
    shm_hdr* hdr;
    unsigned char* slots;                       /* [nranks][slot_bytes] behind the header */
    size_t map_bytes, slot_bytes;
    int registered;
    unsigned char* result;                      /* process-local, pinned: the reduced piece on its way back to the device */
    uint64_t enq_seq;                           /* collectives (pieces) enqueued */
    _Atomic uint64_t run_seq;                   /* pieces executed */
    long timeout_ms;
    int sync_mode, log;
    _Atomic int local_broken;
    /* like the real library, collectives of one communicator are serialised even when the caller changes streams between them */
    hipStream_t last_stream; hipEvent_t last_event; int have_last;
};
typedef struct ncclComm* ncclComm_t;

static size_t dtype_size(ncclDataType_t t)
{
    switch (t) { case ncclInt8: case ncclUint8: return 1; case ncclFloat16: return 2; case ncclInt32: case ncclUint32: case ncclFloat32: return 4; default: return 8; }
}

static double now_ms(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}

/* all ranks meet; -1 when the communicator is (or becomes) broken */
static int barrier(struct ncclComm* c)
{
    shm_hdr* h = c->hdr;
    if (c->nranks == 1) return 0;
    if (atomic_load(&h->broken)) return -1;
    const uint32_t gen = atomic_load(&h->generation);
    if (atomic_fetch_add(&h->arrive, 1) + 1 == (uint32_t)c->nranks) {
        atomic_store(&h->arrive, 0);
        atomic_fetch_add(&h->generation, 1);
        return 0;
    }
    const double t0 = now_ms();
    unsigned spins = 0;
    while (atomic_load(&h->generation) == gen) {
        if (atomic_load(&h->broken)) return -1;
        if ((++spins & 63u) == 0) {
            if (now_ms() - t0 > (double)c->timeout_ms) {
                atomic_store(&h->broken, 1);
                fprintf(stderr, "[fake_rccl] rank %d: a rank did not arrive within %ld ms -- communicator broken\n", c->rank, c->timeout_ms);
                return -1;
            }
            usleep(50);
        } else sched_yield();
    }
    return 0;
}

typedef struct {
    struct ncclComm* c;
    uint64_t seq;
    size_t count;                               /* elements of this piece */
    ncclDataType_t dtype; ncclRedOp_t op;
} piece;

static void reduce_piece(struct ncclComm* c, const piece* p)
{
    const size_t n = p->count;
    const int R = c->nranks;
#define RED(T, EXPR) do { T* out = (T*)c->result; \
        for (size_t i = 0; i < n; i++) { T a = ((const T*)(c->slots))[i]; \
            for (int r = 1; r < R; r++) { const T b = ((const T*)(c->slots + (size_t)r * c->slot_bytes))[i]; a = (EXPR); } out[i] = a; } } while (0)
    switch (p->dtype) {
    case ncclInt32: case ncclUint32:
        if (p->op == ncclSum) RED(int32_t, (int32_t)((uint32_t)a + (uint32_t)b)); else if (p->op == ncclMin) RED(int32_t, a < b ? a : b); else RED(int32_t, a > b ? a : b);
        break;
    case ncclInt64: case ncclUint64:
        if (p->op == ncclSum) RED(int64_t, (int64_t)((uint64_t)a + (uint64_t)b)); else if (p->op == ncclMin) RED(int64_t, a < b ? a : b); else RED(int64_t, a > b ? a : b);
        break;
    case ncclFloat64:
        if (p->op == ncclSum) RED(double, a + b); else if (p->op == ncclMin) RED(double, a < b ? a : b); else RED(double, a > b ? a : b);
        break;
    case ncclFloat32:
        if (p->op == ncclSum) RED(float, a + b); else if (p->op == ncclMin) RED(float, a < b ? a : b); else RED(float, a > b ? a : b);
        break;
    default: memset(c->result, 0x7f, n * dtype_size(p->dtype)); break;
    }
#undef RED
}

/* the body of one piece: runs in stream order (host function) or directly (sync mode) */
static void run_piece(void* arg)
{
    piece* p = (piece*)arg;
    struct ncclComm* c = p->c;
    shm_hdr* h = c->hdr;
    const size_t bytes = p->count * dtype_size(p->dtype);
    int ok = !atomic_load(&c->local_broken);
    const uint64_t expect = atomic_fetch_add(&c->run_seq, 1);
    if (ok && expect != p->seq) {
        fprintf(stderr, "[fake_rccl] rank %d: collectives of one communicator executed out of their enqueue order (piece %llu ran as number %llu)\n",
                c->rank, (unsigned long long)p->seq, (unsigned long long)expect);
        atomic_store(&h->broken, 1); ok = 0;
    }
    if (ok) {
        h->what[c->rank].seq = p->seq; h->what[c->rank].count = p->count; h->what[c->rank].dtype = (uint32_t)p->dtype; h->what[c->rank].op = (uint32_t)p->op;
        atomic_thread_fence(memory_order_seq_cst);
        ok = barrier(c) == 0;
    }
    if (ok) {
        for (int r = 0; r < c->nranks; r++)
            if (h->what[r].seq != p->seq || h->what[r].count != p->count || h->what[r].dtype != (uint32_t)p->dtype || h->what[r].op != (uint32_t)p->op) {
                fprintf(stderr, "[fake_rccl] rank %d: rank %d is in another collective (piece %llu count %llu type %u op %u against piece %llu count %llu type %u op %u)\n",
                        c->rank, r, (unsigned long long)h->what[r].seq, (unsigned long long)h->what[r].count, h->what[r].dtype, h->what[r].op,
                        (unsigned long long)p->seq, (unsigned long long)p->count, (unsigned)p->dtype, (unsigned)p->op);
                atomic_store(&h->broken, 1); ok = 0;
                break;
            }
    }
    if (ok) {
        reduce_piece(c, p);
        ok = barrier(c) == 0;                   /* nobody refills a slot before everybody has read it */
    }
    if (!ok) { atomic_store(&c->local_broken, 1); memset(c->result, 0x7f, bytes); }
    if (c->log) fprintf(stderr, "[fake_rccl] rank %d piece %llu: %zu x type %d op %d %s\n", c->rank, (unsigned long long)p->seq, p->count, (int)p->dtype, (int)p->op, ok ? "ok" : "BROKEN");
    free(p);
}

ncclResult_t ncclGetVersion(int* version) { if (!version) return ncclInvalidArgument; *version = 1; return ncclSuccess; }   /* (1: nobody mistakes it for a release) */

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error (fake_rccl)";
    case ncclSystemError: return "fake_rccl: the communicator is broken (a rank did not arrive, or the ranks disagreed about the collective)";
    case ncclInvalidArgument: return "fake_rccl: invalid argument";
    case ncclInvalidUsage: return "fake_rccl: unsupported usage";
    default: return "fake_rccl: error";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    static _Atomic unsigned counter;
    if (!id) return ncclInvalidArgument;
    memset(id->internal, 0, NCCL_UNIQUE_ID_BYTES);
    struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, NCCL_UNIQUE_ID_BYTES, "/mvhdp_fake_rccl_%d_%lx_%u", (int)getpid(), (unsigned long)ts.tv_nsec ^ (unsigned long)ts.tv_sec << 20, atomic_fetch_add(&counter, 1));
    return ncclSuccess;
}

static struct ncclComm* comm_alloc(int rank, int nranks)
{
    struct ncclComm* c = (struct ncclComm*)calloc(1, sizeof *c);
    if (!c) return NULL;
    c->rank = rank; c->nranks = nranks;
    const char* e;
    c->slot_bytes = (e = getenv("FAKE_RCCL_SLOT_BYTES")) ? (size_t)strtoull(e, NULL, 10) : ((size_t)4 << 20);
    if (c->slot_bytes < 64) c->slot_bytes = 64;
    c->slot_bytes = (c->slot_bytes + 63) & ~(size_t)63;
    c->timeout_ms = (e = getenv("FAKE_RCCL_TIMEOUT_MS")) ? atol(e) : 20000;
    c->sync_mode = (e = getenv("FAKE_RCCL_SYNC")) ? atoi(e) : 0;
    c->log = (e = getenv("FAKE_RCCL_LOG")) ? atoi(e) : 0;
    return c;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    struct ncclComm* c = comm_alloc(rank, nranks);
    if (!c) return ncclSystemError;
    char name[NCCL_UNIQUE_ID_BYTES + 1];
    memcpy(name, id.internal, NCCL_UNIQUE_ID_BYTES); name[NCCL_UNIQUE_ID_BYTES] = 0;
    if (name[0] != '/') { free(c); return ncclInvalidArgument; }
    const size_t hdr_bytes = (sizeof(shm_hdr) + 4095) & ~(size_t)4095;
    c->map_bytes = hdr_bytes + (size_t)nranks * c->slot_bytes;
    int creator = 1;
    int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 && errno == EEXIST) { creator = 0; fd = shm_open(name, O_RDWR, 0600); }
    if (fd < 0) { free(c); return ncclSystemError; }
    if (creator && ftruncate(fd, (off_t)c->map_bytes) != 0) { close(fd); shm_unlink(name); free(c); return ncclSystemError; }
    if (!creator) {                             /* wait until the creator has sized the object */
        const double t0 = now_ms();
        struct stat sb;
        for (;;) {
            if (fstat(fd, &sb) == 0 && (size_t)sb.st_size >= c->map_bytes) break;
            if (now_ms() - t0 > (double)c->timeout_ms) { close(fd); free(c); return ncclSystemError; }
            usleep(200);
        }
    }
    void* p = mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { if (creator) shm_unlink(name); free(c); return ncclSystemError; }
    c->hdr = (shm_hdr*)p;
    c->slots = (unsigned char*)p + hdr_bytes;
    if (creator) {
        c->hdr->nranks = (uint32_t)nranks; c->hdr->slot_bytes = c->slot_bytes;
        atomic_store(&c->hdr->magic, FAKE_MAGIC);
    } else {
        const double t0 = now_ms();
        while (atomic_load(&c->hdr->magic) != FAKE_MAGIC) {
            if (now_ms() - t0 > (double)c->timeout_ms) { munmap(p, c->map_bytes); free(c); return ncclSystemError; }
            usleep(200);
        }
        if (c->hdr->nranks != (uint32_t)nranks || c->hdr->slot_bytes != c->slot_bytes) { munmap(p, c->map_bytes); free(c); return ncclInvalidArgument; }
    }
    atomic_fetch_add(&c->hdr->attached, 1);
    const int met = barrier(c);                 /* like the real ncclCommInitRank: collective */
    if (creator) shm_unlink(name);              /* the mappings keep the object alive; nothing is left in /dev/shm if a test dies */
    if (met != 0) { munmap(p, c->map_bytes); free(c); return ncclSystemError; }
    if (hipHostRegister(p, c->map_bytes, hipHostRegisterDefault) == hipSuccess) c->registered = 1; else (void)hipGetLastError();
    if (hipHostMalloc((void**)&c->result, c->slot_bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        if (c->registered) hipHostUnregister(p);
        munmap(p, c->map_bytes); free(c); return ncclUnhandledCudaError;
    }
    *comm = c;
    return ncclSuccess;
}

/* one process driving several devices is not what this stand-in is for (the host functions of two streams of ONE process would
 * have to wait for each other); a single device is a one-rank communicator */
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist)
{
    (void)devlist;
    if (!comms || ndev != 1) return ncclInvalidUsage;
    struct ncclComm* c = comm_alloc(0, 1);
    if (!c) return ncclSystemError;
    c->hdr = (shm_hdr*)calloc(1, sizeof(shm_hdr));
    c->slots = NULL;
    if (!c->hdr || hipHostMalloc((void**)&c->result, c->slot_bytes, hipHostMallocDefault) != hipSuccess) { free(c->hdr); free(c); return ncclSystemError; }
    c->hdr->nranks = 1;
    comms[0] = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    if (c->result) hipHostFree(c->result);
    if (c->last_event) hipEventDestroy(c->last_event);
    if (c->slots) {
        if (c->registered) hipHostUnregister((void*)c->hdr);
        munmap((void*)c->hdr, c->map_bytes);
    } else free(c->hdr);
    free(c);
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) { return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) { return ncclSuccess; }

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t dtype, ncclRedOp_t op, ncclComm_t c, hipStream_t stream)
{
    if (!c || (!sendbuff && count) || (!recvbuff && count)) return ncclInvalidArgument;
    if (op != ncclSum && op != ncclMin && op != ncclMax) return ncclInvalidUsage;
    if (atomic_load(&c->local_broken) || (c->slots && atomic_load(&c->hdr->broken))) return ncclSystemError;
    const size_t es = dtype_size(dtype);
    if (c->nranks == 1) {
        if (sendbuff != recvbuff && count && hipMemcpyAsync(recvbuff, sendbuff, count * es, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
        return ncclSuccess;
    }
    if (c->have_last && c->last_stream != stream && hipStreamWaitEvent(stream, c->last_event, 0) != hipSuccess) return ncclUnhandledCudaError;
    const size_t per = c->slot_bytes / es;
    for (size_t off = 0; off < count || (count == 0 && off == 0); off += per) {
        const size_t n = count - off < per ? count - off : per;
        piece* p = (piece*)malloc(sizeof *p);
        if (!p) return ncclSystemError;
        p->c = c; p->seq = c->enq_seq++; p->count = n; p->dtype = dtype; p->op = op;
        unsigned char* my = c->slots + (size_t)c->rank * c->slot_bytes;
        if (n && hipMemcpyAsync(my, (const unsigned char*)sendbuff + off * es, n * es, hipMemcpyDeviceToHost, stream) != hipSuccess) { free(p); return ncclUnhandledCudaError; }
        if (c->sync_mode) {
            if (hipStreamSynchronize(stream) != hipSuccess) { free(p); return ncclUnhandledCudaError; }
            run_piece(p);
        } else if (hipLaunchHostFunc(stream, run_piece, p) != hipSuccess) { free(p); return ncclUnhandledCudaError; }
        if (n && hipMemcpyAsync((unsigned char*)recvbuff + off * es, c->result, n * es, hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
        if (count == 0) break;
    }
    if (!c->last_event && hipEventCreateWithFlags(&c->last_event, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventRecord(c->last_event, stream) != hipSuccess) return ncclUnhandledCudaError;
    c->last_stream = stream; c->have_last = 1;
    return ncclSuccess;
}
