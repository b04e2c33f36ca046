// mvhdp_kernels.hip — hand-written CDNA4 (gfx950) kernels of the multi-view HDP
// collapsed-Gibbs sweep.  One 64-lane wavefront owns one entity (document) for
// all of its views; see DESIGN.md for the layout and the roofline of each kernel.
//
// Reference (hmetaxa/MVTopicModel, src/main/java/org/madgik/...):
//   sweep_kernel       <- FastQMVWVWorkerRunnable.sampleTopicsForOneDoc  WRK:301-601
//                         + FastQMVWVUpdaterRunnable count updates         UPD:197-218
//   build_trees_kernel <- FastQMVWVParallelTopicModel.buildFTrees          PTM:2660-2696, FTree FT:96-109
//   build_counts_kernel<- buildInitialTypeTopicCounts                      PTM:600-652
//   draw_p_kernel      <- per-document view weights                        WRK:327-337 (MALLET Randoms.nextBeta)
//   tree_sample()      <- FTree.sample                                     FT:111-136
//
// Build with -ffp-contract=off: every fp64 expression below is evaluated in the
// reference's order with one rounding per operation (Java never fuses a*b+c).
#include "mvhdp_device.h"
#include "../../include/mvhdp.h"

#include "mvhdp_wave.h"
#include <algorithm>

// ---------------------------------------------------------------------------
// build_counts: PTM:600-652.  One thread per token, int32 atomics.
// n_k is privatised in LDS per block and flushed once.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_counts_kernel(MvModel mm, int m, int64_t n_tokens)
{
    extern __shared__ int nk_local[];
    const int K = mm.K;
    for (int i = threadIdx.x; i < K; i += blockDim.x) nk_local[i] = 0;
    __syncthreads();
    int32_t* nwk = mm.counts;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_tokens; i += stride) {
        int topic = mm.z[m][i];
        if (topic == MVHDP_UNASSIGNED_TOPIC) continue;            // PTM:634
        int type = mm.tok[m][i];
        if (type < 0 || type >= mm.V[m] || topic < 0 || topic >= K) continue;
        atomicAdd(&nk_local[topic], 1);                           // PTM:640
        atomicAdd(&nwk[(mm.rowbase[m] + type) * K + topic], 1);   // PTM:643
    }
    __syncthreads();
    int32_t* nk = mm.counts + mm.rowbase[mm.M] * K + (int64_t)m * K;
    for (int i = threadIdx.x; i < K; i += blockDim.x)
        if (nk_local[i]) atomicAdd(&nk[i], nk_local[i]);
}

hipError_t mvhdp_launch_build_counts(const MvModel& mm, const int64_t* n_tokens, hipStream_t s)
{
    const int K = mm.K;
    size_t total = (size_t)(mm.rowbase[mm.M] * K + (int64_t)mm.M * K) * sizeof(int32_t);
    hipError_t e = hipMemsetAsync(mm.counts, 0, total, s);
    if (e != hipSuccess) return e;
    for (int m = 0; m < mm.M; m++) {
        if (n_tokens[m] <= 0) continue;
        int64_t blocks = (n_tokens[m] + 255) / 256;
        int grid = (int)(blocks < 2048 ? blocks : 2048);
        hipLaunchKernelGGL(build_counts_kernel, dim3(grid), dim3(256), (size_t)K * sizeof(int), s, mm, m, n_tokens[m]);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ---------------------------------------------------------------------------
// build_trees: PTM:2660-2696 + FTree.constructTree FT:96-109.
// One wave per (view, type) row; the 2K-double tree is assembled in LDS level
// by level (children always have larger indices, so descending depth is safe)
// and written out whole, coalesced.
// ---------------------------------------------------------------------------
// The tree of one (view, type) row from its K leaves in t[K .. 2K): FT:96-109 level by level (children before parents), then the
// same numbers once more grouped for the descent (MvModel::dtab) and tree[1] by itself.  One wave (a block of 64 threads) per row.
template <bool COHERENT = false>
__device__ __forceinline__ void tree_from_leaves(const MvModel& mm, int64_t row, double* t, int lane, bool write_full)
{
    const int K = mm.K;
    if (lane == 0) t[0] = 0.0;
    __syncthreads();
    if (K > 1) {
        int dmax = 31 - __clz(K - 1);                          // depth of node K-1
        for (int d = dmax; d >= 0; d--) {
            int lo = 1 << d, hi = min(2 << d, K);
            for (int i = lo + lane; i < hi; i += WAVE) t[i] = t[2 * i] + t[2 * i + 1];   // FT:105
            __syncthreads();
        }
    }
    if (write_full) {                                      // FTree.tree itself: generic kernel, get_tree, init_from_trees
        double* out = mm.trees + row * 2 * K;
        for (int i = lane; i < 2 * K; i += WAVE) out[i] = t[i];
    }
    // COHERENT (heavy_refresh_kernel: the table is rewritten WHILE sweep kernels on the other XCDs read it): stores of agent scope -- written
    // through this XCD's L2, which is not coherent with the others' for plain stores -- matched by agent-scope loads in the readers
    if (lane == 0) { if (COHERENT) __hip_atomic_store(&mm.root[row], t[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else mm.root[row] = t[1]; }   // tree[1] by itself: 8 bytes a type, L2-resident (WRK:519 without the walk)
    // the same numbers once more, grouped for the descent (see MvModel::dtab)
    double* dt = mm.dtab + row * (int64_t)mm.dt_nblk * 8;
    // (written 16 bytes a lane, consecutive lanes consecutive addresses: unit u = pair j of block x -- whole 128-byte lines per store
    // instead of a quarter of 64 different sectors; round 4: the rebuild is what every segment border of a segmented or live sweep pays)
    double2* o = (double2*)dt;
    for (int u = lane; u < mm.dt_nblk * 4; u += WAVE) {
        const int x = u >> 2, j = u & 3;
        // block x of the row: block 0 is rooted at depth 0, then 2^dep blocks for dep = dt_f, dt_f+3, ... (arithmetic, not the
        // dt_base / dt_depth arrays: a lane-varying index into a kernel-argument array would go through scratch memory)
        int base = 0, dep = 0;
        if (x >= 1) {
            base = 1; dep = mm.dt_f;
            while (x >= base + (1 << dep)) { base += 1 << dep; dep += 3; }
        }
        const int b = (1 << dep) + (x - base);
        // the block's eight doubles: L[b]; L[2b], L[2b+1]; L[4b .. 4b+3]; tree[1] (block 0 only) -- L[n] = tree[2n], the left-child sum
        double v2[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int q = 2 * j + h;
            const int node = (q == 0) ? b : (q < 3) ? 2 * b + (q - 1) : 4 * b + (q - 3);
            v2[h] = (q == 7) ? ((x == 0) ? t[1] : 0.0) : ((node < K) ? t[2 * node] : 0.0);
        }
        if (COHERENT) {
            __hip_atomic_store(&dt[2 * u], v2[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&dt[2 * u + 1], v2[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else o[u] = make_double2(v2[0], v2[1]);
    }
    __syncthreads();
}

// apply_first: the multi-GPU pipeline's form (mvhdp_apply_delta_rows): the row's all-reduced deltas are added to the
// counts (UPD:197-207) on the way in -- counts += delta, delta = 0 -- and the tree is built from the updated row; the
// tokensPerTopic part has been applied before (mvhdp_apply_delta_begin), every tree needs all of it.
// The 16-bit mirror of the row (MvModel::counts16) is written here too, and with it the row's weight class (MvModel::heavy): a row
// whose type holds more than 65534 tokens in all is HEAVY -- every mirror cell 65535 = "look in the 32-bit table" --, any other row is
// LIGHT and its mirror cells are its counts (none can ever reach 65535, however the tokens move between topics).
// from_mirror (a segment border of a live16 sweep, where the light rows' atomics went to the mirror): light rows are READ from the
// mirror and written through to the 32-bit table; the flags and the mirror stay as they are.
// TB: cells a lane holds in flight in each pass (below).  For a rebuild that has the chip to itself 8 where a row is long (K > 512; C5,
// K = 1000: 1.00 -> 0.73 ms) and 4 below (C4 0.196 -> 0.175 ms, C3 0.102 -> 0.097); 1 -- a quarter of the registers -- for a rebuild beside
// a resident sweep kernel (overlapped live segments): what counts there is how many of its waves fit into the registers the samplers
// leave, not how long one of them takes (TB = 8 there: C3 live 5.2 -> 5.8 ms).  gpurun_out/r5l, r5m.
template <int TB>
__global__ __launch_bounds__(64) void build_trees_kernel(MvModel mm, bool inference_leaves, bool write_full, int64_t row_begin, int64_t row_end,
                                                         bool apply_first, unsigned long long* negatives, bool from_mirror, bool only_heavy = false)
{
    extern __shared__ double t[];                  // 2K doubles
    const int K = mm.K, lane = threadIdx.x;
    const int64_t nrows = mm.rowbase[mm.M];
    const int32_t* nk_all = mm.counts + nrows * K;
    int neg = 0;
    __builtin_amdgcn_s_setprio(3);                 // (a live sweep rebuilds the next segment's trees beside the current segment's samplers)
    for (int64_t row = row_begin + blockIdx.x; row < row_end; row += gridDim.x) {
        if (only_heavy && mm.heavy[row] != MVHDP_ROW_HEAVY) continue;   // (live-rows form: every other word samples its tree branch from its live row)
        int m = 0;
        while (m + 1 < mm.M && row >= mm.rowbase[m + 1]) m++;
        int32_t* cnt = mm.counts + row * K;
        int32_t* dl = mm.delta + row * K;
        const int32_t* nk = nk_all + (int64_t)m * K;
        const double* al = mm.alpha + (int64_t)m * (K + 1);
        const double beta = mm.beta[m], beta_sum = mm.beta_sum[m], gamma = mm.gamma[m];
        uint16_t* c16 = mm.counts16 + row * K;
        const bool light_src = from_mirror && mm.heavy[row] != MVHDP_ROW_HEAVY;
        bool hv = false;
        // Both passes over the row read in batches of TB cells a lane, every load of a batch issued before the first is used: one wave
        // works on one row, so a loop that loads, divides and stores cell by cell pays a cache round trip per iteration -- 20 us a row
        // at K = 400, which is what bounded this kernel (0.19 ms for 60 000 rows with 25 rows in flight per CU; round 4).
        // first pass over the row: the updater's catch-up (apply_first), the row's weight class, the mirror
        if (!from_mirror) {
            long long sum = 0;
            for (int k0 = 0; k0 < K; k0 += WAVE * TB) {
                int cv[TB], dv[TB];
#pragma unroll
                for (int u = 0; u < TB; u++) {
                    const int k = k0 + u * WAVE + lane;
                    cv[u] = (k < K) ? cnt[k] : 0;
                    dv[u] = (apply_first && k < K) ? dl[k] : 0;
                }
#pragma unroll
                for (int u = 0; u < TB; u++) {
                    const int k = k0 + u * WAVE + lane;
                    if (k < K) {
                        int c = cv[u];
                        if (apply_first) {
                            const int d = dv[u];
                            if (d) { c += d; cnt[k] = c; dl[k] = 0; neg += c < 0; }    // UPD:202-215 logs a negative count; here it is reported
                        }
                        sum += c < 0 ? 70000 : c;                          // (a negative count is an error reported elsewhere: keep the row out of the mirror)
                    }
                }
            }
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) sum += __shfl_xor(sum, sft, WAVE);
            hv = sum > 65534;
            // (between the two: a light row whose deltas of one sweep may pass +-32767 -- they stay in the 32-bit delta table, SweepLaunch::delta16)
            if (lane == 0) mm.heavy[row] = hv ? MVHDP_ROW_HEAVY : (sum > 32767 ? MVHDP_ROW_BIG : 0);
        }
        for (int k0 = 0; k0 < K; k0 += WAVE * TB) {
            int cv[TB], nkv[TB];
            double alv[TB];
            bool inact[TB];
#pragma unroll
            for (int u = 0; u < TB; u++) {
                const int k = k0 + u * WAVE + lane;
                const bool in = k < K;
                cv[u] = in ? (light_src ? (int)c16[k] : cnt[k]) : 0;
                nkv[u] = in ? nk[k] : 0;
                alv[u] = (in && !inference_leaves) ? al[k] : 0.0;
                inact[u] = in && !inference_leaves && mm.inactive[k];
            }
#pragma unroll
            for (int u = 0; u < TB; u++) {
                const int k = k0 + u * WAVE + lane;
                if (k < K) {
                    const int c = cv[u];
                    if (light_src) cnt[k] = c;                             // the mirror is the authority for this row: write it through
                    else if (!from_mirror) c16[k] = hv ? (uint16_t)65535 : (uint16_t)c;
                    double leaf;
                    if (inference_leaves) {                                // INF:576: p_wt alone
                        leaf = ((double)c + beta) / ((double)nkv[u] + beta_sum);
                    } else if (inact[u]) {                                 // PTM:2670-2671
                        leaf = 0.0;
                    } else {
                        double p_wt = ((double)c + beta) / ((double)nkv[u] + beta_sum);   // PTM:2676
                        leaf = gamma * alv[u] * p_wt;                      // PTM:2678
                    }
                    t[K + k] = leaf;
                }
            }
        }
        tree_from_leaves(mm, row, t, lane, write_full);
    }
    if (apply_first && negatives) {
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) neg += __shfl_xor(neg, sft, WAVE);
        if (lane == 0 && neg) atomicAdd(negatives, (unsigned long long)neg);
    }
}

hipError_t mvhdp_launch_build_trees(const MvModel& mm, bool inference_leaves, bool write_full, hipStream_t s, bool beside_samplers)
{
    if (!beside_samplers) return mvhdp_launch_build_trees_rows(mm, inference_leaves, write_full, 0, mm.rowbase[mm.M], false, nullptr, s);
    const int64_t nrows = mm.rowbase[mm.M];
    if (nrows <= 0) return hipSuccess;
    int grid = (int)(nrows < 65536 ? nrows : 65536);
    hipLaunchKernelGGL(build_trees_kernel<1>, dim3(grid), dim3(64), (size_t)2 * mm.K * sizeof(double), s, mm, inference_leaves, write_full,
                       (int64_t)0, nrows, false, (unsigned long long*)nullptr, false);
    return hipGetLastError();
}

hipError_t mvhdp_launch_build_trees_rows(const MvModel& mm, bool inference_leaves, bool write_full, int64_t row_begin, int64_t row_end,
                                         bool apply_first, unsigned long long* negatives, hipStream_t s)
{
    int64_t nrows = row_end - row_begin;
    if (nrows <= 0) return hipSuccess;
    int grid = (int)(nrows < 65536 ? nrows : 65536);          // (a block per row: a tenth of that many blocks looping over rows is 6 % slower, gpurun_out/r5c)
    if (mm.K > 512)
        hipLaunchKernelGGL(build_trees_kernel<8>, dim3(grid), dim3(64), (size_t)2 * mm.K * sizeof(double), s, mm, inference_leaves, write_full,
                           row_begin, row_end, apply_first, negatives, false);
    else
        hipLaunchKernelGGL(build_trees_kernel<4>, dim3(grid), dim3(64), (size_t)2 * mm.K * sizeof(double), s, mm, inference_leaves, write_full,
                           row_begin, row_end, apply_first, negatives, false);
    return hipGetLastError();
}

hipError_t mvhdp_launch_build_trees_from_mirror(const MvModel& mm, bool write_full, hipStream_t s, bool beside_samplers)
{
    const int64_t nrows = mm.rowbase[mm.M];
    if (nrows <= 0) return hipSuccess;
    int grid = (int)(nrows < 65536 ? nrows : 65536);
    if (beside_samplers)
        hipLaunchKernelGGL(build_trees_kernel<1>, dim3(grid), dim3(64), (size_t)2 * mm.K * sizeof(double), s, mm, false, write_full,
                           (int64_t)0, nrows, false, (unsigned long long*)nullptr, true);
    else if (mm.K > 512)
        hipLaunchKernelGGL(build_trees_kernel<8>, dim3(grid), dim3(64), (size_t)2 * mm.K * sizeof(double), s, mm, false, write_full,
                           (int64_t)0, nrows, false, (unsigned long long*)nullptr, true);
    else
        hipLaunchKernelGGL(build_trees_kernel<4>, dim3(grid), dim3(64), (size_t)2 * mm.K * sizeof(double), s, mm, false, write_full,
                           (int64_t)0, nrows, false, (unsigned long long*)nullptr, true);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Overlapped segments (MVHDP_SWEEP_SEGMENT_OVERLAP, mvhdp_api.hip enqueue_overlapped): the updater's catch-up for one
// segment runs BESIDE the sampling of the next.  The model is kept twice (counts, mirror, descent tables); this kernel
// brings the copy that segment s + 2 will read up to date from the deltas of segments s - 1 and s:
//     dst += dA (+ dB), dB = 0            (apply2_counts_kernel below; the trees stay those of the sweep start: a form that also rebuilt
//                                          the row's tree here was measured in round 4 -- 2.6-3 ms beside the samplers -- and removed)
// with memory-side atomics for the count cells -- another kernel (segment s + 1's samplers) is running on the same chip,
// and an atomic is what every L2 sees at once (tools/microbench/live_staleness.hip) -- and only where a delta is not zero.
// The row's weight class does not change (a row's total is constant while every token is assigned: the plan refuses
// the mode otherwise), so a light row's mirror cells take the same delta, packed (+-d << 16 for the upper cell: no
// cell of a light row can leave [0, 65534]); a heavy row's mirror cells stay 65535.
// ---------------------------------------------------------------------------------------------------------------
// tokensPerTopic of the same update (M*K words, before the rows: every leaf needs all of it)
__global__ __launch_bounds__(256) void apply2_nk_kernel(int32_t* nk_dst, const int32_t* __restrict__ dA, int32_t* __restrict__ dB, int n, unsigned long long* negatives)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int d = dA[i];
    if (dB) { const int d2 = dB[i]; if (d2) { d += d2; dB[i] = 0; } }
    if (d) { const int c = nk_dst[i] + d; nk_dst[i] = c; if (c < 0 && negatives) atomicAdd(negatives, 1ull); }
}

// The same update without the trees (MVHDP_SWEEP_SEGMENT_OVERLAP keeps the sweep-start trees for every segment -- a deviation from the
// reference, whose updater refreshes the touched leaves per delta, UPD:242-260): dst += dA (+ dB), dB = 0, cell by cell, atomics where a delta is not zero.
// Reads two delta buffers and touches what changed: a tenth of a millisecond of the whole chip at C4, a few tenths beside the samplers.
__global__ __launch_bounds__(256) void apply2_counts_kernel(MvModel mm, const int32_t* __restrict__ dA, int32_t* __restrict__ dB, bool use_mirror,
                                                           int64_t n_cells, int64_t n_all, unsigned long long* negatives)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned int* m32 = (unsigned int*)mm.counts16;
    const int K = mm.K;
    int neg = 0;
    __builtin_amdgcn_s_setprio(3);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_all; i += stride) {
        int d = dA[i];
        if (dB) { const int d2 = dB[i]; if (d2) { d += d2; dB[i] = 0; } }
        if (!d) continue;
        const int old = __hip_atomic_fetch_add(&mm.counts[i], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        neg += old + d < 0;
        if (use_mirror && i < n_cells && mm.heavy[i / K] != MVHDP_ROW_HEAVY) __hip_atomic_fetch_add(&m32[i >> 1], (unsigned int)d << ((i & 1) * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (neg && negatives) atomicAdd(negatives, (unsigned long long)neg);
}

hipError_t mvhdp_launch_apply2_counts(const MvModel& dst, const int32_t* dA, int32_t* dB, bool use_mirror, unsigned long long* negatives, hipStream_t s)
{
    const int64_t cells = dst.rowbase[dst.M] * dst.K, all = cells + (int64_t)dst.M * dst.K;
    int grid = (int)std::min<int64_t>((all + 255) / 256, 4096);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(apply2_counts_kernel, dim3(grid), dim3(256), 0, s, dst, dA, dB, use_mirror, cells, all, negatives);
    return hipGetLastError();
}

// The copy that missed the last segment's deltas takes them (no trees: nothing samples from it before the next rebuild); d = 0.
// Nothing else runs on the chip by then: plain read-modify-write, the packed mirror cells by atomics (two cells share a word).
__global__ __launch_bounds__(256) void apply_sparse_kernel(MvModel mm, int32_t* __restrict__ d, bool use_mirror, int64_t n_cells, int64_t n_all)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned int* m32 = (unsigned int*)mm.counts16;
    const int K = mm.K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_all; i += stride) {
        const int dl = d[i];
        if (!dl) continue;
        mm.counts[i] += dl;
        d[i] = 0;
        if (use_mirror && i < n_cells && mm.heavy[i / K] != MVHDP_ROW_HEAVY) __hip_atomic_fetch_add(&m32[i >> 1], (unsigned int)dl << ((i & 1) * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

hipError_t mvhdp_launch_apply_sparse(const MvModel& dst, int32_t* d, bool use_mirror, hipStream_t s)
{
    const int64_t cells = dst.rowbase[dst.M] * dst.K, all = cells + (int64_t)dst.M * dst.K;
    int grid = (int)std::min<int64_t>((all + 255) / 256, 8192);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(apply_sparse_kernel, dim3(grid), dim3(256), 0, s, dst, d, use_mirror, cells, all);
    return hipGetLastError();
}

// Holds a stream until the work-queue head `*qhead` of a running sweep kernel has passed `threshold` entities (a live sweep's next
// segment is prepared -- trees rebuilt from the live counts -- when the current one is nearly through, not when it starts).  One wave;
// it always terminates (about two seconds at the latest): the kernel it watches never waits for anything.
__global__ __launch_bounds__(64) void gate_kernel(const unsigned long long* qhead, unsigned long long threshold)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 4000000; i++) {
        const unsigned long long v = __hip_atomic_load(qhead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= threshold) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) break;     // 2 s of the 100 MHz counter
        __builtin_amdgcn_s_sleep(32);
    }
}

hipError_t mvhdp_launch_gate(const unsigned long long* qhead, unsigned long long threshold, hipStream_t s)
{
    hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(64), 0, s, qhead, threshold);
    return hipGetLastError();
}

// counts <- mirror for the light rows (the end of a live16 sweep: the 32-bit table is the model again)
__global__ __launch_bounds__(256) void widen_mirror_kernel(MvModel mm)
{
    const int K = mm.K;
    const int64_t nrows = mm.rowbase[mm.M];
    const int lane = threadIdx.x & 63;
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); row < nrows; row += wstride) {
        if (mm.heavy[row] == MVHDP_ROW_HEAVY) continue;
        const uint16_t* c16 = mm.counts16 + row * K;
        int32_t* cnt = mm.counts + row * K;
        for (int k = lane; k < K; k += WAVE) cnt[k] = (int)c16[k];
    }
}

hipError_t mvhdp_launch_widen_mirror(const MvModel& mm, hipStream_t s)
{
    const int64_t nrows = mm.rowbase[mm.M];
    if (nrows <= 0) return hipSuccess;
    int64_t blocks = (nrows + 3) / 4;
    hipLaunchKernelGGL(widen_mirror_kernel, dim3((unsigned int)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, s, mm);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// MVHDP_SWEEP_LIVE, live-rows form (SweepLaunch::live_rows).  The reference's updater refreshes the two touched leaves of the word's
// tree with every FastQDelta (UPD:242-260 -> FT:138-147), so a worker's tree branch (WRK:533-535) always samples from the word's
// CURRENT counts; a stored tree that is rebuilt n times per sweep is up to 1/n sweep behind, and the LL curves price that (DESIGN.md
// section 2: 4 rebuilds per sweep need 1.3-1.9 sweeps per reference sweep in the side views, 32 rebuilds match the reference).  Here the
// tree branch reads the live row itself: leaf_k = coef[m][k] * (n_wk + beta_m) over all K topics by one wave-wide scan (row_sample_live in
// mvhdp_sweep_fast.hip), and tree[1] -- which EVERY token's decision needs (WRK:519) -- is MvModel::root, exact at the segment start
// (below) and moved by one fp64 atomic per changed token.  What a segment start costs is one pass over the counts: no tree, no descent
// table (282 MB at C4) to write.
//   coef[m][k] = (float)(gamma_m * alpha_mk / (n_k + betaSum_m)), 0 for an inactive topic      PTM:2670-2678
//   smp[m][k]  = coef[m][0] * beta_m + ... + coef[m][k] * beta_m (fp32, in topic order): the smoothing part of every leaf of the view
//   root[row]  = smp[m][K-1] + sum_k (double)coef[m][k] * n_wk: per lane over k = lane, lane + 64, ... ascending, then a butterfly
//                over the lanes (the order the oracle restates for the sequential pin, tests/test_gpu_live.py)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void live_coef_kernel(MvModel mm)
{
    // one block per view: the coefficients (row padded to a multiple of 8 with zeros), then (one lane, in topic order) the running sums of
    // the smoothing parts coef_k * beta
    const int K = mm.K, Kp = (K + 7) & ~7, m = blockIdx.x;
    const int32_t* nk = mm.counts + mm.rowbase[mm.M] * K + (int64_t)m * K;
    float* cf = mm.coef + (int64_t)m * Kp;
    float* smp = mm.coef + (int64_t)mm.M * Kp + (int64_t)m * K;
    for (int k = threadIdx.x; k < Kp; k += blockDim.x)
        cf[k] = (k >= K || mm.inactive[k]) ? 0.0f : (float)(mm.gamma[m] * mm.alpha[(int64_t)m * (K + 1) + k] / ((double)nk[k] + mm.beta_sum[m]));
    __syncthreads();
    if (threadIdx.x == 0) {
        const float beta32 = (float)mm.beta[m];
        float run = 0.0f;
        for (int k = 0; k < K; k++) { run += cf[k] * beta32; smp[k] = run; }
    }
}

template <int TB>
__global__ __launch_bounds__(64) void live_rows_prepare_kernel(MvModel mm, bool from_mirror, int32_t* heavy_list, unsigned int* heavy_n, int heavy_cap, int batch_cells)
{
    const int K = mm.K, lane = threadIdx.x;
    const int64_t nrows = mm.rowbase[mm.M];
    for (int64_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        int m = 0;
        while (m + 1 < mm.M && row >= mm.rowbase[m + 1]) m++;
        const int32_t* cnt = mm.counts + row * K;
        uint16_t* c16 = mm.counts16 + row * K;
        const int Kp = (K + 7) & ~7;
        const float* cf = mm.coef + (int64_t)m * Kp;
        const float smp_total = mm.coef[(int64_t)mm.M * Kp + (int64_t)m * K + K - 1];   // S_m: the smoothing parts of the view's leaves, summed
        const bool light_src = from_mirror && mm.heavy[row] != MVHDP_ROW_HEAVY;
        bool hv = false;
        if (!from_mirror) {                                          // the row's weight class, as build_trees_kernel decides it
            long long sum = 0;
            for (int k0 = 0; k0 < K; k0 += WAVE * TB) {
                int cv[TB];
#pragma unroll
                for (int u = 0; u < TB; u++) { const int k = k0 + u * WAVE + lane; cv[u] = (k < K) ? cnt[k] : 0; }
#pragma unroll
                for (int u = 0; u < TB; u++) sum += cv[u] < 0 ? 70000 : cv[u];
            }
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) sum += __shfl_xor(sum, sft, WAVE);
            hv = sum > 65534;
            if (lane == 0) mm.heavy[row] = hv ? MVHDP_ROW_HEAVY : (sum > 32767 ? MVHDP_ROW_BIG : 0);
        }
        double acc = 0.0, acc0 = 0.0;                                // acc0: the topics of the row's first register batch alone (MvModel::mass0)
        for (int k0 = 0; k0 < K; k0 += WAVE * TB) {
            int cv[TB];
            float fv[TB];
#pragma unroll
            for (int u = 0; u < TB; u++) {
                const int k = k0 + u * WAVE + lane;
                const bool in = k < K;
                cv[u] = in ? (light_src ? (int)c16[k] : cnt[k]) : 0;
                fv[u] = in ? cf[k] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < TB; u++) {
                const int k = k0 + u * WAVE + lane;
                if (k < K) {
                    if (!from_mirror) c16[k] = hv ? (uint16_t)65535 : (uint16_t)cv[u];
                    const double t = (double)fv[u] * (double)cv[u];
                    acc += t;
                    if (k < batch_cells) acc0 += t;
                }
            }
        }
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) { acc += __shfl_xor(acc, sft, WAVE); acc0 += __shfl_xor(acc0, sft, WAVE); }
        if (lane == 0) { mm.root[row] = (double)smp_total + acc; mm.mass0[row] = (float)acc0; }
        // the HEAVY rows (a few hundred at most: each holds more than 65534 tokens), listed for the kernel that keeps their stored trees current
        if (lane == 0 && heavy_list && mm.heavy[row] == MVHDP_ROW_HEAVY) {
            const unsigned int i = atomicAdd(heavy_n, 1u);
            if ((int)i < heavy_cap) heavy_list[i] = (int32_t)row;
        }
    }
}

// The stored trees of the HEAVY words of a live sweep in its live-rows form, kept current WHILE the samplers run: the reference's updater
// refreshes the touched leaves of a word's tree with every delta (UPD:242-260); a heavy word's row is not in the mirror (its cells pass
// 16 bits) and too long to sit in a lane's registers, so its tree branch walks a stored tree -- and half the tokens of a Zipf corpus
// belong to heavy words, so a tree of the segment start would bring the staleness back that the live rows remove.
// But the heavy words are FEW (170 of 60 000 at C4): a handful of waves rebuild all their trees from the live counts every few tens of
// microseconds, beside the samplers, until the host's stream says the segment's samplers are done (`stop`, set in stream order behind
// them) -- or two seconds have passed: the kernel always ends.  A sampler that reads a tree while it is being rewritten reads a mixture of
// two nearly equal trees: every descent still ends at a topic (FT:122-132), like the reference's racy reads (PTM:84-87).
__global__ __launch_bounds__(64) void heavy_refresh_kernel(MvModel mm, const int32_t* heavy_list, const unsigned int* ctl /* [0] rows listed, [1] stop */, int heavy_cap, bool write_full)
{
    extern __shared__ double t[];                  // 2K doubles
    const int K = mm.K, lane = threadIdx.x;
    const int64_t nrows = mm.rowbase[mm.M];
    const int32_t* nk_all = mm.counts + nrows * K;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int n = min((int)__hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), heavy_cap);
#ifdef MVHDP_REFRESH_PRINT
    if (blockIdx.x == 0 && lane == 0) printf("[refresher] %d heavy rows\n", n);
#endif
    for (int pass = 0; pass < 1000000; pass++) {
        for (int i = blockIdx.x; i < n; i += gridDim.x) {
            const int64_t row = heavy_list[i];
            int m = 0;
            while (m + 1 < mm.M && row >= mm.rowbase[m + 1]) m++;
            const int32_t* cnt = mm.counts + row * K;
            const int32_t* nk = nk_all + (int64_t)m * K;
            const double* al = mm.alpha + (int64_t)m * (K + 1);
            const double beta = mm.beta[m], beta_sum = mm.beta_sum[m], gamma = mm.gamma[m];
            for (int k = lane; k < K; k += WAVE) {
                const int c = __hip_atomic_load(&cnt[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (past this CU's L1: the samplers' atomics land at the memory side)
                t[K + k] = mm.inactive[k] ? 0.0 : gamma * al[k] * (((double)c + beta) / ((double)nk[k] + beta_sum));   // PTM:2670-2678
            }
            tree_from_leaves<true>(mm, row, t, lane, write_full);
        }
#ifdef MVHDP_REFRESH_SLEEP_ONLY      /* experiment: idle waves that touch no memory, for 3 ms */
        if (__builtin_amdgcn_s_memrealtime() - t0 > 300000ull) break;
        __builtin_amdgcn_s_sleep(127);
        continue;
#endif
        if (n <= (int)blockIdx.x) break;                                      // (a block without a row of its own has nothing to do)
        if (__hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) break;     // 2 s of the 100 MHz counter
    }
}

hipError_t mvhdp_launch_heavy_refresh(const MvModel& mm, const int32_t* heavy_list, const unsigned int* ctl, int heavy_cap, int blocks, hipStream_t s)
{
    hipLaunchKernelGGL(heavy_refresh_kernel, dim3(blocks), dim3(64), (size_t)2 * mm.K * sizeof(double), s, mm, heavy_list, ctl, heavy_cap, getenv("MVHDP_REFRESH_FULL") != nullptr);
    return hipGetLastError();
}

__global__ void set_u32_kernel(unsigned int* p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
hipError_t mvhdp_launch_set_u32(unsigned int* p, unsigned int v, hipStream_t s)
{
    hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, s, p, v);
    return hipGetLastError();
}

hipError_t mvhdp_launch_live_rows_prepare(const MvModel& mm, bool from_mirror, bool with_heavy_trees, int32_t* heavy_list, unsigned int* heavy_ctl, int heavy_cap, int batch_cells, hipStream_t s)
{
    const int64_t nrows = mm.rowbase[mm.M];
    hipLaunchKernelGGL(live_coef_kernel, dim3(mm.M), dim3(64), 0, s, mm);
    if (nrows <= 0) return hipGetLastError();
    const int grid = (int)(nrows < 65536 ? nrows : 65536);
    if (heavy_ctl) { hipError_t e = hipMemsetAsync(heavy_ctl, 0, 2 * sizeof(unsigned int), s); if (e != hipSuccess) return e; }     // rows listed, stop
    if (mm.K > 512) hipLaunchKernelGGL(live_rows_prepare_kernel<8>, dim3(grid), dim3(64), 0, s, mm, from_mirror, heavy_list, heavy_ctl, heavy_cap, batch_cells);
    else hipLaunchKernelGGL(live_rows_prepare_kernel<4>, dim3(grid), dim3(64), 0, s, mm, from_mirror, heavy_list, heavy_ctl, heavy_cap, batch_cells);
    // the HEAVY words (more than 65534 tokens: a few hundred rows at most) keep a stored tree, built from the 32-bit table where their
    // counts live; the flags are those the pass above has just written (or kept)
    if (with_heavy_trees)
        hipLaunchKernelGGL(build_trees_kernel<4>, dim3(grid), dim3(64), (size_t)2 * mm.K * sizeof(double), s, mm, false, false, (int64_t)0, nrows, false,
                           (unsigned long long*)nullptr, true, true);
    return hipGetLastError();
}

// tokensPerTopic part of apply_delta alone (M*K words): UPD:209-218
__global__ __launch_bounds__(256) void apply_nk_kernel(int32_t* counts_nk, int32_t* delta_nk, int n, unsigned long long* negatives)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int d = delta_nk[i];
    if (d) { const int c = counts_nk[i] + d; counts_nk[i] = c; delta_nk[i] = 0; if (c < 0 && negatives) atomicAdd(negatives, 1ull); }
}

hipError_t mvhdp_launch_apply_nk(const MvModel& mm, unsigned long long* negatives, hipStream_t s)
{
    const int n = mm.M * mm.K;
    const int64_t off = mm.rowbase[mm.M] * mm.K;
    hipLaunchKernelGGL(apply_nk_kernel, dim3((n + 255) / 256), dim3(256), 0, s, mm.counts + off, mm.delta + off, n, negatives);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// draw_p: WRK:327-337.  MALLET Randoms.nextBeta restated over a per-(doc,pair)
// Philox uniform stream (uniform n of the pair = word pair (n&1) of Philox
// counter (n>>1, 0x100+pair, doc, sweep)).  One thread per entity.
// ---------------------------------------------------------------------------
struct PStream {
    uint32_t c1, c2, c3, k0, k1, n;
    int have_gauss; double next_gauss;
    __device__ double uniform()
    {
        uint32_t x[4];
        philox4x32_10(n >> 1, c1, c2, c3, k0, k1, x);
        double u = (n & 1) ? bits_to_unit(x[2], x[3]) : bits_to_unit(x[0], x[1]);
        n++;
        return u;
    }
    __device__ double gaussian()
    {
        if (!have_gauss) {
            double v1 = uniform(), v2 = uniform();
            double x1 = sqrt(-2 * log(v1)) * cos(2 * M_PI * v2);
            double x2 = sqrt(-2 * log(v1)) * sin(2 * M_PI * v2);
            next_gauss = x2; have_gauss = 1;
            return x1;
        }
        have_gauss = 0;
        return next_gauss;
    }
    __device__ double beta(double alpha, double beta_)
    {
        if (alpha == 1 && beta_ == 1) return uniform();
        if (alpha >= 1 && beta_ >= 1) {
            double A = alpha - 1, B = beta_ - 1, C = A + B, L = C * log(C), mu = A / C, sigma = 0.5 / sqrt(C);
            double y = gaussian(), x = sigma * y + mu;
            while (x < 0 || x > 1) { y = gaussian(); x = sigma * y + mu; }
            double u = uniform();
            // with beta==1 the B*log((1-x)/B) term is NaN and the comparison false (reference quirk, kept)
            while (log(u) >= A * log(x / A) + B * log((1 - x) / B) + L + 0.5 * y * y) {
                y = gaussian(); x = sigma * y + mu;
                while (x < 0 || x > 1) { y = gaussian(); x = sigma * y + mu; }
                u = uniform();
            }
            return x;
        }
        double v1 = pow(uniform(), 1 / alpha), v2 = pow(uniform(), 1 / beta_);
        while (v1 + v2 > 1) { v1 = pow(uniform(), 1 / alpha); v2 = pow(uniform(), 1 / beta_); }
        return v1 / (v1 + v2);
    }
};

__device__ __forceinline__ double java_round_div1000(double b)
{
    // (double) Math.round(1000 * b) / (double) 1000   WRK:333
    double x = 1000 * b;
    double f = floor(x);
    if (x - f >= 0.5) f += 1.0;
    return (double)(long long)f / (double)1000;
}

__global__ __launch_bounds__(256) void draw_p_kernel(MvModel mm, uint32_t sweep_idx, uint32_t seed_lo, uint32_t seed_hi)
{
    const int M = mm.M;
    int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= mm.D) return;
    int64_t dg = mm.doc_id_base + d;
    double* p = mm.p + d * M * M;
    for (int m = 0; m < M; m++) {
        for (int j = m; j < M; j++) {
            double pRand;
            if (m == j) pRand = 1.0;
            else if (mm.p_a[m][j] == 0) pRand = 0;
            else {
                PStream s;
                s.c1 = 0x100u + (uint32_t)(m * M + j); s.c2 = (uint32_t)dg; s.c3 = sweep_idx;
                s.k0 = seed_lo; s.k1 = seed_hi ^ (uint32_t)((unsigned long long)dg >> 32);
                s.n = 0; s.have_gauss = 0; s.next_gauss = 0;
                pRand = java_round_div1000(s.beta(mm.p_a[m][j], mm.p_b[m][j]));
            }
            p[m * M + j] = (j != 0 && mm.beta[j] == 0.0001) ? 0 : pRand;   // WRK:335
            p[j * M + m] = (m != 0 && mm.beta[m] == 0.0001) ? 0 : pRand;   // WRK:336
        }
    }
}

hipError_t mvhdp_launch_draw_p(const MvModel& mm, uint32_t sweep_idx, uint32_t seed_lo, uint32_t seed_hi, hipStream_t s)
{
    if (mm.M <= 1 || mm.D == 0) return hipSuccess;
    int grid = (int)((mm.D + 255) / 256);
    hipLaunchKernelGGL(draw_p_kernel, dim3(grid), dim3(256), 0, s, mm, sweep_idx, seed_lo, seed_hi);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// The sweep.  Per wave (one entity at a time), LDS holds the "dense index" of
// WRK:376-391 in slot form:
//   bitmap[ceil(K/32)]  topics present in the entity (any view) at entry
//   prefix[...]         exclusive popcount prefix -> slot of a topic
//   sk[S]               slot -> topic, ascending; sign bit = removed (WRK:451-468)
//   sn[M][S]            localTopicCounts of the listed topics
//   soth[S], sden[S]    per-view totalMassOtherModalities (WRK:399-410) and n_k+betaSum
//   scum[S]             topicDocWordMasses (WRK:511)
// Slots are never moved: a removed topic keeps its slot with a zero term, which
// leaves every partial sum bit-identical to the compacted list (x + 0.0 == x).
// Q1/Q2: the list never grows during a visit, exactly as in the reference.
// ---------------------------------------------------------------------------
size_t mvhdp_sweep_wave_bytes(int M, int S_cap)
{
    size_t b = 64 * 4 /*bitmap*/ + 64 * 4 /*prefix*/ + 64 * 4 /*bitmap of the new assignments*/ + 16 * 4 /*wlen + pad*/;
    b += (size_t)S_cap * 4;             // sk
    b += (size_t)M * S_cap * 4;         // sn
    b = (b + 7) & ~(size_t)7;
    b += (size_t)3 * S_cap * 8;         // soth, sden, scum
    return (b + 15) & ~(size_t)15;
}

template <bool DEBUG>
__global__ __launch_bounds__(256) void sweep_kernel(MvModel mm, SweepLaunch sl)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = uniform_i(threadIdx.x >> 6);
    const int K = mm.K, M = mm.M, S = sl.S_cap;
    const int NW = (K + 31) >> 5;
    const bool exact_only = (sl.flags & MVHDP_SWEEP_EXACT_CHAIN) != 0;

    // [M*K] n_k deltas of this block, privatised in LDS -- or none, the deltas going straight to the delta buffer,
    // when M*K is too large for that (sl.nk_global); then [MVHDP_HIST_BINS] tokens by topic-list size class
    const int nkd_len = sl.nk_global ? 0 : M * K;
    int* nkd = (int*)smem;
    unsigned int* hist_s = (unsigned int*)(nkd + nkd_len);
    unsigned int* ent_s = hist_s + MVHDP_HIST_BINS;         // [MVHDP_ENT_BINS] entities by the kernel class of their NEW topic list
    for (int i = threadIdx.x; i < nkd_len + MVHDP_HIST_BINS + MVHDP_ENT_BINS; i += blockDim.x) nkd[i] = 0;
    int32_t* const dnk_g = mm.delta + mm.rowbase[M] * K;    // n_k part of the delta buffer
    __syncthreads();

    unsigned char* wb = smem + sl.block_shared_bytes + (size_t)wave * sl.wave_bytes;
    uint32_t* bitmap = (uint32_t*)wb;
    uint32_t* prefix = bitmap + 64;
    uint32_t* bitmap2 = prefix + 64;                        // topics of the entity's NEW assignments (MvModel::nslots)
    int* wlen = (int*)(bitmap2 + 64);                       // [8] + pad
    int* sk = wlen + 16;
    int* sn = sk + S;
    size_t off = (size_t)(64 + 64 + 64 + 16 + S + M * S) * 4;
    off = (off + 7) & ~(size_t)7;
    double* soth = (double*)(wb + off);
    double* sden = soth + S;
    double* scum = sden + S;

    // no __restrict__: with MVHDP_SWEEP_LIVE the atomics below update this very array (mm.delta == mm.counts)
    const int32_t* nwk = mm.counts;
    const int32_t* nk_all = mm.counts + mm.rowbase[M] * K;
    int32_t* dnwk = mm.delta;

    unsigned int n_tok = 0, n_chg = 0, c_new = 0, c_doc = 0, c_tree = 0, n_oov = 0, n_abort = 0, n_fb = 0;

    // work queue: each wave pulls MVHDP_DOC_BATCH entities at a time from one global head
    const long long q_n1 = sl.q_list_count ? (long long)*sl.q_list_count : 0;
    const long long q_total = q_n1 + sl.q_order_count;
    // (the last pulls of the queue take one entity at a time: the launch ends within one entity's time of its last pull)
    const long long q_single = q_total - 2LL * gridDim.x * (blockDim.x >> 6);
    long long q_seen = 0;
    for (;;) {
      const long long batch = (q_seen >= q_single) ? 1 : MVHDP_DOC_BATCH;
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
        int S_used;
        {
            uint32_t wbits = (lane < NW) ? bitmap[lane] : 0u;
            int cnt = __popc(wbits);
            int incl = wave_incl_scan_i(cnt, lane);
            prefix[lane] = (uint32_t)(incl - cnt);
            S_used = bcast_i(incl, 63);
        }
        LDS_FENCE();
        for (int k0 = 0; k0 < K; k0 += WAVE) {
            int k = k0 + lane;
            if (k < K) {
                uint32_t w = bitmap[k >> 5];
                if ((w >> (k & 31)) & 1u) sk[prefix[k >> 5] + __popc(w & ((1u << (k & 31)) - 1u))] = k;
            }
        }
        for (int m = 0; m < M; m++)
            for (int i = lane; i < S_used; i += WAVE) sn[m * S + i] = 0;
        LDS_FENCE();
        for (int m = 0; m < M; m++) {
            const int64_t b = mm.doc_off[m][d], e = mm.doc_off[m][d + 1];
            for (int64_t i = b + lane; i < e; i += WAVE) {
                int zz = mm.z[m][i];
                if (zz >= 0) {
                    uint32_t w = bitmap[zz >> 5];
                    int slot = prefix[zz >> 5] + __popc(w & ((1u << (zz & 31)) - 1u));
                    atomicAdd(&sn[m * S + slot], 1);                       // WRK:357
                }
            }
        }
        LDS_FENCE();

        const double* pd = (M > 1) ? (mm.p + d * M * M) : nullptr;        // WRK:327-337 (draw_p_kernel / host override)
        bool aborted = false;

        for (int m = 0; m < M && !aborted; m++) {                         // WRK:393
            const int lenm = uniform_i(wlen[m]);
            if (lenm == 0) continue;
            const double beta_m = mm.beta[m];
            const double scale_m = (double)lenm + mm.gamma[m] * mm.alpha_sum[m];
            const double p_mm = pd ? pd[m * M + m] : 1.0;
            const int32_t* nk = nk_all + (int64_t)m * K;

            // WRK:395-410 totalMassOtherModalities for the listed topics (frozen for this view, Q3)
            for (int i = lane; i < S_used; i += WAVE) {
                int k = sk[i] & 0x7fffffff;
                double acc = 0.0;
                for (int j = 0; j < M; j++) {
                    int lj = wlen[j];
                    if (j != m && lj != 0) {
                        acc += pd[m * M + j] * ((double)sn[j * S + i] + mm.gamma[j] * mm.alpha[(int64_t)j * (K + 1) + k])
                               / ((double)lj + mm.gamma[j] * mm.alpha_sum[j]);
                    }
                }
                soth[i] = acc * scale_m;
                sden[i] = (double)nk[k] + mm.beta_sum[m];                   // tokensPerTopic + betaSum  WRK:507
            }
            // WRK:413-418 newTopicMassAllModalities
            double newAll = 0.0;
            for (int j = 0; j < M; j++) {
                double pmj = pd ? pd[m * M + j] : 1.0;
                newAll += pmj * (mm.gamma[j] * mm.alpha[(int64_t)j * (K + 1) + K]) / ((double)wlen[j] + mm.gamma[j] * mm.alpha_sum[j]);
            }
            newAll = newAll * scale_m;
            const double newMass = (mm.first_inactive < 0) ? 0.0 : newAll / (double)K;   // WRK:515
            LDS_FENCE();

            const int64_t base = mm.doc_off[m][d];
            const int64_t row0 = mm.rowbase[m];
            const int Vm = mm.V[m];

            for (int c0 = 0; c0 < lenm && !aborted; c0 += WAVE) {
                // one lane per token of the chunk: token id, old topic, slot, RNG, tree root
                const int ti = c0 + lane;
                const bool tvalid = ti < lenm;
                int w_l = tvalid ? mm.tok[m][base + ti] : 0;
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
                const bool in_vocab = tvalid && w_l >= 0 && w_l < Vm;
                double root_l = in_vocab ? mm.root[row0 + w_l] : 0.0;
                int znew_l = z_l;
                const int nt = min(WAVE, lenm - c0);

                for (int t = 0; t < nt; t++) {                              // WRK:425
                    const int w = bcast_i(w_l, t);
                    if (w < 0 || w >= Vm) { n_oov++; continue; }            // WRK:427-428
                    const int so = bcast_i(so_l, t);
                    const double u1 = bcast_d(u1_l, t), u2 = bcast_d(u2_l, t);
                    const double root = bcast_d(root_l, t);
                    const int64_t row = row0 + w;
                    const int32_t* __restrict__ cnt = nwk + row * K;

                    // WRK:434-468 decrement the local count; drop the topic from the list when it is gone from all views
                    if (so >= 0) {
                        int c = sn[m * S + so] - 1;
                        if (lane == 0) sn[m * S + so] = c;
                        LDS_FENCE();
                        if (c == 0) {
                            bool gone = true;
                            for (int j = 0; j < M; j++) if (sn[j * S + so] != 0) gone = false;
                            if (gone && lane == 0) sk[so] |= 0x80000000;
                            LDS_FENCE();
                        }
                    }

                    // WRK:496-513 topicDocWordMasses.  pass 0: wave prefix scan (certified below);
                    // pass 1: the reference's sequential left-to-right sum.
                    double mass = 0.0, s0 = 0.0, s1 = 0.0, total = 0.0;
                    int branch = 0;          // 0 new-topic, 1 doc, 2 tree
                    int slot_new = -1;
                    for (int pass = exact_only ? 1 : 0; pass < 2; pass++) {
                        if (pass == 0) {
                            double carry = 0.0;
                            for (int r0 = 0; r0 < S_used; r0 += WAVE) {
                                const int i = r0 + lane;
                                double term = 0.0;
                                if (i < S_used) {
                                    int k = sk[i];
                                    if (k >= 0) {
                                        double p_wt = div_inrange((double)cnt[k] + beta_m, sden[i]);          // WRK:507
                                        term = (p_mm * (double)sn[m * S + i] + soth[i]) * p_wt;            // WRK:509
                                    }
                                }
                                double cum = carry + wave_incl_scan_d_dpp(term);
                                if (i < S_used) scum[i] = cum;
                                carry = bcast_d(cum, 63);
                            }
                            mass = carry;
                        } else {
                            for (int i = lane; i < S_used; i += WAVE) {
                                int k = sk[i];
                                double term = 0.0;
                                if (k >= 0) {
                                    double p_wt = div_inrange((double)cnt[k] + beta_m, sden[i]);
                                    term = (p_mm * (double)sn[m * S + i] + soth[i]) * p_wt;
                                }
                                scum[i] = term;
                            }
                            LDS_FENCE();
                            double c = 0.0;
                            for (int i = 0; i < S_used; i++) {               // WRK:501-513, dense order
                                c += scum[i];
                                if (lane == 0) scum[i] = c;
                            }
                            mass = c;
                        }
                        LDS_FENCE();

                        total = newMass + mass + root;                       // WRK:519
                        s0 = u1 * total;
                        // Certified scan: any summation order of the same non-negative terms differs
                        // from the sequential one by < (2n) ulp-units of the total; if no comparison
                        // below is closer than tol the decisions equal the reference's bit for bit.
                        const double tol = (pass == 0) ? total * (double)(4 * S_used + 16) * 0x1.0p-53 : -1.0;
                        bool unsafe = false;
                        if (s0 < newMass) {                                  // WRK:522
                            branch = 0;
                            if (fabs(s0 - newMass) <= tol) unsafe = true;
                        } else {
                            if (newMass != 0.0 && fabs(s0 - newMass) <= tol) unsafe = true;
                            s1 = s0 - newMass;                               // WRK:528
                            if (fabs(s1 - mass) <= tol) unsafe = true;
                            if (s1 < mass) {                                 // WRK:529
                                branch = 1;
                                slot_new = -1;
                                for (int r0 = 0; r0 < S_used; r0 += WAVE) {  // WRK:531 lower_bound over the live list
                                    const int i = r0 + lane;
                                    bool live = (i < S_used) && (sk[i] >= 0);
                                    double cv = live ? scum[i] : 0.0;
                                    if (__ballot(live && fabs(cv - s1) <= tol)) unsafe = true;
                                    unsigned long long hit = __ballot(live && cv >= s1);
                                    if (hit && slot_new < 0) slot_new = r0 + (int)__builtin_ctzll(hit);
                                }
                            } else {
                                branch = 2;
                            }
                        }
                        if (!(pass == 0 && unsafe)) break;
                        n_fb++;
                    }

                    if (DEBUG) {
                        if (sl.tok_dbg[m] && lane == 0) {
                            double* g = sl.tok_dbg[m] + (base + c0 + t) * 4;
                            g[0] = newMass; g[1] = mass; g[2] = root; g[3] = s0;
                        }
                        for (int q = 0; q < sl.n_trace; q++) {
                            if (sl.trace_doc[q] == d && sl.trace_view[q] == m && sl.trace_pos[q] == c0 + t) {
                                double* out = sl.trace_out + (int64_t)q * (K + 1);
                                const double* tr = mm.trees + row * 2 * K;
                                for (int k = lane; k < K; k += WAVE) out[k] = tr[K + k] / total;
                                __threadfence();
                                if (lane == 0) {
                                    double prev = 0.0;
                                    for (int i = 0; i < S_used; i++) {
                                        if (sk[i] >= 0) { out[sk[i]] += (scum[i] - prev) / total; }
                                        prev = scum[i];
                                    }
                                    out[K] = newMass / total;
                                }
                            }
                        }
                    }

                    int znew;
                    if (branch == 0) {                                       // WRK:523-526
                        c_new++;
                        znew = mm.first_inactive;
                    } else if (branch == 1) {                                // WRK:530-531
                        c_doc++;
                        if (slot_new < 0) { aborted = true; break; }         // lower_bound == -1 -> exception, Q11
                        znew = uniform_i(sk[slot_new]);
                    } else {                                                 // WRK:533-535
                        c_tree++;
                        znew = tree_sample(mm.trees + row * 2 * K, K, u2, root, lane);
                    }
                    if (znew < 0) znew = K - 1;                              // WRK:549-552
                    znew = uniform_i(znew);

                    // WRK:557-560
                    if (lane == t) znew_l = znew;
                    if (branch != 1) {
                        uint32_t wbit = bitmap[znew >> 5];
                        slot_new = ((wbit >> (znew & 31)) & 1u) ? (int)(prefix[znew >> 5] + __popc(wbit & ((1u << (znew & 31)) - 1u))) : -1;
                    }
                    if (slot_new >= 0 && lane == 0) sn[m * S + slot_new] += 1;
                    LDS_FENCE();
                    n_tok++;

                }

                // WRK:587-589 + UPD:197-218 for the whole chunk at once (lane t owns token t): wave-wide atomics on
                // the delta rows and on the block's n_k table, issued after the token loop so that no token waits on them
                {
                    const bool chg = tvalid && (w_l >= 0) && (w_l < Vm) && (znew_l != z_l) && !(sl.flags & MVHDP_SWEEP_FROZEN);
                    n_chg += (unsigned int)__popcll(__builtin_amdgcn_ballot_w64(chg));
                    if (chg) {
                        const int64_t rowK = (row0 + w_l) * K;
                        if (z_l >= 0) {
                            __hip_atomic_fetch_add(&dnwk[rowK + z_l], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (sl.nk_global) __hip_atomic_fetch_add(&dnk_g[m * K + z_l], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            else __hip_atomic_fetch_add(&nkd[m * K + z_l], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        __hip_atomic_fetch_add(&dnwk[rowK + znew_l], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (sl.nk_global) __hip_atomic_fetch_add(&dnk_g[m * K + znew_l], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else __hip_atomic_fetch_add(&nkd[m * K + znew_l], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (mm.first_inactive >= 0 && mm.inactive[znew_l]) {          // UPD:263
                            long long key = (long long)MVHDP_ACT_KEY(dg, m, ti, znew_l);
                            atomicMin(sl.act_key, key);
                        }
                    }
                }
                if (tvalid) mm.z[m][base + ti] = znew_l;                     // coalesced write-back of the chunk
                if (sl.flags & MVHDP_SL_STRICT_LIVE) {                       // (diagnostics: mvhdp_tuning.single_wave)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                if (tvalid && znew_l >= 0) atomicOr(&bitmap2[znew_l >> 5], 1u << (znew_l & 31));
            }
        }
        if (aborted) n_abort++;
        LDS_FENCE();
        {   // the entity's topic list at its NEXT visit: distinct topics of the assignments just written
            int c2 = __popc(lane < NW ? bitmap2[lane] : 0u);
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) c2 += __shfl_xor(c2, sft, WAVE);
            if (lane == 0) {
                mm.nslots[d] = aborted ? (uint16_t)MVHDP_NSLOTS_UNKNOWN : (uint16_t)c2;
                if (sl.slot_hist) {
                    if (c2 > 0) atomicAdd(&hist_s[min((c2 + 63) >> 6, MVHDP_HIST_BINS) - 1], (unsigned int)doc_tokens);
                    atomicAdd(&ent_s[aborted ? MVHDP_N_CLASSES : mvhdp_class_of(c2, longest_view > 65535)], 1u);
                }
            }
        }
        LDS_FENCE();
      }
    }

    __syncthreads();
    for (int i = threadIdx.x; i < nkd_len; i += blockDim.x)
        if (nkd[i]) atomicAdd(&dnk_g[i], nkd[i]);
    if (sl.slot_hist && threadIdx.x < MVHDP_HIST_BINS + MVHDP_ENT_BINS && hist_s[threadIdx.x]) atomicAdd(&sl.slot_hist[threadIdx.x], (unsigned long long)hist_s[threadIdx.x]);
    if (lane == 0) {
        if (n_tok) atomicAdd(&sl.stats[ST_TOKENS], (unsigned long long)n_tok);
        if (n_chg) atomicAdd(&sl.stats[ST_CHANGED], (unsigned long long)n_chg);
        if (c_new) atomicAdd(&sl.stats[ST_NEW], (unsigned long long)c_new);
        if (c_doc) atomicAdd(&sl.stats[ST_DOC], (unsigned long long)c_doc);
        if (c_tree) atomicAdd(&sl.stats[ST_TREE], (unsigned long long)c_tree);
        if (n_oov) atomicAdd(&sl.stats[ST_OOV], (unsigned long long)n_oov);
        if (n_abort) atomicAdd(&sl.stats[ST_ABORT], (unsigned long long)n_abort);
        if (n_fb) atomicAdd(&sl.stats[ST_FALLBACK], (unsigned long long)n_fb);
    }
}

hipError_t mvhdp_sweep_set_max_lds(size_t bytes)
{
    hipError_t e = hipFuncSetAttribute((const void*)sweep_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)sweep_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

int mvhdp_sweep_generic_occupancy(bool debug, int block_threads, size_t lds_bytes)
{
    hipFuncAttributes a;
    const void* f = debug ? (const void*)sweep_kernel<true> : (const void*)sweep_kernel<false>;
    if (hipFuncGetAttributes(&a, f) != hipSuccess) return 1;
    int regs = (a.numRegs + 7) / 8 * 8;
    int waves_simd = regs > 0 ? 512 / regs : 8;
    if (waves_simd > 8) waves_simd = 8;
    if (waves_simd < 1) waves_simd = 1;
    int wpb = block_threads / 64;
    int b = waves_simd * 4 / wpb;
    int by_lds = (int)((160 * 1024) / (lds_bytes > 0 ? lds_bytes : 1));
    if (by_lds < b) b = by_lds;
    if (32 / wpb < b) b = 32 / wpb;
    return b < 1 ? 1 : b;
}

int mvhdp_sweep_generic_regs(bool debug)
{
    hipFuncAttributes a;
    const void* f = debug ? (const void*)sweep_kernel<true> : (const void*)sweep_kernel<false>;
    if (hipFuncGetAttributes(&a, f) != hipSuccess) return 128;
    return a.numRegs;
}

hipError_t mvhdp_launch_sweep(const MvModel& mm, const SweepLaunch& sl, int grid_blocks, bool debug, hipStream_t s)
{
    size_t lds = sl.block_shared_bytes + (size_t)sl.waves_per_block * sl.wave_bytes;
    dim3 block(64 * sl.waves_per_block);
    if (debug) hipLaunchKernelGGL(sweep_kernel<true>, dim3(grid_blocks), block, lds, s, mm, sl);
    else       hipLaunchKernelGGL(sweep_kernel<false>, dim3(grid_blocks), block, lds, s, mm, sl);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// apply_delta: counts += delta; delta = 0 (the updater's effect, UPD:197-218,
// after the optional cross-GPU all-reduce of the delta buffer).
// ---------------------------------------------------------------------------
// delta16 (or null): the n_wk deltas the sweep kept in 16-bit cells biased by 0x8000 (SweepLaunch::delta16), n16 cells: added on top and
// set back to the bias.
__global__ __launch_bounds__(256) void apply_delta_kernel(int32_t* counts, int32_t* delta, int64_t n, unsigned long long* stats, uint16_t* delta16, int64_t n16)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int neg = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        int dl = delta[i];
        if (delta16 && i < n16) {
            const int v = (int)delta16[i];
            if (v != 0x8000) { dl += v - 0x8000; delta16[i] = (uint16_t)0x8000; if (!dl) delta[i] = 0; }
        }
        if (dl) {
            int c = counts[i] + dl;
            counts[i] = c;
            delta[i] = 0;
            if (c < 0) neg++;                              // UPD:202-215 logs this; here it is a hard error
        }
    }
    if (neg) atomicAdd(&stats[ST_NEGATIVE], (unsigned long long)neg);
}

// One launch instead of a handful of fills: the counters a sweep (n_stats words), or a segment of one, starts from.
__global__ __launch_bounds__(256) void ctl_reset_kernel(unsigned long long* stats, int n_stats, long long* act_key, unsigned long long* meta, int n_meta,
                                                        unsigned int* class_counts, unsigned long long* qheads)
{
    const int t = threadIdx.x;
    if (stats) for (int i = t; i < n_stats; i += blockDim.x) stats[i] = 0ull;
    if (meta) for (int i = t; i < n_meta; i += blockDim.x) meta[i] = 0ull;
    if (class_counts && t < MVHDP_N_CLASSES) class_counts[t] = 0u;
    if (qheads && t < 8) qheads[t] = 0ull;
    if (act_key && t == 0) *act_key = 0x7fffffffffffffffLL;      // MVHDP_ACT_KEY_NONE
}

hipError_t mvhdp_launch_ctl_reset(unsigned long long* stats, int n_stats, long long* act_key, unsigned long long* meta, int n_meta,
                                  unsigned int* class_counts, unsigned long long* qheads, hipStream_t s)
{
    hipLaunchKernelGGL(ctl_reset_kernel, dim3(1), dim3(256), 0, s, stats, n_stats, act_key, meta, n_meta, class_counts, qheads);
    return hipGetLastError();
}

// Holds a stream for about `microseconds` (one wave, s_sleep against the constant 100 MHz counter; always terminates).  An experiment
// (PlanTuning::fork_delay_us, off by default): put between the fork event and the primary kernel it lets the wider class kernels --
// which wait for that event on side streams, ~16 us longer than the next launch on the same stream takes -- become resident first
// instead of when the primary's first blocks drain.  In the kernel trace the segment shortens by 0.14 ms; un-profiled the sweep does
// not (segmented 34.76 vs 34.66 ms) and a deferred sweep whose wider class is still populous loses 1.5 ms (gpurun_out/r3_m6_*).
__global__ __launch_bounds__(64) void delay_kernel(unsigned int ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 100000 && __builtin_amdgcn_s_memrealtime() - t0 < ticks; i++) __builtin_amdgcn_s_sleep(16);
}

hipError_t mvhdp_launch_delay(int microseconds, hipStream_t s)
{
    if (microseconds <= 0) return hipSuccess;
    hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, s, (unsigned int)microseconds * 100u);
    return hipGetLastError();
}

hipError_t mvhdp_launch_apply_delta(const MvModel& mm, unsigned long long* stats, hipStream_t s, bool with_delta16)
{
    int64_t n = mm.rowbase[mm.M] * mm.K + (int64_t)mm.M * mm.K;
    int grid = (int)((n + 255) / 256);
    if (grid > 8192) grid = 8192;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(apply_delta_kernel, dim3(grid), dim3(256), 0, s, mm.counts, mm.delta, n, stats, with_delta16 ? mm.delta16 : nullptr, mm.rowbase[mm.M] * mm.K);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// MVHDP_SWEEP_LIVE helpers.  A live sweep updates `counts` in place (UPD:197-218 applied while the
// workers sample, as the reference's updater threads do).  For document shards on several GPUs the
// host still needs "what did THIS shard change": live_begin stores -counts in the delta buffer,
// live_end turns it into (after - before) and puts the sweep-start snapshot back into counts, so the
// all-reduce + mvhdp_apply_delta sequence of the deferred mode applies unchanged (AD-LDA: each
// replica is live for its own documents and one sweep stale for the others).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void live_begin_kernel(const int32_t* __restrict__ counts, int32_t* __restrict__ delta, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) delta[i] = -counts[i];
}

__global__ __launch_bounds__(256) void live_end_kernel(int32_t* __restrict__ counts, int32_t* __restrict__ delta, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int c = counts[i], dl = delta[i] + c;          // after - before
        delta[i] = dl;
        counts[i] = c - dl;                                  // the sweep-start snapshot again
    }
}

// UPD:202-215 logs a negative count; a live sweep has no apply pass to notice one, so it is looked for afterwards
__global__ __launch_bounds__(256) void check_negative_kernel(const int32_t* __restrict__ counts, int64_t n, unsigned long long* stats)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int neg = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) neg += counts[i] < 0;
    if (neg) atomicAdd(&stats[ST_NEGATIVE], (unsigned long long)neg);
}

hipError_t mvhdp_launch_live_helper(const MvModel& mm, int which, unsigned long long* stats, hipStream_t s)
{
    const int64_t n = mm.rowbase[mm.M] * mm.K + (int64_t)mm.M * mm.K;
    int grid = (int)((n + 255) / 256);
    if (grid > 8192) grid = 8192;
    if (grid < 1) grid = 1;
    if (which == 0) hipLaunchKernelGGL(live_begin_kernel, dim3(grid), dim3(256), 0, s, mm.counts, mm.delta, n);
    else if (which == 1) hipLaunchKernelGGL(live_end_kernel, dim3(grid), dim3(256), 0, s, mm.counts, mm.delta, n);
    else hipLaunchKernelGGL(check_negative_kernel, dim3(grid), dim3(256), 0, s, mm.counts, n, stats);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// doc_topic_hist: topicDocCounts[m][k][n_dk] and docLengthCounts[m][len]
// (PTM:620-651) recomputed from z.  One wave per entity.  Bucket 0 is filled
// afterwards from the totals (docs with the view minus docs holding the topic).
// ---------------------------------------------------------------------------
// The low buckets take nearly every increment (a topic an entity holds, it mostly holds once or a few times), so each
// workgroup keeps its own copy of buckets [0, CL) and of the holder counts in LDS and adds them to the global arrays once at
// the end: the global atomics no longer queue on K hot words per bucket (25.7 -> a few ms per view at C4).
__global__ __launch_bounds__(256) void doc_topic_hist_kernel(MvModel mm, int m, int32_t* hist, int32_t hist_len,
                                                             int32_t* doc_len_counts, int32_t len_len, int32_t* docs_with_view, int CL)
{
    extern __shared__ int ldk[];                           // [waves][K] per-entity counts, [K][CL] low buckets, [K] holders, [1] entities
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, K = mm.K, nwaves = blockDim.x >> 6;
    int* my = ldk + wave * K;
    int* low = ldk + nwaves * K;
    int* holders = low + K * CL;
    int* nview = holders + K;
    for (int i = threadIdx.x; i < K * CL + K + 1; i += blockDim.x) low[i] = 0;
    __syncthreads();
    const int64_t wstride = (int64_t)gridDim.x * nwaves;
    for (int64_t d = (int64_t)blockIdx.x * nwaves + wave; d < mm.D; d += wstride) {
        const int64_t b = mm.doc_off[m][d], e = mm.doc_off[m][d + 1];
        if (mm.present[m] ? !mm.present[m][d] : e == b) continue;          // the entity lacks the view (a present view without tokens counts)
        for (int k = lane; k < K; k += WAVE) my[k] = 0;
        LDS_FENCE();
        for (int64_t i = b + lane; i < e; i += WAVE) { int zz = mm.z[m][i]; if (zz >= 0) atomicAdd(&my[zz], 1); }
        LDS_FENCE();
        for (int k = lane; k < K; k += WAVE) {
            int c = my[k];
            if (c > 0 && hist) {
                if (c < CL) atomicAdd(&low[k * CL + c], 1);
                else if (c < hist_len) atomicAdd(&hist[(int64_t)k * hist_len + c], 1);
                atomicAdd(&holders[k], 1);                 // entities holding the topic at all (for bucket 0)
            }
        }
        if (lane == 0) {
            atomicAdd(nview, 1);
            if (doc_len_counts && e - b < len_len) atomicAdd(&doc_len_counts[e - b], 1);
        }
        LDS_FENCE();
    }
    __syncthreads();
    if (hist) {
        for (int i = threadIdx.x; i < K * CL; i += blockDim.x) {
            const int v = low[i], k = i / CL, c = i - k * CL;
            if (v && c < hist_len) atomicAdd(&hist[(int64_t)k * hist_len + c], v);
        }
        for (int k = threadIdx.x; k < K; k += blockDim.x) if (holders[k]) atomicAdd(&docs_with_view[1 + k], holders[k]);
    }
    if (threadIdx.x == 0 && *nview) atomicAdd(docs_with_view, *nview);
}

// bucket 0 = entities with the view that do not hold the topic (PTM:647-649); entities holding it more than hist_len-1
// times are in no bucket (the caller asked for a shorter histogram) and must not be taken for non-holders
__global__ void hist_bucket0_kernel(int32_t* hist, int32_t hist_len, int K, const int32_t* docs_with_view)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    hist[(int64_t)k * hist_len] = docs_with_view[0] - docs_with_view[1 + k];
}

hipError_t mvhdp_launch_doc_topic_hist(const MvModel& mm, int m, int32_t* hist, int32_t hist_len,
                                       int32_t* doc_len_counts, int32_t len_len, hipStream_t s)
{
    int32_t* dwv = nullptr;                                  // [1 + K]: entities with the view, entities holding topic k
    hipError_t e = hipMalloc(&dwv, (size_t)(1 + mm.K) * sizeof(int32_t));
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(dwv, 0, (size_t)(1 + mm.K) * sizeof(int32_t), s);
    if (e == hipSuccess && hist) e = hipMemsetAsync(hist, 0, (size_t)mm.K * hist_len * sizeof(int32_t), s);
    if (e == hipSuccess && doc_len_counts) e = hipMemsetAsync(doc_len_counts, 0, (size_t)len_len * sizeof(int32_t), s);
    if (e == hipSuccess) {
        int wpb = 4;
        while (wpb > 1 && (size_t)wpb * mm.K * sizeof(int) > 60000) wpb >>= 1;
        int CL = 16384 / mm.K;                               // low buckets kept per workgroup: at most 64 KiB of LDS
        if (CL > 32) CL = 32;
        if (CL < 2) CL = 2;
        if (CL > hist_len) CL = hist_len;
        const size_t lds = ((size_t)wpb * mm.K + (size_t)mm.K * CL + mm.K + 1) * sizeof(int);
        if (lds > 65536) {
            e = hipFuncSetAttribute((const void*)doc_topic_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        int64_t blocks = (mm.D + wpb - 1) / wpb;
        int grid = (int)(blocks < 1024 ? (blocks < 1 ? 1 : blocks) : 1024);
        if (e == hipSuccess) hipLaunchKernelGGL(doc_topic_hist_kernel, dim3(grid), dim3(64 * wpb), lds, s,
                                                mm, m, hist, hist_len, doc_len_counts, len_len, dwv, CL);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess && hist) {
            hipLaunchKernelGGL(hist_bucket0_kernel, dim3((mm.K + 63) / 64), dim3(64), 0, s, hist, hist_len, mm.K, dwv);
            e = hipGetLastError();
        }
    }
    const hipError_t es = hipStreamSynchronize(s);           // a kernel fault surfaces here, not on a later call
    hipFree(dwv);
    return e != hipSuccess ? e : es;
}


// ---------------------------------------------------------------------------
// slot_hist: how many distinct topics each entity holds (over all views), binned by
// ceil(n/64) = 1,2,3,4,>4.  Sizes the register-resident sweep kernel before the first
// sweep (afterwards the sweep kernels keep the histogram current themselves).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void slot_hist_kernel(MvModel mm, unsigned long long* hist)
{
    __shared__ uint32_t bm[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ unsigned int hs[MVHDP_HIST_BINS + MVHDP_ENT_BINS];
    if (threadIdx.x < MVHDP_HIST_BINS + MVHDP_ENT_BINS) hs[threadIdx.x] = 0;
    __syncthreads();
    const int64_t wstride = (int64_t)gridDim.x * 4;
    for (int64_t d = (int64_t)blockIdx.x * 4 + wave; d < mm.D; d += wstride) {
        bm[wave][lane] = 0;
        LDS_FENCE();
        int doc_tokens = 0, longest_view = 0;
        for (int m = 0; m < mm.M; m++) {
            const int64_t b = mm.doc_off[m][d], e = mm.doc_off[m][d + 1];
            doc_tokens += (int)(e - b);
            longest_view = max(longest_view, (int)(e - b));
            for (int64_t i = b + lane; i < e; i += WAVE) {
                int zz = mm.z[m][i];
                if (zz >= 0) atomicOr(&bm[wave][zz >> 5], 1u << (zz & 31));
            }
        }
        LDS_FENCE();
        int cnt = __popc(bm[wave][lane]);
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) cnt += __shfl_xor(cnt, s, WAVE);
        if (lane == 0) {
            mm.nslots[d] = (uint16_t)cnt;
            if (cnt > 0) atomicAdd(&hs[min((cnt + 63) >> 6, MVHDP_HIST_BINS) - 1], (unsigned int)doc_tokens);   // weighted by tokens
            atomicAdd(&hs[MVHDP_HIST_BINS + mvhdp_class_of(cnt, longest_view > 65535)], 1u);
        }
        LDS_FENCE();
    }
    __syncthreads();
    if (threadIdx.x < MVHDP_HIST_BINS + MVHDP_ENT_BINS && hs[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)hs[threadIdx.x]);
}

hipError_t mvhdp_launch_slot_hist(const MvModel& mm, unsigned long long* hist, hipStream_t s)
{
    if (mm.D <= 0) return hipSuccess;
    int64_t blocks = (mm.D + 3) / 4;
    int grid = (int)(blocks < 4096 ? blocks : 4096);
    hipLaunchKernelGGL(slot_hist_kernel, dim3(grid), dim3(256), 0, s, mm, hist);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// route: before a sweep (or a segment of one), the entities that MAY hold more topics than the primary kernel
// variant has slots (those with more tokens than slots: a static prefix of the longest-first order) are listed for
// the narrowest kernel variant that holds their topic list.  The list sizes are known: every sweep kernel leaves
// MvModel::nslots behind for the entities it visited, so this is one 2-byte read per entity, no pass over z.
// The sweep kernels of all classes then run side by side, the widest (longest entities) first.
// One thread per entity; a wave appends its entities of a class with ONE atomic (ballot + rank), so the lists keep
// the longest-first order up to the interleaving of waves.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void route_kernel(MvModel mm, ClassifyArgs ca)
{
    // one atomic per class and BLOCK of 256 entities (per wave it was 0.29 ms for a million entities: 31 k atomics on two words).  Blocks
    // of four waves, not sixteen: with overlapped segments this kernel starts while the previous segment's kernels hold all but one block
    // slot per CU, and a block that needs four waves on every SIMD at once waited for them to drain (1.4 - 5 ms in the kernel trace).
    __shared__ unsigned int wave_cnt[4][MVHDP_N_CLASSES], wave_base[4][MVHDP_N_CLASSES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int c = -1;
    int64_t d = 0;
    if (q < ca.n) {
        const int64_t oq = ca.start + q * ca.stride;
        d = ca.order ? (int64_t)ca.order[oq] : oq;
        const unsigned int ns = mm.nslots[d];
        bool beyond = false;
        if (ca.check_views)
            for (int m = 0; m < mm.M; m++) beyond = beyond || (mm.doc_off[m][d + 1] - mm.doc_off[m][d] > 65535);
        c = (ns == MVHDP_NSLOTS_UNKNOWN) ? -1 : ca.class_map[mvhdp_class_of((int)ns, beyond)];
        if (c < 0) atomicAdd(ca.misrouted, 1ull);              // size not known (the host recounts first) or no kernel for it: the host fails the sweep
    }
    unsigned long long mine[MVHDP_N_CLASSES];
#pragma unroll
    for (int k = 0; k < MVHDP_N_CLASSES; k++) {
        mine[k] = __ballot(c == k);
        if (lane == 0) wave_cnt[wave][k] = (unsigned int)__popcll(mine[k]);
    }
    __syncthreads();
    if (threadIdx.x < MVHDP_N_CLASSES) {                         // thread k: class k -- the block's total, one atomic, the waves' offsets in order
        const int k = threadIdx.x;
        unsigned int tot = 0;
        for (int w = 0; w < nwaves; w++) { wave_base[w][k] = tot; tot += wave_cnt[w][k]; }
        const unsigned int base = tot ? atomicAdd(&ca.counts[k], tot) : 0u;
        for (int w = 0; w < nwaves; w++) wave_base[w][k] += base;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MVHDP_N_CLASSES; k++)
        if (c == k) ca.lists[k][wave_base[wave][k] + __popcll(mine[k] & ((1ull << lane) - 1ull))] = (int32_t)d;
}

hipError_t mvhdp_launch_classify(const MvModel& mm, const ClassifyArgs& ca, hipStream_t s)
{
    if (ca.n <= 0) return hipSuccess;
    const int64_t blocks = (ca.n + 255) / 256;
    hipLaunchKernelGGL(route_kernel, dim3((unsigned int)blocks), dim3(256), 0, s, mm, ca);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// init_from_trees: INF:169-199.  One wave per (view, entity): each token's topic is drawn from its
// type's tree (FTree.sample FT:111-136); out-of-vocabulary tokens get 0 (Java's new int[]).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void init_from_trees_kernel(MvModel mm, uint32_t seed_lo, uint32_t seed_hi)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wstride = (int64_t)gridDim.x * 4;
    for (int64_t d = (int64_t)blockIdx.x * 4 + wave; d < mm.D; d += wstride) {
        const int64_t dg = mm.doc_id_base + d;
        for (int m = 0; m < mm.M; m++) {
            const int64_t b = mm.doc_off[m][d], e = mm.doc_off[m][d + 1];
            for (int64_t i = b + lane; i < e; i += WAVE) {             // one lane per token: 64 descents side by side
                const int type = mm.tok[m][i];
                int topic = 0;                                         // out-of-vocabulary: Java's new int[] (INF:193-199)
                if (type >= 0 && type < mm.V[m]) {
                    uint32_t x[4];
                    philox4x32_10((uint32_t)(i - b), (uint32_t)m, (uint32_t)dg, 0xFFFFFFFFu,
                                  seed_lo, seed_hi ^ (uint32_t)((unsigned long long)dg >> 32), x);
                    topic = dtab_sample(mm, mm.rowbase[m] + type, bits_to_unit(x[0], x[1]));
                }
                mm.z[m][i] = topic;
            }
        }
    }
}

hipError_t mvhdp_launch_init_from_trees(const MvModel& mm, uint32_t seed_lo, uint32_t seed_hi, hipStream_t s)
{
    if (mm.D <= 0) return hipSuccess;
    int64_t blocks = (mm.D + 3) / 4;
    int grid = (int)(blocks < 8192 ? blocks : 8192);
    hipLaunchKernelGGL(init_from_trees_kernel, dim3(grid), dim3(256), 0, s, mm, seed_lo, seed_hi);
    return hipGetLastError();
}
