// live_staleness.hip — how old is what a load returns while other waves update the same table with agent-scope atomics?
//
// The question behind MVHDP_SWEEP_LIVE: the sweep's chunk-end atomics (UPD:197-207) land on the n_wk rows that
// every later token gathers with ordinary loads.  Atomics execute at the memory side; a gather may be served by the CU's vector L1 or
// by its XCD's L2, neither of which another XCD's atomic refreshes.  So how stale is a gathered value, by load flavour and table size?
//
// Method: a table of T lines of 128 bytes; every wave alternates R reads of random lines with one write to a random line.  A write is
// atomicMax(cell, now) with `now` = s_memrealtime (the 100 MHz counter every CU shares), so a cell holds the time of its last update.
// A read loads the cell with the flavour under test and records age = now - value.  Writes reach a line as a Poisson stream of mean
// interval I = T / (writes per second), so a read that sees the memory's current content has mean age I (memoryless); a read served
// from a cache has the age of the cached copy on top.  Reported: mean age / I per flavour (1 = fresh) and the share of reads older
// than 8 I.
//   read flavours:  plain | sc1 (agent scope: bypasses the CU's L1) | sc0 sc1 (system scope) | nt
//   write flavours: atomic (global_atomic_umax_x2, executed at the memory side) | sc1 store (write-through) | plain store (stays dirty in
//                   the writer's L2 until it is written back: the case that is NOT coherent across XCDs)
//   hipcc --offload-arch=gfx950 -O3 -o live_staleness live_staleness.hip && ./live_staleness [blocks iters reads_per_write write_flavour]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef const __attribute__((address_space(1))) unsigned long long* gcell_t;

template <int FLAVOUR>
__device__ __forceinline__ unsigned long long load_cell(const unsigned long long* p)
{
    gcell_t q = (gcell_t)p;
    if (FLAVOUR == 1) return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // global_load_dwordx2 ... sc1
    if (FLAVOUR == 2) return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // ... sc0 sc1
    if (FLAVOUR == 3) return __builtin_nontemporal_load(q);                                          // ... nt
    return *q;
}

// one lane per (wave, lane) stream: the 64 lanes of a wave read 64 different random lines per step (as a gather does)
template <int WRITE>
__device__ __forceinline__ void store_cell(unsigned long long* p, unsigned long long v)
{
    if (WRITE == 0) __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (WRITE == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // global_store_dwordx2 ... sc1
    else *(volatile unsigned long long*)p = v;
}

template <int FLAVOUR, int WRITE>
__global__ __launch_bounds__(256) void staleness_kernel(unsigned long long* table, unsigned int lines, int iters, int reads_per_write,
                                                        unsigned long long threshold, unsigned long long* out /*[4]: sum age, reads, old reads, writes*/)
{
    const unsigned int tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int rs = 0x9E3779B9u * (tid + 1);
    unsigned long long sum = 0, n = 0, old = 0, w = 0;
    for (int it = 0; it < iters; it++) {
        for (int r = 0; r < reads_per_write; r++) {
            rs = rs * 1664525u + 1013904223u;
            const unsigned int line = (unsigned int)(((unsigned long long)(rs >> 4) * lines) >> 28);
            const unsigned long long v = load_cell<FLAVOUR>(table + (size_t)line * 16);
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (v != 0 && now > v) { const unsigned long long age = now - v; sum += age; n++; old += age > threshold; }
        }
        rs = rs * 1664525u + 1013904223u;
        const unsigned int line = (unsigned int)(((unsigned long long)(rs >> 4) * lines) >> 28);
        store_cell<WRITE>(table + (size_t)line * 16, (unsigned long long)__builtin_amdgcn_s_memrealtime());
        w++;
    }
    // wave totals -> four atomics per wave
    for (int s = 32; s >= 1; s >>= 1) { sum += __shfl_xor(sum, s, 64); n += __shfl_xor(n, s, 64); old += __shfl_xor(old, s, 64); w += __shfl_xor(w, s, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], sum); atomicAdd(&out[1], n); atomicAdd(&out[2], old); atomicAdd(&out[3], w); }
}

template <int FLAVOUR, int WRITE>
static void run(const char* name, unsigned long long* table, unsigned int lines, int blocks, int iters, int rpw, double interval_guess_ticks)
{
    unsigned long long* out; CK(hipMalloc(&out, 32));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    // pass 1 fills the cells and measures the write rate; pass 2 is the one reported (threshold = 8 intervals of pass 1)
    double interval = interval_guess_ticks;
    for (int pass = 0; pass < 2; pass++) {
        CK(hipMemset(out, 0, 32));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((staleness_kernel<FLAVOUR, WRITE>), dim3(blocks), dim3(256), 0, 0, table, lines, iters, rpw, (unsigned long long)(8 * interval), out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        unsigned long long h[4]; CK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
        interval = (double)ms * 1e-3 * 1e8 * (double)lines / (double)h[3];        // ticks of 10 ns between two writes to one line
        if (pass == 1)
            printf("%-8s lines %8u (%6.1f MB)  %7.2f ms  writes/line every %9.2f us  mean age %10.2f us = %7.2f intervals   older than 8 intervals: %6.3f %%   reads %.2f G/s\n",
                   name, lines, lines * 128.0 / 1e6, ms, interval * 1e-2, (double)h[0] / (double)h[1] * 1e-2, (double)h[0] / (double)h[1] / interval,
                   100.0 * (double)h[2] / (double)h[1], (double)h[1] / ms * 1e-6);
    }
    CK(hipFree(out));
}

int main(int argc, char** argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 256 * 7;          // 7 waves per SIMD, like the 1-round sweep kernel
    const int iters = argc > 2 ? atoi(argv[2]) : 400;
    const int rpw = argc > 3 ? atoi(argv[3]) : 8;
    const int wf = argc > 4 ? atoi(argv[4]) : 0;
    printf("writes: %s\n", wf == 0 ? "agent-scope atomic max (memory side)" : wf == 1 ? "sc1 (write-through) stores" : "plain stores");
    const unsigned int sizes[] = {8192u, 65536u, 114688u /* 14 MB: C3's mirror */, 393216u /* 48 MB: C4's mirror */, 786432u /* 96 MB: C4's counts */};
    for (unsigned int lines : sizes) {
        unsigned long long* table; CK(hipMalloc(&table, (size_t)lines * 128));
        CK(hipMemset(table, 0, (size_t)lines * 128));
        if (wf == 0) { run<0, 0>("plain", table, lines, blocks, iters, rpw, 100.0); run<1, 0>("sc1", table, lines, blocks, iters, rpw, 100.0);
                       run<2, 0>("sc0sc1", table, lines, blocks, iters, rpw, 100.0); run<3, 0>("nt", table, lines, blocks, iters, rpw, 100.0); }
        else if (wf == 1) { run<0, 1>("plain", table, lines, blocks, iters, rpw, 100.0); run<1, 1>("sc1", table, lines, blocks, iters, rpw, 100.0); }
        else { run<0, 2>("plain", table, lines, blocks, iters, rpw, 100.0); run<1, 2>("sc1", table, lines, blocks, iters, rpw, 100.0); }
        CK(hipFree(table));
        printf("\n");
    }
    return 0;
}
