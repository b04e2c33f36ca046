// fetch_calib.hip — calibrate rocprofv3's memory-side read counters (FETCH_SIZE, TCC_EA0_RDREQ*) on the
// access patterns of the sweep kernel, where the bytes touched are known by construction.
//
// The sweep kernel reads (per token) a sparse, sorted 4-byte gather inside one 4K-byte n_wk row and three 64-byte
// blocks of the descent table.  MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide coalesced streams (it
// reports 1/2 of the bytes there); "other access widths are uncalibrated".  Each kernel below touches a
// host-computable set of 32/64/128-byte units; run the binary under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./fetch_calib
//   rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B -- ./fetch_calib
//   rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS TCC_READ TCC_REQ -- ./fetch_calib
// and compare (profiles/calib_summary.py).  The binary prints one JSON line per kernel with the true counts.
//
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>
#include <unordered_set>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__host__ __device__ static inline uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__host__ __device__ static inline uint64_t row_of(uint32_t wave, uint32_t it, uint64_t rows)
{
    const uint64_t h = ((uint64_t)mix32(wave * 0x9E3779B9u + it) << 32) | mix32(it * 0x85EBCA6Bu + wave + 0x1234567u);
    return h % rows;
}

// ---- 1. wide coalesced stream: 16 bytes per lane, every byte of the buffer once ----
__global__ __launch_bounds__(256) void calib_stream16(const int4* __restrict__ buf, size_t n16, unsigned long long* out)
{
    long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const int4 v = buf[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 0x7fffffffffffLL) out[0] = acc;
}

// ---- 2. sparse sorted 4-byte gather inside random rows (the n_wk gather): R loads per lane, blocked slots ----
template <int R>
__global__ __launch_bounds__(256) void calib_row_gather4(const int* __restrict__ table, uint64_t rows, int K,
                                                         const int* __restrict__ cols /*[waves][64*R]*/, int iters, unsigned long long* out)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int koff[R];
#pragma unroll
    for (int r = 0; r < R; r++) koff[r] = cols[(size_t)wave * 64 * R + lane * R + r];
    long long acc = 0;
    for (int it = 0; it < iters; it++) {
        const int* rp = table + row_of(wave, it, rows) * K;
#pragma unroll
        for (int r = 0; r < R; r++) acc += rp[koff[r]];
    }
    if (acc == 0x7fffffffffffLL) out[0] = acc;
}

// ---- 3. dense random rows, 16 bytes per lane ----
__global__ __launch_bounds__(256) void calib_row_dense(const int* __restrict__ table, uint64_t rows, int K, int iters, unsigned long long* out)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    long long acc = 0;
    for (int it = 0; it < iters; it++) {
        const int4* rp = (const int4*)(table + row_of(wave, it, rows) * K);
        for (int c = lane; c * 4 < K; c += 64) { const int4 v = rp[c]; acc += v.x + v.y + v.z + v.w; }
    }
    if (acc == 0x7fffffffffffLL) out[0] = acc;
}

// ---- 4. one random 64-byte aligned block per lane (four 16-byte loads): the descent-table read ----
__global__ __launch_bounds__(256) void calib_block64(const double2* __restrict__ blocks, uint64_t nblocks, int iters, unsigned long long* out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0;
    for (int it = 0; it < iters; it++) {
        const double2* b = blocks + row_of(tid, it, nblocks) * 4;
        const double2 q0 = b[0], q1 = b[1], q2 = b[2], q3 = b[3];
        acc += q0.x + q0.y + q1.x + q1.y + q2.x + q2.y + q3.x + q3.y;
    }
    if (acc == 1.2345e300) out[0] = 1;
}

struct Units { double u32 = 0, u64 = 0, u128 = 0, accesses = 0; };

static void report(const char* name, const char* kernel, double ms, const Units& u, double bytes_useful, const char* served)
{
    printf("{\"case\": \"%s\", \"kernel\": \"%s\", \"ms\": %.4f, \"accesses\": %.0f, \"useful_bytes\": %.0f, "
           "\"units32\": %.0f, \"units64\": %.0f, \"units128\": %.0f, \"served_from\": \"%s\"}\n",
           name, kernel, ms, u.accesses, bytes_useful, u.u32, u.u64, u.u128, served);
    fflush(stdout);
}

template <typename F>
static double timed(F&& launch)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms;
}

int main(int argc, char** argv)
{
    const int K = 400, waves = 6144;
    const int iters = (argc > 1) ? atoi(argv[1]) : 2000;
    unsigned long long* out; CK(hipMalloc(&out, 16));
    // one allocation serves every case: 2 GiB (far beyond the 256 MiB Infinity Cache and the 32 MiB of L2)
    const size_t big_bytes = (size_t)2 << 30;
    char* big; CK(hipMalloc(&big, big_bytes + 4096));
    CK(hipMemset(big, 1, big_bytes));
    std::mt19937 g(7);

    // 1. stream
    {
        const size_t n16 = big_bytes / 16;
        const double ms = timed([&] { hipLaunchKernelGGL(calib_stream16, dim3(256 * 8), dim3(256), 0, 0, (const int4*)big, n16, out); });
        Units u; u.accesses = (double)n16; u.u32 = big_bytes / 32.0; u.u64 = big_bytes / 64.0; u.u128 = big_bytes / 128.0;
        report("stream16_2GiB", "calib_stream16", ms, u, (double)big_bytes, "HBM");
    }
    // 2./3. row gathers on a 96 MB table (the size of C4's n_wk: Infinity-Cache resident) and on a 1.92 GB table (HBM)
    for (int big_table = 0; big_table < 2; big_table++) {
        const uint64_t rows = big_table ? 1200000 : 60000;
        const char* served = big_table ? "HBM (1.92 GB table)" : "Infinity Cache (96 MB table)";
        for (int R : {1, 2}) {
            const int S = 64 * R, used = (R == 1) ? 56 : 90;          // typical topic-list sizes at C4
            std::vector<int> cols((size_t)waves * S);
            for (int w = 0; w < waves; w++) {
                std::vector<int> all(K); for (int i = 0; i < K; i++) all[i] = i;
                std::shuffle(all.begin(), all.end(), g);
                std::sort(all.begin(), all.begin() + used);
                for (int i = 0; i < S; i++) cols[(size_t)w * S + i] = all[std::min(i, used - 1)];
            }
            int* dcols; CK(hipMalloc(&dcols, cols.size() * 4));
            CK(hipMemcpy(dcols, cols.data(), cols.size() * 4, hipMemcpyHostToDevice));
            double ms;
            if (R == 1) ms = timed([&] { hipLaunchKernelGGL((calib_row_gather4<1>), dim3(waves / 4), dim3(256), 0, 0, (const int*)big, rows, K, dcols, iters, out); });
            else        ms = timed([&] { hipLaunchKernelGGL((calib_row_gather4<2>), dim3(waves / 4), dim3(256), 0, 0, (const int*)big, rows, K, dcols, iters, out); });
            // units touched per (wave, iteration): distinct 32/64/128-byte units of the row's gathered columns
            Units u;
            for (int w = 0; w < waves; w++) {
                // the unit pattern depends on the row's base alignment modulo 128 only (row stride 1600 B = 12.5 lines)
                double n32[2], n64[2], n128[2];
                for (int par = 0; par < 2; par++) {
                    std::unordered_set<uint64_t> s32, s64, s128;
                    const uint64_t base = (uint64_t)par * 1600;
                    for (int i = 0; i < used; i++) {
                        const uint64_t a = base + (uint64_t)cols[(size_t)w * S + i] * 4;
                        s32.insert(a / 32); s64.insert(a / 64); s128.insert(a / 128);
                    }
                    n32[par] = s32.size(); n64[par] = s64.size(); n128[par] = s128.size();
                }
                for (int it = 0; it < iters; it++) {
                    const int par = (int)(row_of(w, it, rows) & 1);
                    u.u32 += n32[par]; u.u64 += n64[par]; u.u128 += n128[par];
                }
            }
            u.accesses = (double)waves * iters;
            char nm[96]; snprintf(nm, sizeof nm, "row_gather4_R%d_%s", R, big_table ? "1p92GB" : "96MB");
            report(nm, R == 1 ? "calib_row_gather4<1>" : "calib_row_gather4<2>", ms, u, (double)waves * iters * used * 4, served);
            CK(hipFree(dcols));
        }
        {
            const double ms = timed([&] { hipLaunchKernelGGL(calib_row_dense, dim3(waves / 4), dim3(256), 0, 0, (const int*)big, rows, K, iters, out); });
            Units u; u.accesses = (double)waves * iters;
            for (int w = 0; w < waves; w++)
                for (int it = 0; it < iters; it++) {
                    const int par = (int)(row_of(w, it, rows) & 1);
                    u.u32 += 50; u.u64 += 25; u.u128 += par ? 13 : 13;      // 1600 B from a 64-byte aligned base: 12.5 lines -> 13 either way
                }
            char nm[96]; snprintf(nm, sizeof nm, "row_dense_%s", big_table ? "1p92GB" : "96MB");
            report(nm, "calib_row_dense", ms, u, (double)waves * iters * 1600, served);
        }
    }
    // 4. 64-byte blocks: a 282 MB table (C4's descent table) and the 2 GiB one
    for (int big_table = 0; big_table < 2; big_table++) {
        const uint64_t nblocks = big_table ? big_bytes / 64 : (uint64_t)282 * 1000 * 1000 / 64;
        const int it4 = iters / 4 > 0 ? iters / 4 : 1;
        const double ms = timed([&] { hipLaunchKernelGGL(calib_block64, dim3(waves / 4), dim3(256), 0, 0, (const double2*)big, nblocks, it4, out); });
        Units u; u.accesses = (double)waves * 64 * it4; u.u32 = 2 * u.accesses; u.u64 = u.accesses; u.u128 = u.accesses;
        char nm[96]; snprintf(nm, sizeof nm, "block64_%s", big_table ? "2GiB" : "282MB");
        report(nm, "calib_block64", ms, u, u.accesses * 64, big_table ? "HBM" : "HBM / Infinity Cache (282 MB table)");
    }
    CK(hipFree(big)); CK(hipFree(out));
    return 0;
}
