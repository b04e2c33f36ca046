// gather_patterns.hip — what a 64-lane gather of S sorted column indices from one random row of an
// [rows][K] int32 table costs on MI355X, by access pattern.  Diagnostic for the sweep kernel's n_wk gather
// (DESIGN.md §8): every wave reads `iters` random rows; per row it needs the values of S = 64*R columns chosen
// at random but sorted, as the topic list of an entity is.
//   hipcc --offload-arch=gfx950 -O3 -o gather_patterns gather_patterns.hip && ./gather_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int R, int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const int* __restrict__ table, const unsigned short* __restrict__ table16,
                                                     int rows, int K, const int* __restrict__ cols /*[waves][64*R] sorted*/,
                                                     int iters, unsigned long long* out)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int* mycols = cols + (size_t)wave * 64 * R;
    int koff[R];
#pragma unroll
    for (int r = 0; r < R; r++) koff[r] = (MODE == 1) ? mycols[r * 64 + lane] : mycols[lane * R + r];   // interleaved : blocked
    unsigned int rs = 0x9E3779B9u * (wave + 1);
    long long acc = 0;
    for (int it = 0; it < iters; it++) {
        rs = rs * 1664525u + 1013904223u;
        const size_t row = (size_t)(rs >> 8) % (size_t)rows;
        if (MODE == 0 || MODE == 1) {            // 4-byte gathers, blocked / interleaved slot layout
            const int* rp = table + row * K;
#pragma unroll
            for (int r = 0; r < R; r++) acc += rp[koff[r]];
        } else if (MODE == 2) {                  // 16-bit table, aligned dword loads
            const char* rp = (const char*)(table16 + row * K);
#pragma unroll
            for (int r = 0; r < R; r++) { unsigned v = *(const unsigned*)(rp + ((koff[r] * 2) & ~3)); acc += (v >> ((koff[r] & 1) * 16)) & 0xffff; }
        } else {                                 // dense row, 16 bytes per lane per load
            const int4* rp = (const int4*)(table + row * K);
            for (int c = lane; c * 4 < K; c += 64) { int4 v = rp[c]; acc += v.x + v.y + v.z + v.w; }
        }
    }
    if (acc == 0x7fffffffffffLL) out[0] = acc;
    if (lane == 0 && wave == 0) out[1] = 1;
}

template <int R, int MODE>
static double run(const int* table, const unsigned short* t16, int rows, int K, const int* cols, int waves, int iters)
{
    unsigned long long* out; CK(hipMalloc(&out, 16));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = waves / 4;
    hipLaunchKernelGGL((gather_kernel<R, MODE>), dim3(blocks), dim3(256), 0, 0, table, t16, rows, K, cols, iters / 4, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((gather_kernel<R, MODE>), dim3(blocks), dim3(256), 0, 0, table, t16, rows, K, cols, iters, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipFree(out));
    return (double)waves * iters / (ms * 1e-3);      // rows per second
}

int main()
{
    const int K = 400, rows = 60000, waves = 6144, iters = 2000;
    int *table; unsigned short* t16;
    CK(hipMalloc(&table, (size_t)rows * K * 4 + 64)); CK(hipMalloc(&t16, (size_t)rows * K * 2 + 64));
    CK(hipMemset(table, 1, (size_t)rows * K * 4)); CK(hipMemset(t16, 1, (size_t)rows * K * 2));
    std::mt19937 g(1);
    const char* names[4] = {"4-byte gather, blocked slots (lane*R+r)", "4-byte gather, interleaved slots (r*64+lane)",
                            "16-bit table, dword loads, blocked", "dense row, 16 B per lane"};
    for (int R : {1, 2}) {
        const int S = 64 * R;
        std::vector<int> cols((size_t)waves * S);
        for (int w = 0; w < waves; w++) {
            std::vector<int> all(K); for (int i = 0; i < K; i++) all[i] = i;
            std::shuffle(all.begin(), all.end(), g);
            const int used = (R == 1) ? 56 : 90;                  // typical topic-list sizes
            std::sort(all.begin(), all.begin() + used);
            for (int i = 0; i < S; i++) cols[(size_t)w * S + i] = all[std::min(i, used - 1)];
        }
        int* dcols; CK(hipMalloc(&dcols, cols.size() * 4)); CK(hipMemcpy(dcols, cols.data(), cols.size() * 4, hipMemcpyHostToDevice));
        double v[4];
        if (R == 1) { v[0] = run<1, 0>(table, t16, rows, K, dcols, waves, iters); v[1] = run<1, 1>(table, t16, rows, K, dcols, waves, iters);
                      v[2] = run<1, 2>(table, t16, rows, K, dcols, waves, iters); v[3] = run<1, 3>(table, t16, rows, K, dcols, waves, iters); }
        else        { v[0] = run<2, 0>(table, t16, rows, K, dcols, waves, iters); v[1] = run<2, 1>(table, t16, rows, K, dcols, waves, iters);
                      v[2] = run<2, 2>(table, t16, rows, K, dcols, waves, iters); v[3] = run<2, 3>(table, t16, rows, K, dcols, waves, iters); }
        for (int m = 0; m < 4; m++) printf("R=%d  %-48s %8.2f G rows/s\n", R, names[m], v[m] / 1e9);
        CK(hipFree(dcols));
    }
    return 0;
}
