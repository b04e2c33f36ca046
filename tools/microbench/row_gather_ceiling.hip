// row_gather_ceiling.hip — what a BARE gather of the sweep kernel's access pattern reaches on MI355X: the ceiling the `physical`
// roofline of bench.py is held against.
//
// The settled C4 sweep kernel (1-round variant on the 16-bit mirror) gathers, per token, the ~45 listed topics' cells of one
// random 800-byte row of the [60000][400] uint16 mirror (48 MB): 2-byte loads by lane, sorted columns, one row ahead in flight.
// Here 7 waves per SIMD do nothing else: every wave reads `iters` random rows, two rows in flight.  Reported: rows per second, the
// 128-byte lines a row's gather touches (from the column lists and the four alignments an 800-byte row can have), and their product
// -- the fabric bytes per second of the bare gather -- for the sparse 2-byte gather and for the whole row read 16 bytes per lane.
//   hipcc --offload-arch=gfx950 -O3 -o row_gather_ceiling row_gather_ceiling.hip && ./row_gather_ceiling [K rows used]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool DENSE>
__global__ __launch_bounds__(256, 7) void gather_kernel(const unsigned short* __restrict__ table, int rows, int K, const int* __restrict__ cols,
                                                        int iters, unsigned long long* out)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int col = cols[(size_t)wave * 64 + lane];
    unsigned int rs = 0x9E3779B9u * (wave + 1);
    unsigned long long acc = 0;
    for (int it = 0; it < iters; it += 2) {
        rs = rs * 1664525u + 1013904223u;
        const size_t r0 = (size_t)(rs >> 8) % (size_t)rows;
        rs = rs * 1664525u + 1013904223u;
        const size_t r1 = (size_t)(rs >> 8) % (size_t)rows;
        if (DENSE) {
            const uint4* p0 = (const uint4*)(table + r0 * K);
            const uint4* p1 = (const uint4*)(table + r1 * K);
            const int n16 = K * 2 / 16;
            uint4 a = make_uint4(0, 0, 0, 0), b = a;
            if (lane < n16) { a = p0[lane]; b = p1[lane]; }
            acc += a.x + a.w + b.x + b.w;
        } else {
            const unsigned int a = table[r0 * K + col], b = table[r1 * K + col];
            acc += a + b;
        }
    }
    if (acc == 0x7fffffffffffULL) out[0] = acc;
}

template <bool DENSE>
static double run(const unsigned short* t, int rows, int K, const int* dcols, int waves, int iters)
{
    unsigned long long* out; CK(hipMalloc(&out, 16));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather_kernel<DENSE>), dim3(waves / 4), dim3(256), 0, 0, t, rows, K, dcols, iters / 4, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((gather_kernel<DENSE>), dim3(waves / 4), dim3(256), 0, 0, t, rows, K, dcols, iters, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipFree(out));
    return (double)waves * iters / (ms * 1e-3);
}

int main(int argc, char** argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 400, rows = argc > 2 ? atoi(argv[2]) : 60000, used = argc > 3 ? atoi(argv[3]) : 45;
    const int waves = 256 * 4 * 7, iters = 4000;
    unsigned short* t16;
    CK(hipMalloc(&t16, (size_t)rows * K * 2 + 256)); CK(hipMemset(t16, 1, (size_t)rows * K * 2));
    std::mt19937 g(1);
    std::vector<int> cols((size_t)waves * 64);
    double lines = 0;
    for (int w = 0; w < waves; w++) {
        std::vector<int> all(K); for (int i = 0; i < K; i++) all[i] = i;
        std::shuffle(all.begin(), all.end(), g);
        std::sort(all.begin(), all.begin() + used);
        for (int i = 0; i < 64; i++) cols[(size_t)w * 64 + i] = all[std::min(i, used - 1)];
        // distinct 128-byte lines of the gather, averaged over the alignments a row of 2K bytes can have
        const int phases = 128 / std::__gcd(128, (2 * K) % 128 ? (2 * K) % 128 : 128);
        double l = 0;
        for (int ph = 0; ph < phases; ph++) {
            const int off = (ph * 2 * K) % 128;
            std::set<int> s;
            for (int i = 0; i < used; i++) s.insert((off + 2 * all[i]) / 128);
            l += (double)s.size();
        }
        lines += l / phases;
    }
    lines /= waves;
    const double dense_lines = (2.0 * K + 127.0) / 128.0 + ((2 * K) % 128 ? 0.5 : 0.0);     // a row that does not start on a line boundary touches one more, half the time
    int* dcols; CK(hipMalloc(&dcols, cols.size() * 4)); CK(hipMemcpy(dcols, cols.data(), cols.size() * 4, hipMemcpyHostToDevice));
    const double sparse = run<false>(t16, rows, K, dcols, waves, iters), dense = run<true>(t16, rows, K, dcols, waves, iters);
    printf("table [%d][%d] uint16 = %.1f MB, %d waves (7 per SIMD), %d cells of a row per gather\n", rows, K, rows * K * 2.0 / 1e6, waves, used);
    printf("sparse 2-byte gather: %7.3f G rows/s x %5.2f lines x 128 B = %7.1f GB/s of fabric reads\n", sparse / 1e9, lines, sparse * lines * 128 / 1e9);
    printf("dense row, 16 B/lane: %7.3f G rows/s x %5.2f lines x 128 B = %7.1f GB/s of fabric reads\n", dense / 1e9, dense_lines, dense * dense_lines * 128 / 1e9);
    return 0;
}
