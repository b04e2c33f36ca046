// mvhdp_stats.hip — device side of the steps either side of the sweep (SURVEY §8f "next" rows):
//   count_hist_kernel     countHistogram of optimizeBeta                   PTM:2295-2309
//   view_overlap_kernel   pDistr_Mean[m][i][doc] of optimizeP              PTM:2706-2782
//   loglik_doc_kernel     document half of modelLogLikelihood              PTM:3341-3373
//   loglik_topic_kernel   topic-word half of modelLogLikelihood            PTM:3387-3415
// Integer outputs are exact; floating-point per-entity values are produced with the reference's
// own expression order and summed on the host in entity order (the reference's order), so only the
// device libm (log) can differ from the CPU restatement.
#include "mvhdp_device.h"
#include "../../include/mvhdp.h"
#include "mvhdp_wave.h"

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void count_hist_kernel(const int32_t* __restrict__ nwk, int64_t n_cells, int32_t* hist, int32_t len)
{
    __shared__ int local[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) local[i] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cells; i += stride) {
        int c = nwk[i];
        if (c > 0 && c < len) {                                   // PTM:2305 count > 0
            if (c < 1024) atomicAdd(&local[c], 1);
            else atomicAdd(&hist[c], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024 && i < len; i += blockDim.x)
        if (local[i]) atomicAdd(&hist[i], local[i]);
}

hipError_t mvhdp_launch_count_hist(const MvModel& mm, int m, int32_t* hist, int32_t len, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(hist, 0, (size_t)len * sizeof(int32_t), s);
    if (e != hipSuccess) return e;
    int64_t n = (int64_t)mm.V[m] * mm.K;
    int64_t blocks = (n + 255) / 256;
    int grid = (int)(blocks < 2048 ? (blocks < 1 ? 1 : blocks) : 2048);
    hipLaunchKernelGGL(count_hist_kernel, dim3(grid), dim3(256), 0, s, mm.counts + mm.rowbase[m] * mm.K, n, hist, len);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// optimizeP: one wave per entity.  out[(m*M+i)*D + doc] = pDistr_Mean[m][i][doc].
// The TreeMap<Integer,Byte> keyed by view length (PTM:2717,2741) keeps one view per distinct
// length (the later view wins); views are then visited by descending length (PTM:2744).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void view_overlap_kernel(MvModel mm, double* out)
{
    __shared__ uint32_t bm[4][MVHDP_MAXM][64];                   // topic-presence bitmaps per view
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, M = mm.M;
    const int64_t wstride = (int64_t)gridDim.x * 4;
    for (int64_t d = (int64_t)blockIdx.x * 4 + wave; d < mm.D; d += wstride) {
        int len[MVHDP_MAXM];
        for (int m = 0; m < M; m++) {
            bm[wave][m][lane] = 0;
            len[m] = (int)(mm.doc_off[m][d + 1] - mm.doc_off[m][d]);
        }
        LDS_FENCE();
        for (int m = 0; m < M; m++) {
            const int64_t b = mm.doc_off[m][d];
            for (int i = lane; i < len[m]; i += WAVE) {
                int zz = mm.z[m][b + i];
                if (zz >= 0) atomicOr(&bm[wave][m][zz >> 5], 1u << (zz & 31));   // localTopicCounts[m][z] > 0
            }
        }
        LDS_FENCE();
        // TreeMap put / descending iteration, on every lane alike
        int keys[MVHDP_MAXM], vals[MVHDP_MAXM], nkeys = 0;
        for (int m = 0; m < M; m++) {
            int found = -1;
            for (int q = 0; q < nkeys; q++) if (keys[q] == len[m]) found = q;
            if (found >= 0) vals[found] = m; else { keys[nkeys] = len[m]; vals[nkeys] = m; nkeys++; }
        }
        for (int a = 0; a < nkeys; a++) for (int b2 = a + 1; b2 < nkeys; b2++)
            if (keys[b2] > keys[a]) { int t = keys[a]; keys[a] = keys[b2]; keys[b2] = t; t = vals[a]; vals[a] = vals[b2]; vals[b2] = t; }
        for (int i = lane; i < M * M; i += WAVE) out[(int64_t)i * mm.D + d] = 0.0;
        for (int q = 1; q < nkeys; q++) {                               // PTM:2751
            const int m = vals[q];
            if (len[m] <= 0) continue;                                    // Assignments[m] == null
            const int64_t b = mm.doc_off[m][d];
            for (int pi = 0; pi < q; pi++) {                              // previousViews (PTM:2769)
                const int iv = vals[pi];
                int c = 0;
                for (int i0 = 0; i0 < len[m]; i0 += WAVE) {
                    const int i = i0 + lane;
                    bool hit = false;
                    if (i < len[m]) { int zz = mm.z[m][b + i]; hit = zz >= 0 && ((bm[wave][iv][zz >> 5] >> (zz & 31)) & 1u); }
                    c += (int)__popcll(__builtin_amdgcn_ballot_w64(hit));
                }
                if (lane == 0) {
                    // PTM:2771: c additions of 1.0/docLength[m] (the zero additions change nothing)
                    const double x = 1.0 / (double)len[m];
                    double acc = 0.0;
                    for (int r = 0; r < c; r++) acc += x;
                    out[(int64_t)(m * M + iv) * mm.D + d] = acc;
                    out[(int64_t)(iv * M + m) * mm.D + d] = acc;          // PTM:2772
                }
            }
        }
        LDS_FENCE();
    }
}

hipError_t mvhdp_launch_view_overlap(const MvModel& mm, double* out, hipStream_t s)
{
    if (mm.D <= 0) return hipSuccess;
    int64_t blocks = (mm.D + 3) / 4;
    int grid = (int)(blocks < 4096 ? blocks : 4096);
    hipLaunchKernelGGL(view_overlap_kernel, dim3(grid), dim3(256), 0, s, mm, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// MALLET Dirichlet.logGammaStirling (restated from the 2.0.8 class file)
// ---------------------------------------------------------------------------
__device__ __forceinline__ double log_gamma_stirling(double z)
{
    const double HALF_LOG_TWO_PI = 0.91893853320467274178;       // log(6.283185307179586) / 2.0
    int shift = 0;
    while (z < 2.0) { z = z + 1; shift++; }
    double result = HALF_LOG_TWO_PI + (z - 0.5) * log(z) - z + 1 / (12.0 * z) - 1 / (360.0 * z * z * z)
                    + 1 / (1260.0 * z * z * z * z * z);
    while (shift > 0) { shift--; z = z - 1; result = result - log(z); }
    return result;
}

// document half: out[doc] = sum_k [lgs(gamma*alpha_k + n_dk) - lgs(gamma*alpha_k)] - lgs(gamma*alphaSum + backing length)
// (0 and not counted when the view is absent); PTM:3347-3370 incl. the backing-array phantom zeros.
__global__ __launch_bounds__(256) void loglik_doc_kernel(MvModel mm, int m, double* out)
{
    extern __shared__ int ldk[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, K = mm.K;
    int* cnt = ldk + wave * K;
    const double g = mm.gamma[m];
    const double* al = mm.alpha + (int64_t)m * (K + 1);
    const int wpb = blockDim.x >> 6;
    const int64_t wstride = (int64_t)gridDim.x * wpb;
    for (int64_t d = (int64_t)blockIdx.x * wpb + wave; d < mm.D; d += wstride) {
        const int64_t b = mm.doc_off[m][d], e = mm.doc_off[m][d + 1];
        if (mm.present[m] ? !mm.present[m][d] : e == b) { if (lane == 0) out[d] = 0.0; continue; }   // the entity lacks the view
        for (int k = lane; k < K; k += WAVE) cnt[k] = 0;
        LDS_FENCE();
        for (int64_t i = b + lane; i < e; i += WAVE) { int zz = mm.z[m][i]; atomicAdd(&cnt[zz < 0 ? 0 : zz], 1); }
        const int backing = (int)(e - b) > 2 ? (int)(e - b) : 2;
        if (lane == 0 && backing > (int)(e - b)) atomicAdd(&cnt[0], backing - (int)(e - b));
        LDS_FENCE();
        double acc = 0.0;
        for (int k = lane; k < K; k += WAVE) {
            int c = cnt[k];
            if (c > 0) acc += log_gamma_stirling(g * al[k] + c) - log_gamma_stirling(g * al[k]);
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s, WAVE);
        if (lane == 0) out[d] = acc - log_gamma_stirling((double)g * mm.alpha_sum[m] + backing);
        LDS_FENCE();
    }
}

// topic-word half: per-block partial sums of lgs(beta + count) over count > 0, and the number of such pairs
__global__ __launch_bounds__(256) void loglik_topic_kernel(const int32_t* __restrict__ nwk, int64_t n_cells, double beta,
                                                           double* partial, unsigned long long* nonzero)
{
    __shared__ double red[256];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    unsigned int nz = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cells; i += stride) {
        int c = nwk[i];
        if (c > 0) { nz++; acc += log_gamma_stirling(beta + c); }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
    if (nz) atomicAdd(nonzero, (unsigned long long)nz);
}

hipError_t mvhdp_launch_loglik_doc(const MvModel& mm, int m, double* doc_out, hipStream_t s)
{
    if (mm.D <= 0) return hipSuccess;
    int wpb = 4;
    while (wpb > 1 && (size_t)wpb * mm.K * sizeof(int) > 60000) wpb >>= 1;
    int64_t blocks = (mm.D + wpb - 1) / wpb;
    int grid = (int)(blocks < 4096 ? blocks : 4096);
    hipLaunchKernelGGL(loglik_doc_kernel, dim3(grid), dim3(64 * wpb), (size_t)wpb * mm.K * sizeof(int), s, mm, m, doc_out);
    return hipGetLastError();
}

hipError_t mvhdp_launch_loglik_topic(const MvModel& mm, int m, double* partial, int n_partial, unsigned long long* nonzero, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(nonzero, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    int64_t n = (int64_t)mm.V[m] * mm.K;
    hipLaunchKernelGGL(loglik_topic_kernel, dim3(n_partial), dim3(256), 0, s, mm.counts + mm.rowbase[m] * mm.K, n,
                       mm.beta[m], partial, nonzero);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// doc_topic_prop: the per-entity topic proportions of printDocumentTopics (PTM:2871-2899, the same lines in
// the inferencer INF:383-411): per view the topic counts of the entity's assignments, then for every topic
//   sum_m w[m] * (n_dk[m][k] + gamma[m]*alpha[m][k]) / (len[m] + gamma[m]*alphaSum[m])  /  sum_m w[m]
// with w[m] = (m == 0 ? 1 : discrWeightPerModality[m]) * pMean[0][m], products and quotient in the
// reference's left-to-right order.  One wave per entity, counts in LDS.
// The reference reuses topicCounts[m] / docLen[m] across its entity loop and refreshes them only when the entity HAS
// view m (PTM:2873-2886): an entity without view m is scored with the counts and length of the last earlier entity
// that had it (zeros before the first).  carry.src[m][d] names that entity (d itself when it has the view, -1 none).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void doc_topic_prop_kernel(MvModel mm, DocTopicCarry carry, const double* __restrict__ w, int64_t d0, int64_t d1, double* out)
{
    extern __shared__ int ndk[];                           // [M][K]
    const int lane = threadIdx.x, K = mm.K, M = mm.M;
    for (int64_t d = d0 + blockIdx.x; d < d1; d += gridDim.x) {
        for (int i = lane; i < M * K; i += WAVE) ndk[i] = 0;
        __syncthreads();
        for (int m = 0; m < M; m++) {
            const int64_t sd = carry.src[m][d];
            if (sd < 0) continue;
            const int64_t b = mm.doc_off[m][sd], e = mm.doc_off[m][sd + 1];
            for (int64_t i = b + lane; i < e; i += WAVE) { const int zz = mm.z[m][i]; if (zz >= 0 && zz < K) atomicAdd(&ndk[m * K + zz], 1); }
        }
        __syncthreads();
        double norm = 0;
        for (int m = 0; m < M; m++) norm += w[m];
        double* o = out + (d - d0) * K;
        for (int k = lane; k < K; k += WAVE) {
            double tp = 0;
            for (int m = 0; m < M; m++) {
                const int64_t sd = carry.src[m][d];
                const int len = sd < 0 ? 0 : (int)(mm.doc_off[m][sd + 1] - mm.doc_off[m][sd]);
                tp += w[m] * ((double)ndk[m * K + k] + mm.gamma[m] * mm.alpha[(int64_t)m * (K + 1) + k]) / (len + mm.gamma[m] * mm.alpha_sum[m]);
            }
            o[k] = tp / norm;
        }
        __syncthreads();
    }
}

hipError_t mvhdp_launch_doc_topic_prop(const MvModel& mm, const DocTopicCarry& carry, const double* w_dev, int64_t d0, int64_t d1, double* out_dev, hipStream_t s)
{
    if (d1 <= d0) return hipSuccess;
    int64_t n = d1 - d0;
    int grid = (int)(n < 16384 ? n : 16384);
    hipLaunchKernelGGL(doc_topic_prop_kernel, dim3(grid), dim3(64), (size_t)mm.M * mm.K * sizeof(int), s, mm, carry, w_dev, d0, d1, out_dev);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// gamma_doc_stats: the document-level auxiliary variables of optimizeGamma (PTM:2415-2433, Teh et al. 2006): over the
// entities that have view m, of length j,
//     qs = sum Bernoulli(j / (j + gamma_m)),     qw = sum log Beta(gamma_m + 1, j).
// The reference draws them one entity after the other from `samp`, a RandomSamplers over ThreadLocalRandom -- a stream
// nobody can seed or replay -- ten times per view and optimisation step: 26 M sequential draws at C4, 1.9 s of host time
// where a sweep takes 36 ms.  Here every entity draws from its own Philox stream (counter = global entity id, view,
// round), so the two sums are the same random variables in distribution, deterministic for a given seed, and independent
// of the launch geometry (per-block partials are summed in block order on the host).
// Gamma(a), a >= 1: Marsaglia & Tsang (2000); normals by Box-Muller.  Beta(a, b) = G_a / (G_a + G_b).
// ---------------------------------------------------------------------------
struct PhiloxStream {
    uint32_t c0, c1, c2, c3, k0, k1;
    uint32_t x[4]; int have;
    __device__ double uniform()
    {
        if (have == 0) { philox4x32_10(c0, c1, c2, c3, k0, k1, x); c0++; have = 2; }
        have--;
        return have == 1 ? bits_to_unit(x[0], x[1]) : bits_to_unit(x[2], x[3]);
    }
};

__device__ double mt_gamma(PhiloxStream& r, double a)      // a >= 1
{
    const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        double u1 = r.uniform(), u2 = r.uniform();
        if (u1 <= 0.0) u1 = 0x1.0p-53;
        const double n = sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
        const double t = 1.0 + c * n;
        if (t <= 0.0) continue;
        const double v = t * t * t;
        double u = r.uniform();
        if (u <= 0.0) u = 0x1.0p-53;
        if (u < 1.0 - 0.0331 * (n * n) * (n * n)) return d * v;
        if (log(u) < 0.5 * n * n + d * (1.0 - v + log(v))) return d * v;
    }
}

__global__ __launch_bounds__(256) void gamma_doc_stats_kernel(MvModel mm, int m, double gamma_m, uint32_t seed_lo, uint32_t seed_hi,
                                                              uint32_t round, double* partial /*[gridDim.x][2]*/)
{
    __shared__ double sh[2][256];
    double qs = 0.0, qw = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; d < mm.D; d += stride) {
        const int64_t j = mm.doc_off[m][d + 1] - mm.doc_off[m][d];
        if (j <= 0) continue;                                           // the entity lacks the view (MTA:19)
        const int64_t dg = mm.doc_id_base + d;
        PhiloxStream r;
        r.c0 = 0; r.c1 = 0x200u + (uint32_t)m; r.c2 = (uint32_t)dg; r.c3 = round;
        r.k0 = seed_lo; r.k1 = seed_hi ^ (uint32_t)((unsigned long long)dg >> 32); r.have = 0;
        qs += (r.uniform() < (double)j / ((double)j + gamma_m)) ? 1.0 : 0.0;            // PTM:2418-2419
        const double ga = mt_gamma(r, gamma_m + 1.0), gb = mt_gamma(r, (double)j);
        qw += log(ga / (ga + gb));                                                    // PTM:2420-2421
    }
    sh[0][threadIdx.x] = qs; sh[1][threadIdx.x] = qw;
    __syncthreads();
    for (int s2 = 128; s2 >= 1; s2 >>= 1) {                             // fixed-order tree: the same bits for the same grid
        if ((int)threadIdx.x < s2) { sh[0][threadIdx.x] += sh[0][threadIdx.x + s2]; sh[1][threadIdx.x] += sh[1][threadIdx.x + s2]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = sh[0][0]; partial[2 * blockIdx.x + 1] = sh[1][0]; }
}

// optimizeDP, the view-table simulation PTM:2454-2488, one thread per (topic t, count i) cell of topicDocCounts[m]: a cell with
// i == 1 holds entities with one table each; a cell with i > 1 takes ONE draw of the number of tables a Chinese restaurant process of
// concentration conc[t] = gamma_m * alpha[m][t] makes of i items (Antoniak 1974), times the entities in the cell (the reference draws
// once per cell, PTM:2471-2477).  The number of tables of a CRP is a sum of independent Bernoullis, table l + 1 opened with probability
// conc / (conc + l): the distribution Samplers.randAntoniak draws from through its table of Stirling numbers (Samplers.java:1086-1110),
// without that table -- and without the call's in-place scaling of the CACHED row, which in the reference carries one call's alpha powers
// into the next.  Streams: Philox keyed by (seed, round), counter (draw, view, topic, count): reproducible, independent of the launch.
// mk[t] = sum over the cells; active[t] = 1 iff a cell with i >= 1 holds an entity (the topic leaves inActiveTopicIndex, PTM:2461,2480).
__global__ __launch_bounds__(256) void dp_tables_kernel(const int32_t* __restrict__ hist, int hist_len, int m, const double* __restrict__ conc,
                                                        uint32_t seed_lo, uint32_t seed_hi, uint32_t round, double* __restrict__ mk, uint8_t* __restrict__ active)
{
    __shared__ double sh[256];
    __shared__ int any;
    const int t = blockIdx.x;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    const double a = conc[t];
    double acc = 0.0;
    for (int i = 1 + (int)threadIdx.x; i < hist_len; i += blockDim.x) {
        const int n = hist[(int64_t)t * hist_len + i];
        if (n <= 0) continue;
        any = 1;
        if (i == 1) { acc += (double)n; continue; }                     // PTM:2479-2483
        int tables = 1;                                                 // (the first item opens the first table)
        if (a > 0.0) {
            PhiloxStream r;
            r.c0 = 0; r.c1 = 0x300u + (uint32_t)m; r.c2 = (uint32_t)t; r.c3 = (uint32_t)i;
            r.k0 = seed_lo ^ round; r.k1 = seed_hi; r.have = 0;
            for (int l = 1; l < i; l++) tables += (r.uniform() * (a + (double)l) < a) ? 1 : 0;
        }
        acc += (double)n * (double)tables;                              // PTM:2477
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s2 = 128; s2 >= 1; s2 >>= 1) { if ((int)threadIdx.x < s2) sh[threadIdx.x] += sh[threadIdx.x + s2]; __syncthreads(); }
    if (threadIdx.x == 0) { mk[t] = sh[0]; active[t] = any ? 1 : 0; }
}

hipError_t mvhdp_launch_dp_tables(const int32_t* hist, int hist_len, int K, int m, const double* conc, uint32_t seed_lo, uint32_t seed_hi, uint32_t round,
                                  double* mk, uint8_t* active, hipStream_t s)
{
    hipLaunchKernelGGL(dp_tables_kernel, dim3(K), dim3(256), 0, s, hist, hist_len, m, conc, seed_lo, seed_hi, round, mk, active);
    return hipGetLastError();
}

// n independent Antoniak draws, one thread each (optimizeDP's root level PTM:2491-2517: K * M of them, of up to MAXSTIRLING items each):
// tables[j] = the number of tables a CRP(conc[j]) makes of items[j] items.  items <= 0: 0; 1: 1; more than 20000: 1 -- the reference's
// table of Stirling numbers ends there (Samplers.java:1024,1034: allss = new double[MAXSTIRLING][]), its call throws and the caller
// falls back to one table (PTM:2507-2509).  Filling that table up to the largest count asked for is what an optimising iteration of the
// host loop spends its time on (n^2 / 2 products: 0.4 s at C4).
__global__ __launch_bounds__(256) void antoniak_draws_kernel(int n, const int32_t* __restrict__ items, const double* __restrict__ conc,
                                                             uint32_t seed_lo, uint32_t seed_hi, uint32_t round, int32_t* __restrict__ tables)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int it = items[j];
    const double a = conc[j];
    int t = it <= 0 ? 0 : 1;
    if (it > 1 && it <= 20000 && a > 0.0) {
        PhiloxStream r;
        r.c0 = 0; r.c1 = 0x400u; r.c2 = (uint32_t)j; r.c3 = round;
        r.k0 = seed_lo; r.k1 = seed_hi; r.have = 0;
        for (int l = 1; l < it; l++) t += (r.uniform() * (a + (double)l) < a) ? 1 : 0;
    }
    tables[j] = t;
}

hipError_t mvhdp_launch_antoniak_draws(int n, const int32_t* items, const double* conc, uint32_t seed_lo, uint32_t seed_hi, uint32_t round, int32_t* tables, hipStream_t s)
{
    hipLaunchKernelGGL(antoniak_draws_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, items, conc, seed_lo, seed_hi, round, tables);
    return hipGetLastError();
}

hipError_t mvhdp_launch_gamma_doc_stats(const MvModel& mm, int m, double gamma_m, uint32_t seed_lo, uint32_t seed_hi, uint32_t round,
                                        double* partial, int n_blocks, hipStream_t s)
{
    hipLaunchKernelGGL(gamma_doc_stats_kernel, dim3(n_blocks), dim3(256), 0, s, mm, m, gamma_m, seed_lo, seed_hi, round, partial);
    return hipGetLastError();
}
