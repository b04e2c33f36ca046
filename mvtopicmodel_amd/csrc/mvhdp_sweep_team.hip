// mvhdp_sweep_team.hip — the sweep for entities with long topic lists: one WORKGROUP (four waves) per
// entity instead of one wave.
//
// Why: tokens of an entity are sequential (WRK:425 reads the counts the previous token wrote), so an
// entity's time is tokens x the latency of one token.  With 512 or 1024 topic slots in ONE wave that
// latency is 1.8-2.7 us (8-16 slot rounds per lane, 200-256 VGPRs, two waves per SIMD); a 2048-token
// entity then takes 4-6 ms and the long entities of a power-law corpus are the sweep's critical path.
// Here the slot list is spread over 256 lanes (slot i = (wave*64 + lane)*R + r, R <= 4), each wave does a
// quarter of the per-token work with the register footprint of the narrow variants, and the four waves
// meet twice per token through LDS: once for the wave totals of the prefix scan (WRK:501-513), once for
// the first slot whose cumulative mass reaches the sample (WRK:531).  All other per-token values are
// wave-uniform and computed redundantly by the four waves, so every wave takes the same branches and the
// barriers always match.
//
// Same arithmetic, same certified scan (tolerance check + exact sequential fallback), same RNG and the
// same results as sweep_fast_kernel / sweep_kernel; only the summation order inside the tolerance differs.
#include "mvhdp_device.h"
#include "../../include/mvhdp.h"
#include "mvhdp_wave.h"

#define TEAM_WAVES 4
#define TEAM_LANES 256
// Barrier of the token loop: the four waves exchange through LDS only, so a wave has to drain its LDS
// traffic (lgkmcnt) before it signals -- not its global loads: __syncthreads() would also wait for the n_wk
// gathers issued two tokens ahead and put their full latency on every token.
#define TEAM_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

size_t mvhdp_sweep_team_bytes(int M, int S_cap)
{
    size_t b = (size_t)(64 + 64 + 16) * 4          // bitmap, prefix, wlen
             + 64                                  // queue word + padding
             + 3 * 2 * TEAM_WAVES * 8              // exchange: totals, hits, hits of the exact pass (by token parity)
             + (size_t)S_cap * 4                   // slot -> topic
             + (size_t)M * S_cap * 2               // per-view slot counts, 16-bit
             + (size_t)S_cap * 8 + 16;             // terms, for the exact sequential pass
    return (b + 15) & ~(size_t)15;
}

template <int RT, bool DEBUG>
__global__ __launch_bounds__(TEAM_LANES, (RT == 4 ? 3 : (RT == 2 ? 4 : 5))) void sweep_team_kernel(MvModel mm, SweepLaunch sl)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = uniform_i(threadIdx.x >> 6);
    const int gl = wv * 64 + lane;                          // lane of the team
    const int K = mm.K, M = mm.M, S = sl.S_cap;
    const int NW = (K + 31) >> 5;
    const bool exact_only = (sl.flags & MVHDP_SWEEP_EXACT_CHAIN) != 0;
    const bool w0 = (wv == 0);

    int* nkd = (int*)smem;                                  // [M*K] n_k deltas of this block
    unsigned int* hist_s = (unsigned int*)(nkd + M * K);
    for (int i = threadIdx.x; i < M * K + MVHDP_HIST_BINS; i += blockDim.x) nkd[i] = 0;

    unsigned char* tb = smem + sl.block_shared_bytes;
    uint32_t* bitmap = (uint32_t*)tb;
    uint32_t* prefix = bitmap + 64;
    int* wlen = (int*)(prefix + 64);
    long long* xq = (long long*)(wlen + 16);                // work-queue word
    double* xt = (double*)(xq + 8);                         // [2][4] wave totals
    long long* xh = (long long*)(xt + 2 * TEAM_WAVES);      // [2][4] first hit (low 32 bits) | near flag (bit 32)
    long long* xh2 = xh + 2 * TEAM_WAVES;                   // the same for the exact pass
    int* sk = (int*)(xh2 + 2 * TEAM_WAVES);
    unsigned short* sn = (unsigned short*)(sk + S);
    double* tl = (double*)(((uintptr_t)(sn + (size_t)M * S) + 7) & ~(uintptr_t)7);
    __syncthreads();

    const int32_t* __restrict__ nwk = mm.counts;
    const int32_t* __restrict__ nk_all = mm.counts + mm.rowbase[M] * K;
    int32_t* dnwk = mm.delta;

    unsigned int n_tok = 0, n_chg = 0, c_new = 0, c_doc = 0, c_tree = 0, n_oov = 0, n_abort = 0, n_fb = 0, n_misclass = 0;
    unsigned int par = 0;                                   // token parity: which half of the exchange buffers

    const long long q_n1 = sl.q_list_count ? (long long)*sl.q_list_count : 0;
    const long long q_total = q_n1 + sl.q_order_count;
    for (;;) {
      __syncthreads();
      if (threadIdx.x == 0) xq[0] = (long long)atomicAdd(sl.doc_counter, (unsigned long long)MVHDP_DOC_BATCH);
      __syncthreads();
      const long long q0 = xq[0];
      if (q0 >= q_total) break;
      const long long q1 = (q0 + MVHDP_DOC_BATCH < q_total) ? q0 + MVHDP_DOC_BATCH : q_total;
      for (long long q = q0; q < q1; q++) {
        int64_t d;
        if (q < q_n1) d = (int64_t)sl.q_list[q];
        else { const int64_t o = sl.q_order_start + (q - q_n1); d = sl.q_order ? (int64_t)sl.q_order[o] : o; }
        const int64_t dg = mm.doc_id_base + d;

        // ---- WRK:339-391: the entity's topics -> slot list, by all 256 lanes ----
        __syncthreads();                                    // the previous entity's LDS state is dead
        if (w0) bitmap[lane] = 0;
        __syncthreads();
        int doc_tokens = 0;
        for (int m = 0; m < M; m++) {
            const int64_t b = mm.doc_off[m][d], e = mm.doc_off[m][d + 1];
            doc_tokens += (int)(e - b);
            if (threadIdx.x == 0) wlen[m] = (int)(e - b);
            for (int64_t i = b + threadIdx.x; i < e; i += TEAM_LANES) {
                int zz = mm.z[m][i];
                if (zz >= 0) atomicOr(&bitmap[zz >> 5], 1u << (zz & 31));
            }
        }
        __syncthreads();
        int S_used;
        {
            uint32_t wbits = (lane < NW) ? bitmap[lane] : 0u;
            int cnt = __popc(wbits);
            int incl = wave_incl_scan_i(cnt, lane);
            if (w0) prefix[lane] = (uint32_t)(incl - cnt);
            S_used = bcast_i(incl, 63);
        }
        __syncthreads();
        if (sl.slot_hist && threadIdx.x == 0 && S_used > 0) atomicAdd(&hist_s[min((S_used + 63) >> 6, MVHDP_HIST_BINS) - 1], (unsigned int)doc_tokens);
        // slots per lane: 1, 2 or 4 consecutive slots (slot i = gl*R_eff + r)
        const int lg = (S_used <= 256) ? 0 : (S_used <= 512) ? 1 : 2;
        const int R_eff = 1 << lg;
        bool too_long = false;
        for (int m = 0; m < M; m++) too_long |= wlen[m] > 65535;
        if (S_used > 1024 || R_eff > RT || too_long) { if (threadIdx.x == 0) n_misclass++; continue; }
        for (int k0 = 0; k0 < K; k0 += TEAM_LANES) {
            int k = k0 + (int)threadIdx.x;
            if (k < K) {
                uint32_t w = bitmap[k >> 5];
                if ((w >> (k & 31)) & 1u) sk[prefix[k >> 5] + __popc(w & ((1u << (k & 31)) - 1u))] = k;
            }
        }
        for (int i = threadIdx.x; i < ((M * S + 1) >> 1); i += TEAM_LANES) ((unsigned int*)sn)[i] = 0;
        __syncthreads();
        for (int m = 0; m < M; m++) {
            const int64_t b = mm.doc_off[m][d], e = mm.doc_off[m][d + 1];
            for (int64_t i = b + threadIdx.x; i < e; i += TEAM_LANES) {
                int zz = mm.z[m][i];
                if (zz >= 0) {
                    uint32_t w = bitmap[zz >> 5];
                    int slot = prefix[zz >> 5] + __popc(w & ((1u << (zz & 31)) - 1u));
                    const int idx = m * S + slot;
                    atomicAdd(&((unsigned int*)sn)[idx >> 1], 1u << ((idx & 1) * 16));   // WRK:357
                }
            }
        }
        __syncthreads();
        int skr[RT], koff[RT];
#pragma unroll
        for (int r = 0; r < RT; r++) {
            const int i = gl * R_eff + r;
            skr[r] = (r < R_eff && i < S_used) ? sk[i] : (int)0x80000000;  // unused slot = removed topic 0
            koff[r] = (skr[r] & 0x7fffffff) << 2;
        }

        const double* pd = (M > 1) ? (mm.p + d * M * M) : nullptr;        // WRK:327-337
        bool aborted = false;

        for (int m = 0; m < M && !aborted; m++) {                         // WRK:393
            const int lenm = uniform_i(wlen[m]);
            if (lenm == 0) continue;
            const double beta_m = mm.beta[m];
            const double scale_m = (double)lenm + mm.gamma[m] * mm.alpha_sum[m];
            const double p_mm = pd ? pd[m * M + m] : 1.0;
            const int32_t* nk = nk_all + (int64_t)m * K;

            // per-view slot registers; WRK:395-410 totalMassOtherModalities (frozen for this view, Q3)
            int cn[RT];
            double oth[RT], den[RT];
            unsigned int onz = 0;
#pragma unroll
            for (int r = 0; r < RT; r++) {
                cn[r] = 0; oth[r] = 0.0; den[r] = 1.0;
                const int i = gl * R_eff + r;
                if (r < R_eff && i < S_used) {
                    const int k = skr[r] & 0x7fffffff;
                    cn[r] = (int)sn[m * S + i];
                    double acc = 0.0;
                    for (int j = 0; j < M; j++) {
                        if (j == m) continue;
                        const int cj = (int)sn[j * S + i];
                        if (cj != 0) onz |= 1u << r;
                        const int lj = wlen[j];
                        if (lj != 0)
                            acc += pd[m * M + j] * ((double)cj + mm.gamma[j] * mm.alpha[(int64_t)j * (K + 1) + k])
                                   / ((double)lj + mm.gamma[j] * mm.alpha_sum[j]);
                    }
                    oth[r] = acc * scale_m;
                    den[r] = (double)nk[k] + mm.beta_sum[m];
                }
            }
            // WRK:413-418 newTopicMassAllModalities
            double newAll = 0.0;
            for (int j = 0; j < M; j++) {
                double pmj = pd ? pd[m * M + j] : 1.0;
                newAll += pmj * (mm.gamma[j] * mm.alpha[(int64_t)j * (K + 1) + K]) / ((double)wlen[j] + mm.gamma[j] * mm.alpha_sum[j]);
            }
            newAll = newAll * scale_m;
            const double newMass = (mm.first_inactive < 0) ? 0.0 : newAll / (double)K;   // WRK:515

            const int64_t base = mm.doc_off[m][d];
            const int64_t row0 = mm.rowbase[m];
            const int Vm = mm.V[m];

            for (int c0 = 0; c0 < lenm && !aborted; c0 += WAVE) {
                // one lane per token of the chunk, in every wave alike: token id, old topic, its slot, RNG,
                // and the F+tree descent of the token's word (FT:118-132; see sweep_fast_kernel)
                const int ti = c0 + lane;
                const bool tvalid = ti < lenm;
                int w_l = tvalid ? mm.tok[m][base + ti] : -1;
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
                if (w_l >= Vm) w_l = -1;                                     // WRK:427-428 marks OOV
                int znew_l = z_l;
                const int nt = min(WAVE, lenm - c0);

                double root_l = 0.0;
                int zt_l = -1, st_l = -1;
                {
                    const bool act = tvalid && w_l >= 0;
                    const double* __restrict__ tr = mm.trees + (row0 + max(w_l, 0)) * 2 * K;
                    if (act) root_l = tr[1];
                    double u = u2_l * root_l;                                // FT:120
                    int i = 1;
                    while (__builtin_amdgcn_ballot_w64(act && i < K)) {      // FT:122
                        if (act && i < K) {
                            const double l = tr[2 * i];
                            if (u < l) i = 2 * i;                            // FT:124-125
                            else { u = u - l; i = 2 * i + 1; }               // FT:127-128
                        }
                    }
                    if (act) {
                        zt_l = i - K;                                        // FT:132
                        const uint32_t wbit = bitmap[zt_l >> 5];
                        if ((wbit >> (zt_l & 31)) & 1u) st_l = (int)(prefix[zt_l >> 5] + __popc(wbit & ((1u << (zt_l & 31)) - 1u)));
                    }
                }

                // n_wk values of the lane's slots, two tokens ahead in two register buffers (fixed load count per token)
                int gA[RT], gB[RT];
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    const int wa = bcast_i(w_l, min(a, nt - 1));
                    const char* __restrict__ cp = (const char*)(nwk + (row0 + max(wa, 0)) * K);
#pragma unroll
                    for (int r = 0; r < RT; r++) {
                        const int v = *(const int32_t*)(cp + koff[r]);
                        if (a == 0) gA[r] = v; else gB[r] = v;
                    }
                }

                for (int t = 0; t < nt; t += 2) {                           // WRK:425
#define TOK_T t
#define TOK_G gA
#include "mvhdp_sweep_team_token.inc"
#undef TOK_T
#undef TOK_G
                    if (aborted) break;
                    if (t + 1 < nt) {
#define TOK_T (t + 1)
#define TOK_G gB
#include "mvhdp_sweep_team_token.inc"
#undef TOK_T
#undef TOK_G
                        if (aborted) break;
                    }
                }

                // WRK:587-589 + UPD:197-218 for the whole chunk at once, by the first wave (every wave holds the same znew_l)
                if (w0) {
                    const bool chg = tvalid && (w_l >= 0) && (znew_l != z_l) && !(sl.flags & MVHDP_SWEEP_FROZEN);
                    n_chg += (unsigned int)__popcll(__ballot(chg));
                    if (chg) {
                        const int64_t rowK = (row0 + w_l) * K;
                        if (z_l >= 0) {
                            __hip_atomic_fetch_add(&dnwk[rowK + z_l], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_fetch_add(&nkd[m * K + z_l], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        __hip_atomic_fetch_add(&dnwk[rowK + znew_l], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_fetch_add(&nkd[m * K + znew_l], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (mm.first_inactive >= 0 && mm.inactive[znew_l]) {          // UPD:263
                            long long key = (dg << 34) | ((long long)m << 31) | ((long long)ti << 11) | (long long)znew_l;
                            atomicMin(sl.act_key, key);
                        }
                    }
                    if (tvalid) mm.z[m][base + ti] = znew_l;                 // coalesced write-back of the chunk
                }
            }

            // the view's counts go back to LDS: later views read them (WRK:404) and test them (WRK:445); own slots only
#pragma unroll
            for (int r = 0; r < RT; r++) {
                const int i = gl * R_eff + r;
                if (r < R_eff && i < S_used) sn[m * S + i] = (unsigned short)cn[r];
            }
            LDS_FENCE();
        }
        if (aborted && threadIdx.x == 0) n_abort++;
      }
    }

    __syncthreads();
    int32_t* dnk = mm.delta + mm.rowbase[M] * K;
    for (int i = threadIdx.x; i < M * K; i += blockDim.x)
        if (nkd[i]) atomicAdd(&dnk[i], nkd[i]);
    if (sl.slot_hist && threadIdx.x < MVHDP_HIST_BINS && hist_s[threadIdx.x]) atomicAdd(&sl.slot_hist[threadIdx.x], (unsigned long long)hist_s[threadIdx.x]);
    if (threadIdx.x == 0) {
        if (n_tok) atomicAdd(&sl.stats[ST_TOKENS], (unsigned long long)n_tok);
        if (n_chg) atomicAdd(&sl.stats[ST_CHANGED], (unsigned long long)n_chg);
        if (c_new) atomicAdd(&sl.stats[ST_NEW], (unsigned long long)c_new);
        if (c_doc) atomicAdd(&sl.stats[ST_DOC], (unsigned long long)c_doc);
        if (c_tree) atomicAdd(&sl.stats[ST_TREE], (unsigned long long)c_tree);
        if (n_oov) atomicAdd(&sl.stats[ST_OOV], (unsigned long long)n_oov);
        if (n_abort) atomicAdd(&sl.stats[ST_ABORT], (unsigned long long)n_abort);
        if (n_fb) atomicAdd(&sl.stats[ST_FALLBACK], (unsigned long long)n_fb);
        if (n_misclass) atomicAdd(&sl.stats[ST_MISCLASS], (unsigned long long)n_misclass);
    }
}

template <int RT>
static hipError_t launch_team(const MvModel& mm, const SweepLaunch& sl, int grid_blocks, bool debug, hipStream_t s)
{
    size_t lds = sl.block_shared_bytes + sl.wave_bytes;     // wave_bytes = the team's region
    if (lds > 65536) {
        hipError_t e = hipFuncSetAttribute(debug ? (const void*)sweep_team_kernel<RT, true> : (const void*)sweep_team_kernel<RT, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (debug) hipLaunchKernelGGL((sweep_team_kernel<RT, true>), dim3(grid_blocks), dim3(TEAM_LANES), lds, s, mm, sl);
    else       hipLaunchKernelGGL((sweep_team_kernel<RT, false>), dim3(grid_blocks), dim3(TEAM_LANES), lds, s, mm, sl);
    return hipGetLastError();
}

hipError_t mvhdp_launch_sweep_team(const MvModel& mm, const SweepLaunch& sl, int rt, int grid_blocks, bool debug, hipStream_t s)
{
    switch (rt) {
    case 1: return launch_team<1>(mm, sl, grid_blocks, debug, s);
    case 2: return launch_team<2>(mm, sl, grid_blocks, debug, s);
    case 4: return launch_team<4>(mm, sl, grid_blocks, debug, s);
    default: return hipErrorInvalidValue;
    }
}

template <int RT>
static int team_regs(bool debug)
{
    hipFuncAttributes a;
    if (hipFuncGetAttributes(&a, debug ? (const void*)sweep_team_kernel<RT, true> : (const void*)sweep_team_kernel<RT, false>) != hipSuccess) return 128;
    return a.numRegs;
}

// resident teams (blocks) per CU from the kernel's register count and LDS need
int mvhdp_sweep_team_occupancy(int rt, bool debug, size_t lds_bytes)
{
    int regs = (rt == 1) ? team_regs<1>(debug) : (rt == 2) ? team_regs<2>(debug) : team_regs<4>(debug);
    regs = (regs + 7) / 8 * 8;
    int waves_simd = regs > 0 ? 512 / regs : 8;
    if (waves_simd > 8) waves_simd = 8;
    if (waves_simd < 1) waves_simd = 1;
    int by_regs = waves_simd;                               // a team is one wave on each of the four SIMDs
    int by_lds = (int)((160 * 1024) / (lds_bytes > 0 ? lds_bytes : 1));
    int b = by_regs < by_lds ? by_regs : by_lds;
    if (b > 8) b = 8;
    return b < 1 ? 1 : b;
}
