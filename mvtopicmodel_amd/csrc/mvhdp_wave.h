// mvhdp_wave.h — wave-level (64-lane) device helpers shared by the gfx950 kernels:
// broadcasts, scans, Philox4x32-10 and FTree.sample on a stored tree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WAVE 64
// LDS accesses of one wave execute in issue order, so within a wave only the
// compiler has to be kept from reordering / caching LDS traffic.
#define LDS_FENCE() asm volatile("" ::: "memory")

// ---------------------------------------------------------------------------
// small wave-level helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ int uniform_i(int x) { return __builtin_amdgcn_readfirstlane(x); }

__device__ __forceinline__ int bcast_i(int v, int src_lane)
{
    return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src_lane));
}

__device__ __forceinline__ float bcast_f(float v, int src_lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), __builtin_amdgcn_readfirstlane(src_lane)));
}

// v with lane `lane` (wave-uniform) set to `val` (wave-uniform): one v_writelane_b32 instead of a compare and a select.  This clang has
// no builtin for it; the lane select goes through M0 (two different SGPR operands would exceed the constant-bus limit of the
// encoding).  M0 is not saved: the compiler treats it as reserved and only ever loads it immediately before an instruction that
// reads it (none in these kernels: tests/test_abi.py disassembles the built library and fails on any other mention of m0) -- and the asm
// declares the clobber, so a build with other flags stays sound without that test.  The scalar unit, which these moves run on, is as
// busy as the vector unit in the token loop.
__device__ __forceinline__ int wave_writelane(int v, int val, int lane)
{
    val = __builtin_amdgcn_readfirstlane(val);           // (free where the compiler knows the value to be uniform; where it does not, this is
    lane = __builtin_amdgcn_readfirstlane(lane);         // what makes it a scalar register -- the "s" constraint alone does not)
    asm volatile("s_mov_b32 m0, %2\n\t"
                 "v_writelane_b32 %0, %1, m0"
                 : "+v"(v) : "s"(val), "s"(lane) : "m0");
    return v;
}
// The address of a row of a table: base + index * row_bytes with a 32 x 32 -> 64-bit scalar multiply (two scalar instructions instead of
// the five of a 64-bit one), handed on through an empty asm so that the compiler keeps it a scalar base: the per-lane part of a gather's
// address is then the 32-bit offset operand of the load (global_load ..., v_off, s[base]) -- no vector address arithmetic, no 64-bit
// per-lane pointers held in registers.
typedef const __attribute__((address_space(1))) char* gptr_t;          // a pointer known to be to global memory (global_load, not flat_load)
__device__ __forceinline__ gptr_t scalar_row(const char* base, unsigned int index, unsigned int row_bytes)
{
    unsigned long long a = (unsigned long long)base + (unsigned long long)index * row_bytes;
    unsigned int lo = (unsigned int)a, hi = (unsigned int)(a >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));
    return (gptr_t)(((unsigned long long)hi << 32) | lo);
}
// ... and the per-lane 32-bit offset that goes with it: the empty asm keeps its zero-extension next to the load (hoisted out of the
// loop it would come back as a 64-bit register pair and a vector add per load).
// (The asm "touches" the caller's register in place -- no copy; once per token, on every path, or the two sides of a branch would
// disagree about the register and a copy would reconcile them.)
__device__ __forceinline__ void touch_lane_off(int& byte_offset) { asm volatile("" : "+v"(byte_offset)); }
__device__ __forceinline__ double uniform_d(double x)
{
    const long long b = __double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ float uniform_f(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }

__device__ __forceinline__ double bcast_d(double v, int src_lane)
{
    int s = __builtin_amdgcn_readfirstlane(src_lane);
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), s);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), s);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// inclusive prefix sums across the 64 lanes (any association order is fine:
// the fp64 one is only used under the certified-scan tolerance, see below)
__device__ __forceinline__ int wave_incl_scan_i(int v, int lane)
{
#pragma unroll
    for (int s = 1; s < WAVE; s <<= 1) {
        int o = __shfl_up(v, s, WAVE);
        if (lane >= s) v += o;
    }
    return v;
}

__device__ __forceinline__ double wave_incl_scan_d(double v, int lane)
{
#pragma unroll
    for (int s = 1; s < WAVE; s <<= 1) {
        double o = __shfl_up(v, s, WAVE);
        if (lane >= s) v += o;
    }
    return v;
}

// fp64 inclusive scan over the 64 lanes with DPP moves (no LDS crossbar): four row_shr
// steps scan each 16-lane row, row_bcast:15 / row_bcast:31 carry the row totals across
// (gfx9 DPP controls; 2 x v_mov_b32_dpp + 1 x v_add_f64 per step).
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ double dpp_src_d(double x)
{
    long long b = __double_as_longlong(x);
    int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, ROW_MASK, 0xf, BOUND);
    int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, BOUND);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// (Tried in round 3: skipping the long strides for short topic lists -- a step of stride s changes nothing below lane s -- with
// wave-uniform branches on the list size: 30.1 ms instead of 28.8 per settled C4 sweep.  The five extra scalar branches per token
// cost more than the skipped DPP steps save; the scalar unit is 64 % busy in this kernel, the vector unit 94 %.
// Also tried: the mirror's saturation test without the slot test, the removed-slot test folded into the count (one select instead of
// two), the new-topic comparisons skipped when there are no inactive topics -- the compiler answered with MORE vector instructions
// (107 instead of 102.5 per token, SQ_INSTS_VALU) and the kernel was 1 % slower; source-level trimming of this loop has run out.)
__device__ __forceinline__ double wave_incl_scan_d_dpp(double v)
{
    v += dpp_src_d<0x111, 0xf, true>(v);    // row_shr:1
    v += dpp_src_d<0x112, 0xf, true>(v);    // row_shr:2
    v += dpp_src_d<0x114, 0xf, true>(v);    // row_shr:4
    v += dpp_src_d<0x118, 0xf, true>(v);    // row_shr:8
    v += dpp_src_d<0x142, 0xa, false>(v);   // row_bcast:15 -> rows 1 and 3
    v += dpp_src_d<0x143, 0xc, false>(v);   // row_bcast:31 -> rows 2 and 3
    return v;
}


// The same scan in fp32: DPP is a 32-bit facility, so a step is ONE v_add_f32 with a DPP source (three instructions for fp64), and
// fp32 issues at twice the fp64 rate.  Used by the fp32 screening of the token loop (mvhdp_sweep_fast_token.inc).
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_src_f(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, BOUND));
}

__device__ __forceinline__ float wave_incl_scan_f_dpp(float v)
{
    v += dpp_src_f<0x111, 0xf, true>(v);    // row_shr:1   (the compiler fuses these four into v_add_f32_dpp)
    v += dpp_src_f<0x112, 0xf, true>(v);    // row_shr:2
    v += dpp_src_f<0x114, 0xf, true>(v);    // row_shr:4
    v += dpp_src_f<0x118, 0xf, true>(v);    // row_shr:8
    // row_bcast:15 -> rows 1 and 3, row_bcast:31 -> rows 2 and 3: with a row mask the rows that are not named keep their value, which
    // is exactly the scan step -- one instruction each; written out because the compiler does not fuse a masked DPP move (it emits a
    // zero move, the DPP move and an add).  The s_nop are the wait states a DPP read of a just-written VGPR needs: the hazard
    // recogniser does not look inside inline assembly.
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
                 : "+v"(v));
    return v;
}

// IEEE-754 correctly rounded n/d for operands that need no rescaling: exactly the
// Newton/fma sequence hipcc emits for an fp64 divide (v_rcp_f64, two refinements,
// quotient, residual, final fma) without v_div_scale / v_div_fixup, which are the
// identity when n, d and n/d are normal numbers far from the exponent limits.  Here
// n = count + beta in [1e-4, 2^31] and d = n_k + betaSum in [1e-4, 2^32].
__device__ __forceinline__ double div_inrange(double n, double d)
{
    double x = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, x, 1.0);
    x = __builtin_fma(x, e, x);
    e = __builtin_fma(-d, x, 1.0);
    x = __builtin_fma(x, e, x);
    double q = n * x;
    double r = __builtin_fma(-d, q, n);
    return __builtin_fma(r, x, q);
}

// ---------------------------------------------------------------------------
// Philox4x32-10 (Random123); the stream contract is in DESIGN.md §RNG
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double bits_to_unit(uint32_t hi, uint32_t lo)
{
    // the 53-bit shape of ThreadLocalRandom.nextDouble() (WRK:517,534)
    unsigned long long x = ((unsigned long long)hi << 32) | lo;
    return (double)(x >> 11) * 0x1.0p-53;
}

// ---------------------------------------------------------------------------
// FTree.sample (FT:111-136) against the stored tree of one (view,type).
// The descent reads whole sub-trees per round: the 62 nodes of the five levels
// below the current node sit in 5 contiguous runs of the tree array, one node
// per lane, so a K<=2048 descent needs at most 3 dependent load rounds instead
// of log2(K) of them.  All lanes walk the same path (u is wave-uniform).
// ---------------------------------------------------------------------------
__device__ __forceinline__ int tree_sample(const double* __restrict__ tree, int K, double u2, double root, int lane)
{
    int i = 1;
    double u = u2 * root;                                  // FT:120  u = u * tree[1]
    const int j = lane + 2;
    const int t = 31 - __clz(j);                           // 1..6 (6 only for lanes 62,63: unused)
    const int o = j - (1 << t);
    while (i < K) {                                        // FT:122
        long long idx = ((long long)i << t) + o;
        double v = (lane < 62 && idx < 2LL * K) ? tree[idx] : 0.0;
        int rel_t = 0, rel_o = 0;
#pragma unroll
        for (int step = 0; step < 5; step++) {
            if (i < K) {
                int src = (2 << rel_t) + 2 * rel_o - 2;   // lane holding tree[2*i]
                double l = bcast_d(v, src);
                if (u < l) { i = 2 * i; rel_o = 2 * rel_o; }               // FT:124-125
                else { u = u - l; i = 2 * i + 1; rel_o = 2 * rel_o + 1; }  // FT:127-128
                rel_t++;
            }
        }
        i = uniform_i(i);
    }
    return i - K;                                          // FT:132
}


// Same descent, but the first round (tree[1..63] of the word, one node per lane: lane l holds
// tree[l+1]) was loaded ahead of time together with the token's n_wk gather, so a K<=2048 descent
// pays one dependent load round instead of two.  Lane 0 of `first` is tree[1], the root (FT:120).
__device__ __forceinline__ int tree_sample_preloaded(const double* __restrict__ tree, int K, double u2, double first, int lane)
{
    int i = 1;
    double u = u2 * bcast_d(first, 0);                     // FT:120
    {
        int rel_t = 0, rel_o = 0;
#pragma unroll
        for (int step = 0; step < 5; step++) {
            if (i < K) {
                int src = (2 << rel_t) + 2 * rel_o - 1;   // lane holding tree[2*i] (node n sits in lane n-1)
                double l = bcast_d(first, src);
                if (u < l) { i = 2 * i; rel_o = 2 * rel_o; }
                else { u = u - l; i = 2 * i + 1; rel_o = 2 * rel_o + 1; }
                rel_t++;
            }
        }
        i = uniform_i(i);
    }
    const int j = lane + 2;
    const int t = 31 - __clz(j);
    const int o = j - (1 << t);
    while (i < K) {
        long long idx = ((long long)i << t) + o;
        double v = (lane < 62 && idx < 2LL * K) ? tree[idx] : 0.0;
        int rel_t = 0, rel_o = 0;
#pragma unroll
        for (int step = 0; step < 5; step++) {
            if (i < K) {
                int src = (2 << rel_t) + 2 * rel_o - 2;
                double l = bcast_d(v, src);
                if (u < l) { i = 2 * i; rel_o = 2 * rel_o; }
                else { u = u - l; i = 2 * i + 1; rel_o = 2 * rel_o + 1; }
                rel_t++;
            }
        }
        i = uniform_i(i);
    }
    return i - K;
}
