/*
 * mvhdp_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 * See mvhdp_oracle.h for the scope statement.  Plain C; build with
 * -ffp-contract=off so that no a*b+c is fused (Java never fuses).
 *
 * Reference aliases (all under /root/reference/src/main/java/org/madgik/):
 *   WRK = MVTopicModel/FastQMVWVWorkerRunnable.java
 *   UPD = MVTopicModel/FastQMVWVUpdaterRunnable.java
 *   PTM = MVTopicModel/FastQMVWVParallelTopicModel.java
 *   FT  = utils/FTree.java      QD = utils/FastQDelta.java
 * Third-party arithmetic (cc.mallet:mallet:2.0.8, class files only in the
 * reference) is restated from its published algorithm: Randoms.nextUniform /
 * nextGaussian / nextBeta, and java.util.Random from its documented contract.
 */
#include "mvhdp_oracle.h"
#include "mvhdp_oracle_internal.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================== */
/* Philox4x32-10 (Random123)                                                 */
/* ======================================================================== */

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline double bits_to_unit(uint32_t hi, uint32_t lo)
{
    /* same 53-bit shape as ThreadLocalRandom.nextDouble() (WRK:517,534) */
    uint64_t x = ((uint64_t)hi << 32) | lo;
    return (double)(x >> 11) * 0x1.0p-53;
}

void orc_token_uniforms(uint64_t seed, uint32_t sweep, int64_t doc, int view, uint32_t pos,
                        double* u1, double* u2)
{
    uint32_t ctr[4] = { pos, (uint32_t)view, (uint32_t)doc, sweep };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)((uint64_t)doc >> 32) };
    uint32_t x[4];
    orc_philox4x32_10(ctr, key, x);
    *u1 = bits_to_unit(x[0], x[1]);
    *u2 = bits_to_unit(x[2], x[3]);
}

/* ======================================================================== */
/* java.util.Random + MALLET Randoms                                         */
/* ======================================================================== */

void orc_jrand_seed(orc_jrand* r, int64_t seed)
{
    r->s = ((uint64_t)seed ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1);
    r->have_gauss = 0;
    r->next_gauss = 0.0;
}

int32_t orc_jrand_next(orc_jrand* r, int bits)
{
    r->s = (r->s * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
    return (int32_t)(uint32_t)(r->s >> (48 - bits));
}

int32_t orc_jrand_next_int(orc_jrand* r, int32_t bound)
{
    /* java.util.Random.nextInt(int), used by PTM:500,503,506 */
    int32_t rr = orc_jrand_next(r, 31);
    int32_t m = bound - 1;
    if ((bound & m) == 0) {
        rr = (int32_t)(((int64_t)bound * (int64_t)rr) >> 31);
    } else {
        int32_t u = rr;
        for (;;) {
            rr = u % bound;
            /* Java int arithmetic wraps */
            int32_t chk = (int32_t)((uint32_t)u - (uint32_t)rr + (uint32_t)m);
            if (chk >= 0) break;
            u = orc_jrand_next(r, 31);
        }
    }
    return rr;
}

double orc_mallet_next_uniform(orc_jrand* r)
{
    int64_t hi = orc_jrand_next(r, 26);
    int64_t lo = orc_jrand_next(r, 27);
    int64_t l = (hi << 27) + lo;
    return (double)l / (double)(1LL << 53);
}

double orc_mallet_next_gaussian(orc_jrand* r)
{
    if (!r->have_gauss) {
        double v1 = orc_mallet_next_uniform(r), v2 = orc_mallet_next_uniform(r);
        double x1 = sqrt(-2 * log(v1)) * cos(2 * M_PI * v2);
        double x2 = sqrt(-2 * log(v1)) * sin(2 * M_PI * v2);
        r->next_gauss = x2;
        r->have_gauss = 1;
        return x1;
    } else {
        r->have_gauss = 0;
        return r->next_gauss;
    }
}

/* Generic over the uniform source so that the LCG stream (reference worker)
 * and the Philox stream (device contract) run the very same algorithm. */
typedef struct {
    int kind;               /* 0 = LCG, 1 = Philox */
    orc_jrand* lcg;
    uint32_t ctr1, ctr2, ctr3; uint32_t key[2];
    uint32_t n;             /* Philox: next uniform index */
    int have_gauss; double next_gauss;
} usrc;

static double usrc_uniform(usrc* s)
{
    if (s->kind == 0) return orc_mallet_next_uniform(s->lcg);
    uint32_t ctr[4] = { s->n >> 1, s->ctr1, s->ctr2, s->ctr3 };
    uint32_t x[4];
    orc_philox4x32_10(ctr, s->key, x);
    double u = (s->n & 1) ? bits_to_unit(x[2], x[3]) : bits_to_unit(x[0], x[1]);
    s->n++;
    return u;
}

static double usrc_gaussian(usrc* s)
{
    if (s->kind == 0) return orc_mallet_next_gaussian(s->lcg);
    if (!s->have_gauss) {
        double v1 = usrc_uniform(s), v2 = usrc_uniform(s);
        double x1 = sqrt(-2 * log(v1)) * cos(2 * M_PI * v2);
        double x2 = sqrt(-2 * log(v1)) * sin(2 * M_PI * v2);
        s->next_gauss = x2;
        s->have_gauss = 1;
        return x1;
    } else {
        s->have_gauss = 0;
        return s->next_gauss;
    }
}

/* MALLET Randoms.nextBeta(alpha, beta).  Note the reference quirk kept here:
 * for alpha>=1, beta==1 the term B*log((1-x)/B) is 0*inf = NaN, the rejection
 * comparison is false, and the first in-range Gaussian proposal is returned. */
static double usrc_beta(usrc* s, double alpha, double beta)
{
    if (alpha == 1 && beta == 1) {
        return usrc_uniform(s);
    } else if (alpha >= 1 && beta >= 1) {
        double A = alpha - 1, B = beta - 1, C = A + B, L = C * log(C),
               mu = A / C, sigma = 0.5 / sqrt(C);
        double y = usrc_gaussian(s), x = sigma * y + mu;
        while (x < 0 || x > 1) {
            y = usrc_gaussian(s);
            x = sigma * y + mu;
        }
        double u = usrc_uniform(s);
        while (log(u) >= A * log(x / A) + B * log((1 - x) / B) + L + 0.5 * y * y) {
            y = usrc_gaussian(s);
            x = sigma * y + mu;
            while (x < 0 || x > 1) {
                y = usrc_gaussian(s);
                x = sigma * y + mu;
            }
            u = usrc_uniform(s);
        }
        return x;
    } else {
        double v1 = pow(usrc_uniform(s), 1 / alpha), v2 = pow(usrc_uniform(s), 1 / beta);
        while (v1 + v2 > 1) {
            v1 = pow(usrc_uniform(s), 1 / alpha);
            v2 = pow(usrc_uniform(s), 1 / beta);
        }
        return v1 / (v1 + v2);
    }
}

double orc_mallet_next_beta(orc_jrand* r, double a, double b)
{
    usrc s; memset(&s, 0, sizeof s); s.kind = 0; s.lcg = r;
    return usrc_beta(&s, a, b);
}

int64_t orc_java_round(double x)
{
    /* Java 8 Math.round: floor(x + 1/2) computed without the double-rounding bug */
    if (isnan(x)) return 0;
    double f = floor(x);
    double d = x - f;
    if (d >= 0.5) f += 1.0;
    return (int64_t)f;
}

/* ======================================================================== */
/* FTree (FT)                                                                */
/* ======================================================================== */

void orc_ftree_construct(double* tree, int size, const double* weights)
{
    /* FT:96-109 */
    for (int i = 0; i < 2 * size; i++) tree[i] = 0;
    for (int i = 2 * size - 1; i > 0; --i) {
        if (i >= size) tree[i] = weights[i - size];
        else tree[i] = tree[2 * i] + tree[2 * i + 1];
    }
}

int orc_ftree_sample(const double* tree, int size, double u)
{
    /* FT:111-136 */
    if (u > 1) return -2; /* IllegalArgumentException */
    int i = 1;
    u = u * tree[i];
    while (i < size) {
        if (u < tree[2 * i]) {
            i = 2 * i;
        } else {
            u = u - tree[2 * i];
            i = 2 * i + 1;
        }
    }
    return i - size;
}

void orc_ftree_update(double* tree, int size, int topic, double new_value)
{
    /* FT:138-147 */
    int i = topic + size;
    double delta = new_value - tree[i];
    while (i > 0) {
        tree[i] += delta;
        i = i / 2;
    }
}

int orc_lower_bound(const double* arr, double key, int len)
{
    /* WRK:257-277 (note: reads arr[mid] with mid = (0 + -1)/2 = 0 when len==0,
     * exactly like the Java, callers never pass len==0 on the live path) */
    int lo = 0;
    int hi = len - 1;
    int mid = (lo + hi) / 2;
    for (;;) {
        if (arr[mid] >= key) {
            hi = mid - 1;
            if (hi < lo) return mid;
        } else {
            lo = mid + 1;
            if (hi < lo) return mid < len - 1 ? mid + 1 : -1;
        }
        mid = (lo + hi) / 2;
    }
}

/* ======================================================================== */
/* model                                                                     */
/* ======================================================================== */

/* struct orc_model lives in mvhdp_oracle_internal.h (shared with ref_threaded.c) */

orc_model* orc_create(int K, int M, const int32_t* V)
{
    if (K <= 0 || M <= 0 || M > ORC_MAX_M) return NULL;
    orc_model* o = (orc_model*)calloc(1, sizeof *o);
    o->K = K; o->M = M;
    o->rowbase[0] = 0;
    for (int m = 0; m < M; m++) { o->V[m] = V[m]; o->rowbase[m + 1] = o->rowbase[m] + V[m]; }
    int64_t sumV = o->rowbase[M];
    o->nwk = (int32_t*)calloc((size_t)sumV * K, sizeof(int32_t));
    o->nk = (int32_t*)calloc((size_t)M * K, sizeof(int32_t));
    o->trees = (double*)calloc((size_t)sumV * 2 * K, sizeof(double));
    o->alpha = (double*)calloc((size_t)M * (K + 1), sizeof(double));
    o->inactive = (uint8_t*)calloc((size_t)K, 1);
    /* PTM:207-214 defaults are set by the caller through orc_set_hyper */
    return o;
}

void orc_destroy(orc_model* o)
{
    if (!o) return;
    for (int m = 0; m < o->M; m++) { free(o->doc_off[m]); free(o->tokens[m]); free(o->z[m]); }
    free(o->nwk); free(o->nk); free(o->trees); free(o->alpha); free(o->inactive);
    free(o);
}

int orc_set_corpus(orc_model* o, int m, int64_t D, const int64_t* doc_off, const int32_t* tokens)
{
    if (m < 0 || m >= o->M) return -1;
    if (m > 0 && o->doc_off[0] && D != o->D) return -2;
    o->D = D;
    free(o->doc_off[m]); free(o->tokens[m]); free(o->z[m]);
    int64_t N = doc_off[D];
    o->N[m] = N;
    o->doc_off[m] = (int64_t*)malloc((size_t)(D + 1) * sizeof(int64_t));
    memcpy(o->doc_off[m], doc_off, (size_t)(D + 1) * sizeof(int64_t));
    o->tokens[m] = (int32_t*)malloc((size_t)(N > 0 ? N : 1) * sizeof(int32_t));
    memcpy(o->tokens[m], tokens, (size_t)N * sizeof(int32_t));
    o->z[m] = (int32_t*)malloc((size_t)(N > 0 ? N : 1) * sizeof(int32_t));
    for (int64_t i = 0; i < N; i++) o->z[m][i] = -1; /* UNASSIGNED_TOPIC PTM:63 */
    return 0;
}

int orc_set_assignments(orc_model* o, int m, const int32_t* z)
{
    if (m < 0 || m >= o->M || !o->z[m]) return -1;
    memcpy(o->z[m], z, (size_t)o->N[m] * sizeof(int32_t));
    return 0;
}

int orc_get_assignments(const orc_model* o, int m, int32_t* z)
{
    if (m < 0 || m >= o->M || !o->z[m]) return -1;
    memcpy(z, o->z[m], (size_t)o->N[m] * sizeof(int32_t));
    return 0;
}

void orc_set_hyper(orc_model* o, const double* alpha, const double* alpha_sum,
                   const double* beta, const double* beta_sum, const double* gamma,
                   const double* p_a, const double* p_b, const uint8_t* inactive)
{
    int M = o->M, K = o->K;
    memcpy(o->alpha, alpha, (size_t)M * (K + 1) * sizeof(double));
    for (int m = 0; m < M; m++) {
        o->alpha_sum[m] = alpha_sum[m]; o->beta[m] = beta[m];
        o->beta_sum[m] = beta_sum[m];   o->gamma[m] = gamma[m];
        for (int j = 0; j < M; j++) { o->p_a[m][j] = p_a[m * M + j]; o->p_b[m][j] = p_b[m * M + j]; }
    }
    if (inactive) memcpy(o->inactive, inactive, (size_t)K);
    else memset(o->inactive, 0, (size_t)K);
}

void orc_get_alpha(const orc_model* o, double* alpha)
{
    memcpy(alpha, o->alpha, (size_t)o->M * (o->K + 1) * sizeof(double));
}

void orc_get_inactive(const orc_model* o, uint8_t* inactive)
{
    memcpy(inactive, o->inactive, (size_t)o->K);
}

void orc_init_assignments(orc_model* o, int64_t seed)
{
    /* PTM:465-515 with previousModel == null; `random` = Randoms(randomSeed) PTM:404-408 */
    orc_jrand r; orc_jrand_seed(&r, seed);
    int K = o->K, M = o->M;
    int32_t* active = NULL; int64_t cap = 0, n_active = 0;
    for (int64_t d = 0; d < o->D; d++) {
        for (int m = 0; m < M; m++) {
            if (m == 0) n_active = 0;                                /* PTM:470-472 */
            if (!o->doc_off[m]) continue;
            int64_t b = o->doc_off[m][d], e = o->doc_off[m][d + 1];
            for (int64_t i = b; i < e; i++) {
                int topic;
                if (m == 0) {
                    topic = orc_jrand_next_int(&r, K);               /* PTM:500 */
                    if (n_active == cap) { cap = cap ? 2 * cap : 256; active = (int32_t*)realloc(active, (size_t)cap * sizeof(int32_t)); }
                    active[n_active++] = topic;                      /* PTM:501 */
                } else if (n_active > 0) {
                    int ind = orc_jrand_next_int(&r, (int32_t)n_active); /* PTM:503 */
                    topic = active[ind];
                } else {
                    topic = orc_jrand_next_int(&r, K);               /* PTM:506 */
                }
                o->z[m][i] = topic;
            }
        }
    }
    free(active);
}

void orc_build_counts(orc_model* o)
{
    /* PTM:600-652 */
    int K = o->K, M = o->M;
    memset(o->nwk, 0, (size_t)o->rowbase[M] * K * sizeof(int32_t));
    memset(o->nk, 0, (size_t)M * K * sizeof(int32_t));
    for (int m = 0; m < M; m++) {
        if (!o->doc_off[m]) continue;
        for (int64_t i = 0; i < o->N[m]; i++) {
            int topic = o->z[m][i];
            if (topic == -1) continue;                               /* PTM:634 */
            int type = o->tokens[m][i];
            /* A type outside the alphabet cannot occur here in the reference (it would index past typeTopicCounts,
             * PTM:643); it only exists at inference, where the counts are given.  Both this oracle and the
             * library leave such a token out of the counts. */
            if (type < 0 || type >= o->V[m] || topic < 0 || topic >= K) continue;
            o->nk[(size_t)m * K + topic]++;                          /* PTM:640 */
            o->nwk[(size_t)(o->rowbase[m] + type) * K + topic]++;    /* PTM:643 */
        }
    }
}

void orc_build_trees(orc_model* o)
{
    /* PTM:2660-2696 (useVectorsLambda == 0 path) */
    int K = o->K, M = o->M;
    double* temp = (double*)malloc((size_t)K * sizeof(double));
    int any_inactive = 0;
    for (int k = 0; k < K; k++) any_inactive |= o->inactive[k];
    for (int m = 0; m < M; m++) {
        for (int w = 0; w < o->V[m]; ++w) {
            const int32_t* cnt = o->nwk + (size_t)(o->rowbase[m] + w) * K;
            for (int t = 0; t < K; t++) {
                if (any_inactive && o->inactive[t]) {
                    temp[t] = 0;
                } else {
                    double p_wt = (cnt[t] + o->beta[m]) / (o->nk[(size_t)m * K + t] + o->beta_sum[m]);
                    temp[t] = o->gamma[m] * o->alpha[(size_t)m * (K + 1) + t] * p_wt;
                }
            }
            orc_ftree_construct(o->trees + (size_t)(o->rowbase[m] + w) * 2 * K, K, temp);
        }
    }
    free(temp);
}

void orc_build_inference_trees(orc_model* o)
{
    /* FastQMVWVTopicInferencer.initInferencer INF:557-586: leaves are p_wt alone (no gamma*alpha, no inactive test) */
    int K = o->K, M = o->M;
    double* temp = (double*)malloc((size_t)K * sizeof(double));
    for (int m = 0; m < M; m++)
        for (int w = 0; w < o->V[m]; ++w) {
            const int32_t* cnt = o->nwk + (size_t)(o->rowbase[m] + w) * K;
            for (int t = 0; t < K; t++) temp[t] = (cnt[t] + o->beta[m]) / (o->nk[(size_t)m * K + t] + o->beta_sum[m]);   /* INF:576 */
            orc_ftree_construct(o->trees + (size_t)(o->rowbase[m] + w) * 2 * K, K, temp);
        }
    free(temp);
}

void orc_init_assignments_from_trees(orc_model* o, uint64_t seed, int64_t doc_id_base)
{
    /* INF:169-199: every in-vocabulary token starts at trees[m][type].sample(u); out-of-vocabulary tokens keep the
     * 0 of `new int[]`.  u stands in for ThreadLocalRandom: u1 of the token stream with sweep index 0xFFFFFFFF. */
    for (int m = 0; m < o->M; m++)
        for (int64_t d = 0; d < o->D; d++) {
            int64_t b = o->doc_off[m][d], e = o->doc_off[m][d + 1];
            for (int64_t i = b; i < e; i++) {
                int type = o->tokens[m][i];
                if (type >= 0 && type < o->V[m]) {
                    double u1, u2;
                    orc_token_uniforms(seed, 0xFFFFFFFFu, doc_id_base + d, m, (uint32_t)(i - b), &u1, &u2);
                    o->z[m][i] = orc_ftree_sample(o->trees + (size_t)(o->rowbase[m] + type) * 2 * o->K, o->K, u1);
                } else {
                    o->z[m][i] = 0;
                }
            }
        }
}

void orc_get_counts(const orc_model* o, int m, int32_t* nwk, int32_t* nk)
{
    int K = o->K;
    if (nwk) memcpy(nwk, o->nwk + (size_t)o->rowbase[m] * K, (size_t)o->V[m] * K * sizeof(int32_t));
    if (nk) memcpy(nk, o->nk + (size_t)m * K, (size_t)K * sizeof(int32_t));
}

void orc_set_counts(orc_model* o, int m, const int32_t* nwk, const int32_t* nk)
{
    int K = o->K;
    if (nwk) memcpy(o->nwk + (size_t)o->rowbase[m] * K, nwk, (size_t)o->V[m] * K * sizeof(int32_t));
    if (nk) memcpy(o->nk + (size_t)m * K, nk, (size_t)K * sizeof(int32_t));
}

void orc_get_tree(const orc_model* o, int m, int w, double* tree2K)
{
    memcpy(tree2K, o->trees + (size_t)(o->rowbase[m] + w) * 2 * o->K, (size_t)2 * o->K * sizeof(double));
}

void orc_get_doc_topic_hist(const orc_model* o, int m, int32_t* hist, int32_t hist_len,
                            int32_t* doc_len_counts, int32_t len_len)
{
    /* PTM:620-651: for every entity that has view m (here: non-empty span),
     * docLengthCounts[m][len]++ and topicDocCounts[m][k][n_dk]++ for ALL k. */
    int K = o->K;
    if (hist) memset(hist, 0, (size_t)K * hist_len * sizeof(int32_t));
    if (doc_len_counts) memset(doc_len_counts, 0, (size_t)len_len * sizeof(int32_t));
    int32_t* local = (int32_t*)calloc((size_t)K, sizeof(int32_t));
    for (int64_t d = 0; d < o->D; d++) {
        int64_t b = o->doc_off[m][d], e = o->doc_off[m][d + 1];
        if (e == b) continue;
        if (doc_len_counts && e - b < len_len) doc_len_counts[e - b]++;
        memset(local, 0, (size_t)K * sizeof(int32_t));
        for (int64_t i = b; i < e; i++) if (o->z[m][i] != -1) local[o->z[m][i]]++;
        if (hist) for (int k = 0; k < K; k++) if (local[k] < hist_len) hist[(size_t)k * hist_len + local[k]]++;
    }
    free(local);
}

/* ------------------------------------------------------------------------ */
/* view weights p (WRK:327-337)                                              */
/* ------------------------------------------------------------------------ */

static void fill_p_for_doc(const orc_model* o, usrc* s_lcg, uint64_t seed, uint32_t sweep,
                           int64_t doc_global, double* p /* M*M */)
{
    int M = o->M;
    for (int m = 0; m < M; m++) {
        for (int j = m; j < M; j++) {
            double pRand;
            if (m == j) pRand = 1.0;
            else if (o->p_a[m][j] == 0) pRand = 0;
            else {
                double b;
                if (s_lcg) {
                    b = usrc_beta(s_lcg, o->p_a[m][j], o->p_b[m][j]);
                } else {
                    usrc s; memset(&s, 0, sizeof s); s.kind = 1;
                    s.ctr1 = 0x100u + (uint32_t)(m * M + j);
                    s.ctr2 = (uint32_t)doc_global; s.ctr3 = sweep;
                    s.key[0] = (uint32_t)seed;
                    s.key[1] = (uint32_t)(seed >> 32) ^ (uint32_t)((uint64_t)doc_global >> 32);
                    b = usrc_beta(&s, o->p_a[m][j], o->p_b[m][j]);
                }
                pRand = (double)orc_java_round(1000 * b) / (double)1000;   /* WRK:333 */
            }
            p[m * M + j] = (j != 0 && o->beta[j] == 0.0001) ? 0 : pRand;   /* WRK:335 */
            p[j * M + m] = (m != 0 && o->beta[m] == 0.0001) ? 0 : pRand;   /* WRK:336 */
        }
    }
}

void orc_draw_p_mallet(const orc_model* o, orc_jrand* r, double* p)
{
    usrc s; memset(&s, 0, sizeof s); s.kind = 0; s.lcg = r;
    for (int64_t d = 0; d < o->D; d++) fill_p_for_doc(o, &s, 0, 0, d, p + (size_t)d * o->M * o->M);
}

void orc_draw_p_philox(const orc_model* o, uint64_t seed, uint32_t sweep, int64_t doc_id_base, double* p)
{
    for (int64_t d = 0; d < o->D; d++)
        fill_p_for_doc(o, NULL, seed, sweep, doc_id_base + d, p + (size_t)d * o->M * o->M);
}

/* ------------------------------------------------------------------------ */
/* the sweep                                                                 */
/* ------------------------------------------------------------------------ */

typedef struct { int32_t oldT, newT, type, mod; int64_t key; } delta_t; /* QD:12-36 (doc counts not needed: histograms are recomputed) */

typedef struct {
    delta_t* v; size_t n, cap;
} delta_vec;

static void dv_push(delta_vec* dv, delta_t d)
{
    if (dv->n == dv->cap) { dv->cap = dv->cap ? 2 * dv->cap : 1024; dv->v = (delta_t*)realloc(dv->v, dv->cap * sizeof(delta_t)); }
    dv->v[dv->n++] = d;
}

/* WRK:301-601 for one entity against the snapshot (n_wk, n_k, trees).
 * Returns 0, or 1 if the Java would have thrown inside the doc (Q11). */
static int sample_one_doc(orc_model* o, int64_t d, int64_t doc_global, uint32_t sweep, uint64_t seed,
                          const double* p /* M*M */, int first_inactive, orc_stats* st, delta_vec* dv,
                          double* const* tok_dbg,
                          int n_trace, const int64_t* trace_doc, const int32_t* trace_view,
                          const int32_t* trace_pos, double* trace_out,
                          /* scratch, all sized K / M*K: */
                          int32_t* localTopicCounts, int32_t* localTopicIndex,
                          double* topicDocWordMasses, double* totalMassOtherModalities)
{
    const int K = o->K, M = o->M;
    int docLength[ORC_MAX_M];
    memset(localTopicCounts, 0, (size_t)M * K * sizeof(int32_t));

    for (int m = 0; m < M; m++) {                                            /* WRK:327-361 */
        docLength[m] = 0;
        if (o->doc_off[m]) {
            int64_t b = o->doc_off[m][d], e = o->doc_off[m][d + 1];
            docLength[m] = (int)(e - b);
            for (int64_t i = b; i < e; i++) {
                if (o->z[m][i] == -1) continue;
                localTopicCounts[(size_t)m * K + o->z[m][i]]++;
            }
        }
    }

    int denseIndex = 0;                                                       /* WRK:376-391 */
    for (int topic = 0; topic < K; topic++) {
        int i = 0, found = 0;
        while (i < M && !found) {
            if (localTopicCounts[(size_t)i * K + topic] != 0) {
                localTopicIndex[denseIndex] = topic;
                denseIndex++;
                found = 1;
            }
            i++;
        }
    }
    int nonZeroTopics = denseIndex;

    for (int m = 0; m < M; m++) {                                             /* WRK:393 */
        for (int k = 0; k < K; k++) totalMassOtherModalities[k] = 0;          /* WRK:395 */
        for (denseIndex = 0; denseIndex < nonZeroTopics; denseIndex++) {      /* WRK:399-410 */
            int topic = localTopicIndex[denseIndex];
            for (int i = 0; i < M; i++) {
                if (i != m && docLength[i] != 0) {
                    totalMassOtherModalities[topic] += p[m * M + i]
                        * (localTopicCounts[(size_t)i * K + topic] + o->gamma[i] * o->alpha[(size_t)i * (K + 1) + topic])
                        / (docLength[i] + (double)o->gamma[i] * o->alpha_sum[i]);
                }
            }
            totalMassOtherModalities[topic] = totalMassOtherModalities[topic]
                * (docLength[m] + (double)o->gamma[m] * o->alpha_sum[m]);
        }
        double newTopicMassAllModalities = 0;                                 /* WRK:413-418 */
        for (int i = 0; i < M; i++) {
            newTopicMassAllModalities += p[m * M + i] * (o->gamma[i] * o->alpha[(size_t)i * (K + 1) + K])
                / (docLength[i] + (double)o->gamma[i] * o->alpha_sum[i]);
        }
        newTopicMassAllModalities = newTopicMassAllModalities * (docLength[m] + (double)o->gamma[m] * o->alpha_sum[m]);

        if (docLength[m] == 0) continue;
        const int64_t base = o->doc_off[m][d];
        for (int position = 0; position < docLength[m]; position++) {         /* WRK:425 */
            int type = o->tokens[m][base + position];
            if (type >= o->V[m]) { st->oov_skipped++; continue; }             /* WRK:427-428 */
            int oldTopic = o->z[m][base + position];
            const int32_t* currentTypeTopicCounts = o->nwk + (size_t)(o->rowbase[m] + type) * K;
            const double* currentTree = o->trees + (size_t)(o->rowbase[m] + type) * 2 * K;

            if (oldTopic != -1) {                                             /* WRK:434-471 */
                localTopicCounts[(size_t)m * K + oldTopic]--;
                int isDeletedTopic = localTopicCounts[(size_t)m * K + oldTopic] == 0;
                int jj = 0;
                while (isDeletedTopic && jj < M) {
                    isDeletedTopic = localTopicCounts[(size_t)jj * K + oldTopic] == 0;
                    jj++;
                }
                if (isDeletedTopic) {
                    denseIndex = 0;
                    while (localTopicIndex[denseIndex] != oldTopic) {
                        denseIndex++;
                        if (denseIndex >= K) return 1; /* ArrayIndexOutOfBounds -> Q11 */
                    }
                    while (denseIndex < nonZeroTopics) {
                        if (denseIndex < K - 1) localTopicIndex[denseIndex] = localTopicIndex[denseIndex + 1];
                        denseIndex++;
                    }
                    nonZeroTopics--;
                }
            }

            int newTopic = -1;
            double topicDocWordMass = 0.0;                                    /* WRK:496-513 */
            for (denseIndex = 0; denseIndex < nonZeroTopics; denseIndex++) {
                int topic = localTopicIndex[denseIndex];
                int n = localTopicCounts[(size_t)m * K + topic];
                double p_wt = (currentTypeTopicCounts[topic] + o->beta[m]) / (o->nk[(size_t)m * K + topic] + o->beta_sum[m]);
                topicDocWordMass += (p[m * M + m] * n + totalMassOtherModalities[topic]) * p_wt;
                topicDocWordMasses[denseIndex] = topicDocWordMass;
            }
            double newTopicMass = (first_inactive < 0) ? 0 : newTopicMassAllModalities / K; /* WRK:515 */

            double nextUniform, nextUniform2;
            orc_token_uniforms(seed, sweep, doc_global, m, (uint32_t)position, &nextUniform, &nextUniform2);
            double sample = nextUniform * (newTopicMass + topicDocWordMass + currentTree[1]); /* WRK:519 */

            if (tok_dbg && tok_dbg[m]) {
                double* g = tok_dbg[m] + (size_t)(base + position) * 4;
                g[0] = newTopicMass; g[1] = topicDocWordMass; g[2] = currentTree[1]; g[3] = sample;
            }
            for (int t = 0; t < n_trace; t++) {
                if (trace_doc[t] == d && trace_view[t] == m && trace_pos[t] == position) {
                    /* full conditional: P(k) ∝ [k==firstInactive]*newMass
                     *   + 1[k in dense]*term_k + leaf_k ; slot K holds newMass itself. */
                    double* out = trace_out + (size_t)t * (K + 1);
                    double tot = newTopicMass + topicDocWordMass + currentTree[1];
                    for (int k = 0; k < K; k++) out[k] = currentTree[K + k] / tot;
                    double prev = 0;
                    for (int di = 0; di < nonZeroTopics; di++) {
                        out[localTopicIndex[di]] += (topicDocWordMasses[di] - prev) / tot;
                        prev = topicDocWordMasses[di];
                    }
                    out[K] = newTopicMass / tot;
                }
            }

            if (sample < newTopicMass) {                                      /* WRK:522-526 */
                st->new_mass_cnt++;
                newTopic = first_inactive;
            } else {
                sample -= newTopicMass;
                if (sample < topicDocWordMass) {                              /* WRK:529-531 */
                    st->topic_doc_mass_cnt++;
                    int lb = orc_lower_bound(topicDocWordMasses, sample, nonZeroTopics);
                    if (lb < 0) return 1; /* localTopicIndex[-1] throws -> Q11 */
                    newTopic = localTopicIndex[lb];
                } else {                                                      /* WRK:533-535 */
                    st->word_ftree_mass_cnt++;
                    newTopic = orc_ftree_sample(currentTree, K, nextUniform2);
                    if (newTopic == -2) return 1;
                }
            }
            if (newTopic == -1) newTopic = K - 1;                             /* WRK:549-552 */

            o->z[m][base + position] = newTopic;                              /* WRK:557 */
            localTopicCounts[(size_t)m * K + newTopic]++;                     /* WRK:560 */
            /* WRK:563-584: isNewTopic is evaluated after the increment, hence
             * always false (Q1): the dense list never grows. */
            st->tokens++;
            if (newTopic != oldTopic) {                                       /* WRK:587-589 */
                st->changed++;
                delta_t dl = { oldTopic, newTopic, type, m,
                               (int64_t)(((uint64_t)doc_global << 34) | ((uint64_t)m << 31) | ((uint64_t)position << 11) | (uint64_t)newTopic) };
                dv_push(dv, dl);
            }
        }
    }
    return 0;
}

static void apply_deltas(orc_model* o, const delta_vec* dv, orc_stats* st, int32_t* delta_nwk, int32_t* delta_nk, int apply)
{
    /* UPD:181-272 as one updater draining one FIFO queue, minus the two
     * FTree.update calls (trees are rebuilt from the counts at the next sweep
     * start; DESIGN.md "tree freshness") and minus the doc-topic histogram
     * (recomputed on demand by orc_get_doc_topic_hist). */
    const int K = o->K, M = o->M;
    for (size_t i = 0; i < dv->n; i++) {
        const delta_t* dl = &dv->v[i];
        size_t row = (size_t)(o->rowbase[dl->mod] + dl->type) * K;
        if (dl->oldT != -1) {
            if (apply) { o->nwk[row + dl->oldT]--; o->nk[(size_t)dl->mod * K + dl->oldT]--; }   /* UPD:199-216 */
            if (delta_nwk) delta_nwk[row + dl->oldT]--;
            if (delta_nk) delta_nk[(size_t)dl->mod * K + dl->oldT]--;
        }
        if (apply) { o->nwk[row + dl->newT]++; o->nk[(size_t)dl->mod * K + dl->newT]++; }       /* UPD:207,218 */
        if (delta_nwk) delta_nwk[row + dl->newT]++;
        if (delta_nk) delta_nk[(size_t)dl->mod * K + dl->newT]++;
        /* UPD:263-270: the delta that activates a topic is the FIRST one in (global entity, view, position) order, i.e.
         * the one with the smallest key.  A whole sweep visits the entities in that order, so "first queued" and
         * "smallest key" coincide; a list sweep (orc_sweep_list) visits them in list order, and the key decides. */
        if (o->inactive[dl->newT] && dl->key < st->activation_key) {
            st->activated_topic = dl->newT; st->activated_modality = dl->mod; st->activation_key = dl->key;
        }
    }
    if (apply && st->activated_topic >= 0) {
        o->inactive[st->activated_topic] = 0;
        o->alpha[(size_t)st->activated_modality * (K + 1) + st->activated_topic] = o->alpha[(size_t)st->activated_modality * (K + 1) + K];
    }
    (void)M;
}

static int sweep_impl(orc_model* o, uint32_t sweep_idx, uint64_t seed, int64_t doc_id_base,
                      const double* p_in, uint32_t flags, orc_stats* st,
                      int32_t* delta_nwk, int32_t* delta_nk,
                      double* const* tok_dbg,
                      int n_trace, const int64_t* trace_doc, const int32_t* trace_view,
                      const int32_t* trace_pos, double* trace_out,
                      const int64_t* doc_list, int64_t n_list);

int orc_sweep(orc_model* o, uint32_t sweep_idx, uint64_t seed, int64_t doc_id_base,
              const double* p_in, uint32_t flags, orc_stats* st,
              int32_t* delta_nwk, int32_t* delta_nk,
              double* const* tok_dbg,
              int n_trace, const int64_t* trace_doc, const int32_t* trace_view,
              const int32_t* trace_pos, double* trace_out)
{
    return sweep_impl(o, sweep_idx, seed, doc_id_base, p_in, flags, st, delta_nwk, delta_nk, tok_dbg,
                      n_trace, trace_doc, trace_view, trace_pos, trace_out, NULL, 0);
}

/* The same deferred sweep over a LIST of entities only (local indices, visited in list order): one segment of a
 * segmented sweep (MVHDP_SWEEP_SEGMENT_APPLY: the worker threads' slices PTM:1051-1098 taken one after the other, the
 * updater catching up in between).  Entities outside the list are left untouched. */
int orc_sweep_list(orc_model* o, uint32_t sweep_idx, uint64_t seed, int64_t doc_id_base,
                   const double* p_in, uint32_t flags, orc_stats* st, int32_t* delta_nwk, int32_t* delta_nk,
                   const int64_t* doc_list, int64_t n_list)
{
    return sweep_impl(o, sweep_idx, seed, doc_id_base, p_in, flags, st, delta_nwk, delta_nk, NULL,
                      0, NULL, NULL, NULL, NULL, doc_list ? doc_list : (const int64_t*)"", n_list);
}

static int sweep_impl(orc_model* o, uint32_t sweep_idx, uint64_t seed, int64_t doc_id_base,
                      const double* p_in, uint32_t flags, orc_stats* st,
                      int32_t* delta_nwk, int32_t* delta_nk,
                      double* const* tok_dbg,
                      int n_trace, const int64_t* trace_doc, const int32_t* trace_view,
                      const int32_t* trace_pos, double* trace_out,
                      const int64_t* doc_list, int64_t n_list)
{
    const int K = o->K, M = o->M;
    orc_stats local; memset(&local, 0, sizeof local);
    local.activated_topic = -1; local.activated_modality = -1; local.activation_key = INT64_MAX;

    if (!(flags & (ORC_SWEEP_REUSE_TREES | ORC_SWEEP_FROZEN))) orc_build_trees(o);

    int first_inactive = -1;                      /* inActiveTopicIndex.first() WRK:525 */
    for (int k = 0; k < K; k++) if (o->inactive[k]) { first_inactive = k; break; }

    double* p_own = NULL;
    const double* p = p_in;
    if (!p) {
        p_own = (double*)malloc((size_t)(o->D > 0 ? o->D : 1) * M * M * sizeof(double));
        orc_draw_p_philox(o, seed, sweep_idx, doc_id_base, p_own);
        p = p_own;
    }
    if (delta_nwk) memset(delta_nwk, 0, (size_t)o->rowbase[M] * K * sizeof(int32_t));
    if (delta_nk) memset(delta_nk, 0, (size_t)M * K * sizeof(int32_t));

    int32_t* localTopicCounts = (int32_t*)malloc((size_t)M * K * sizeof(int32_t));
    int32_t* localTopicIndex = (int32_t*)malloc((size_t)(K + 1) * sizeof(int32_t));
    double* topicDocWordMasses = (double*)malloc((size_t)(K + 1) * sizeof(double));
    double* totalMassOtherModalities = (double*)malloc((size_t)K * sizeof(double));
    delta_vec dv = { NULL, 0, 0 };

    const int64_t n_visit = doc_list ? n_list : o->D;
    for (int64_t q = 0; q < n_visit; q++) {
        const int64_t d = doc_list ? doc_list[q] : q;
        if (d < 0 || d >= o->D) continue;
        memset(localTopicIndex, 0, (size_t)(K + 1) * sizeof(int32_t));
        topicDocWordMasses[0] = 0;
        int rc = sample_one_doc(o, d, doc_id_base + d, sweep_idx, seed, p + (size_t)d * M * M,
                                first_inactive, &local, &dv, tok_dbg,
                                n_trace, trace_doc, trace_view, trace_pos, trace_out,
                                localTopicCounts, localTopicIndex, topicDocWordMasses, totalMassOtherModalities);
        if (rc) local.aborted_docs++;
    }

    if (flags & ORC_SWEEP_FROZEN) { dv.n = 0; local.changed = 0; }     /* nut == 0: no FastQDelta is ever queued (WRK:587) */
    apply_deltas(o, &dv, &local, delta_nwk, delta_nk, !(flags & ORC_SWEEP_NO_APPLY));

    free(dv.v); free(localTopicCounts); free(localTopicIndex);
    free(topicDocWordMasses); free(totalMassOtherModalities); free(p_own);
    if (st) *st = local;
    return 0;
}

/* ======================================================================== */
/* The LIVE sweep as a sequential algorithm (MVHDP_SWEEP_LIVE with ONE resident  */
/* wavefront: mvhdp_tuning.single_wave).  UPD:197-218 applied while WRK:425-590 */
/* samples, with the visibility rule of the product's kernels:                  */
/*   n_wk   a token sees the deltas of every earlier 64-token chunk (of any     */
/*          entity) -- a chunk's FastQDelta records land together at its end    */
/*   n_k    of the segment start (the block's private table lands at the        */
/*          kernel's end); so are the coefficients and tree[1] of every word    */
/*   tree branch (WRK:533-535)                                                  */
/*     rows = 0   the stored trees of the segment start (FT:111-136)            */
/*     rows = 1   the word's LIVE row: the arithmetic of row_sample_live        */
/*                (mvhdp_sweep_fast.hip) restated operation for operation --    */
/*                fp32, the DPP scan's order of additions, the packed fused     */
/*                multiply-adds; cell16 = 1 (the 16-bit mirror: eight cells a   */
/*                lane; a HEAVY word -- more than 65534 tokens at the sweep     */
/*                start -- walks its stored tree) or 0 (32-bit rows: four)      */
/* Entities are visited in `order` (the product's longest-first order), segment */
/* s taking positions s, s + nseg, ...; a topic is activated (UPD:263-270) at   */
/* the end of the segment whose delta reached it first.                         */
/* ======================================================================== */

/* wave_incl_scan_f_dpp (mvhdp_wave.h): row_shr 1, 2, 4, 8 inside rows of 16 lanes, then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3 */
static void scan64_f32(float v[64])
{
    float t[64];
    for (int sh = 1; sh <= 8; sh <<= 1) {
        for (int i = 0; i < 64; i++) t[i] = ((i & 15) >= sh) ? v[i - sh] : 0.0f;
        for (int i = 0; i < 64; i++) v[i] = v[i] + t[i];
    }
    for (int i = 0; i < 64; i++) t[i] = v[i];
    for (int i = 16; i < 32; i++) v[i] = t[i] + t[15];
    for (int i = 48; i < 64; i++) v[i] = t[i] + t[47];
    for (int i = 0; i < 64; i++) t[i] = v[i];
    for (int i = 32; i < 64; i++) v[i] = t[i] + t[31];
}

static int smoothing_sample_live(const float* smp, int K, float t)
{
    for (int k = 0; k < K; k++) if (smp[k] > t) return k;
    return K - 1;
}

/* row_batch_pick: cells (b*64 + lane)*CPL + j of the row; coefficients zero from K on */
static int row_batch_pick(const int32_t* row, const float* cf, int K, int cpl, int b, float base, float target, int force, float* tot, int* any)
{
    const int ns = cpl / 2;
    float Ax[64][4], Ay[64][4], nx[64][4], ny[64][4], kx[64][4], ky[64][4], acc[64], incl[64];
    for (int l = 0; l < 64; l++) {
        const int kl = (b * 64 + l) * cpl;
        for (int s2 = 0; s2 < ns; s2++) {
            const int ke = kl + 2 * s2, ko = ke + 1;
            nx[l][s2] = (kl < K && ke < K) ? (float)row[ke] : 0.0f; ny[l][s2] = (kl < K && ko < K) ? (float)row[ko] : 0.0f;
            kx[l][s2] = (kl < K && ke < K) ? cf[ke] : 0.0f;         ky[l][s2] = (kl < K && ko < K) ? cf[ko] : 0.0f;
        }
        Ax[l][0] = nx[l][0] * kx[l][0]; Ay[l][0] = ny[l][0] * ky[l][0];
        for (int s2 = 1; s2 < ns; s2++) { Ax[l][s2] = fmaf(nx[l][s2], kx[l][s2], Ax[l][s2 - 1]); Ay[l][s2] = fmaf(ny[l][s2], ky[l][s2], Ay[l][s2 - 1]); }
        acc[l] = Ax[l][ns - 1] + Ay[l][ns - 1];
        incl[l] = acc[l];
    }
    scan64_f32(incl);
    *tot = incl[63];
    int first_pos = -1, last_pos = -1;
    for (int l = 0; l < 64; l++) if (acc[l] > 0.0f) { if (first_pos < 0) first_pos = l; last_pos = l; }
    *any = last_pos >= 0;
    if (last_pos < 0 || !(force || base + *tot > target)) return -1;
    int hl = last_pos;
    if (!force) for (int l = 0; l < 64; l++) if (acc[l] > 0.0f && base + incl[l] > target) { hl = l; break; }
    const float thr = target - (base + (hl > 0 ? incl[hl - 1] : 0.0f));
    const float ev = Ax[hl][ns - 1];
    int cnt = 0;
    for (int s2 = 0; s2 < ns; s2++) cnt += (Ax[hl][s2] <= thr) ? 1 : 0;
    for (int s2 = 0; s2 < ns; s2++) cnt += (ev + Ay[hl][s2] <= thr) ? 1 : 0;
    int pp = force ? cpl : cnt;
    if (pp >= cpl) {
        int lp = 0;
        for (int s2 = 0; s2 < ns; s2++) if (nx[hl][s2] > 0.0f && kx[hl][s2] > 0.0f) lp = s2;
        for (int s2 = 0; s2 < ns; s2++) if (ny[hl][s2] > 0.0f && ky[hl][s2] > 0.0f) lp = ns + s2;
        pp = lp;
    }
    return b * 64 * cpl + hl * cpl + (pp < ns ? 2 * pp : 2 * (pp - ns) + 1);
}

/* mass0: the first batch's mass at the segment start (MvModel::mass0), used by rows of the 16-bit mirror in exactly two batches: a target at or beyond it
   is first looked for in the second batch with that mass as its base; only when the chosen batch does not hold it are the batches scanned
   in order with their live masses */
int orc_row_sample_live(const int32_t* row, const float* cf, const float* smp, int K, int cell16, float u2f, float rootf, float mass0)
{
    const int cpl = cell16 ? 8 : 4;
    const float S = smp[K - 1];
    float target = u2f * rootf;
    if (target < S) return smoothing_sample_live(smp, K, target);
    target -= S;
    float base = 0.0f;
    int lastb = -1;
    const int nb = (K + 64 * cpl - 1) / (64 * cpl);
    if (cell16 && nb == 2) {                              /* (the 16-bit mirror alone: the kernel flavours of the 32-bit table scan in order) */
        const int bq = target >= mass0;
        float tot; int any;
        const int r = row_batch_pick(row, cf, K, cpl, bq, bq ? mass0 : 0.0f, target, 0, &tot, &any);
        if (r >= 0) return r;
    }
    for (int b = 0; b < nb; b++) {
        float tot; int any;
        const int r = row_batch_pick(row, cf, K, cpl, b, base, target, 0, &tot, &any);
        if (r >= 0) return r;
        if (any) lastb = b;
        base += tot;
    }
    if (lastb >= 0) { float tot; int any; return row_batch_pick(row, cf, K, cpl, lastb, 0.0f, 0.0f, 1, &tot, &any); }
    return smoothing_sample_live(smp, K, u2f * S);
}

typedef struct {
    float* coef;      /* [M][K] */
    float* smp;       /* [M][K] */
    double* root;     /* [sumV] */
    float* mass0;     /* [sumV] the first row batch's mass at the segment start */
    /* births (rows = 1): the topics that are inactive at the segment start, in index order; a new-topic draw (WRK:523-526) goes to
       b_list[b_pos], and a chunk whose deltas reach b_list[r] moves b_pos to r + 1 (UPD:263-270 applied chunk by chunk) */
    int births, b_n, b_pos;
    int32_t* b_list;  /* [K] */
    int32_t* b_rank;  /* [K] position in b_list, -1: active */
    int64_t* b_key;   /* [K] the first delta (entity, view, position) that reached b_list[r] */
    uint8_t* heavy;   /* [sumV] more than 65534 tokens at the sweep start */
    int32_t* nk_seg;  /* [M][K] tokensPerTopic of the segment start */
    int32_t* nk_delta;
} live_state;

/* live_coef_kernel + live_rows_prepare_kernel (mvhdp_kernels.hip) */
static void live_prepare(orc_model* o, live_state* ls, int rows, int cell16, int sweep_start)
{
    const int K = o->K, M = o->M;
    const int64_t nrows = o->rowbase[M];
    memcpy(ls->nk_seg, o->nk, (size_t)M * K * sizeof(int32_t));
    memset(ls->nk_delta, 0, (size_t)M * K * sizeof(int32_t));
    orc_build_trees(o);                                   /* the stored trees of the segment start (all of them: only some are read) */
    if (sweep_start)
        for (int64_t r = 0; r < nrows; r++) {
            long long sum = 0;
            for (int k = 0; k < K; k++) { const int c = o->nwk[(size_t)r * K + k]; sum += c < 0 ? 70000 : c; }
            ls->heavy[r] = sum > 65534;
        }
    if (!rows) { for (int64_t r = 0; r < nrows; r++) ls->root[r] = o->trees[(size_t)r * 2 * K + 1]; return; }
    for (int m = 0; m < M; m++) {
        float* cf = ls->coef + (size_t)m * K; float* smp = ls->smp + (size_t)m * K;
        for (int k = 0; k < K; k++)
            cf[k] = o->inactive[k] ? 0.0f : (float)(o->gamma[m] * o->alpha[(size_t)m * (K + 1) + k] / ((double)o->nk[(size_t)m * K + k] + o->beta_sum[m]));
        const float beta32 = (float)o->beta[m];
        float run = 0.0f;
        for (int k = 0; k < K; k++) { run += cf[k] * beta32; smp[k] = run; }
        for (int64_t r = o->rowbase[m]; r < o->rowbase[m + 1]; r++) {
            if (cell16 && ls->heavy[r]) { ls->root[r] = o->trees[(size_t)r * 2 * K + 1]; continue; }   /* build_trees_kernel(only_heavy) overwrites it */
            double acc[64];
            for (int l = 0; l < 64; l++) { acc[l] = 0.0; for (int k = l; k < K; k += 64) acc[l] += (double)cf[k] * (double)o->nwk[(size_t)r * K + k]; }
            for (int sft = 32; sft >= 1; sft >>= 1) { double t[64]; for (int l = 0; l < 64; l++) t[l] = acc[l] + acc[l ^ sft]; memcpy(acc, t, sizeof t); }
            ls->root[r] = (double)smp[K - 1] + acc[0];
            {   /* the same sum over the topics of the first register batch alone */
                const int b0 = cell16 ? 512 : 256;
                double a0[64];
                for (int l = 0; l < 64; l++) { a0[l] = 0.0; for (int k = l; k < K && k < b0; k += 64) a0[l] += (double)cf[k] * (double)o->nwk[(size_t)r * K + k]; }
                for (int sft = 32; sft >= 1; sft >>= 1) { double t[64]; for (int l = 0; l < 64; l++) t[l] = a0[l] + a0[l ^ sft]; memcpy(a0, t, sizeof t); }
                ls->mass0[r] = (float)a0[0];
            }
        }
    }
}

/* WRK:301-601 for one entity of a live sweep (see the head of this section) */
static int sample_one_doc_live(orc_model* o, live_state* ls, int rows, int cell16, int64_t d, int64_t doc_global, uint32_t sweep, uint64_t seed,
                               const double* p, int first_inactive, orc_stats* st, int64_t* act_key,
                               int32_t* localTopicCounts, int32_t* localTopicIndex, double* topicDocWordMasses, double* totalMassOtherModalities)
{
    const int K = o->K, M = o->M;
    int docLength[ORC_MAX_M];
    memset(localTopicCounts, 0, (size_t)M * K * sizeof(int32_t));
    for (int m = 0; m < M; m++) {
        docLength[m] = 0;
        if (o->doc_off[m]) {
            int64_t b = o->doc_off[m][d], e = o->doc_off[m][d + 1];
            docLength[m] = (int)(e - b);
            for (int64_t i = b; i < e; i++) if (o->z[m][i] != -1) localTopicCounts[(size_t)m * K + o->z[m][i]]++;
        }
    }
    int denseIndex = 0;
    for (int topic = 0; topic < K; topic++) {
        int i = 0, found = 0;
        while (i < M && !found) { if (localTopicCounts[(size_t)i * K + topic] != 0) { localTopicIndex[denseIndex++] = topic; found = 1; } i++; }
    }
    int nonZeroTopics = denseIndex;
    for (int m = 0; m < M; m++) {
        for (int k = 0; k < K; k++) totalMassOtherModalities[k] = 0;
        for (denseIndex = 0; denseIndex < nonZeroTopics; denseIndex++) {
            int topic = localTopicIndex[denseIndex];
            for (int i = 0; i < M; i++)
                if (i != m && docLength[i] != 0)
                    totalMassOtherModalities[topic] += p[m * M + i]
                        * (localTopicCounts[(size_t)i * K + topic] + o->gamma[i] * o->alpha[(size_t)i * (K + 1) + topic])
                        / (docLength[i] + (double)o->gamma[i] * o->alpha_sum[i]);
            totalMassOtherModalities[topic] = totalMassOtherModalities[topic] * (docLength[m] + (double)o->gamma[m] * o->alpha_sum[m]);
        }
        double newAll = 0;
        for (int i = 0; i < M; i++) newAll += p[m * M + i] * (o->gamma[i] * o->alpha[(size_t)i * (K + 1) + K]) / (docLength[i] + (double)o->gamma[i] * o->alpha_sum[i]);
        newAll = newAll * (docLength[m] + (double)o->gamma[m] * o->alpha_sum[m]);
        if (docLength[m] == 0) continue;
        const int64_t base = o->doc_off[m][d];
        /* the FastQDelta records of the current 64-token chunk: (type, old, new), applied together at its end */
        int32_t ch_type[64], ch_old[64], ch_new[64]; int64_t ch_key[64]; int n_ch = 0;
        for (int position = 0; position < docLength[m]; position++) {
            const int type = o->tokens[m][base + position];
            if (type >= o->V[m]) { st->oov_skipped++; goto chunk_end; }
            {
                const int oldTopic = o->z[m][base + position];
                const size_t r = (size_t)(o->rowbase[m] + type);
                const int32_t* cnt = o->nwk + r * K;
                if (oldTopic != -1) {
                    localTopicCounts[(size_t)m * K + oldTopic]--;
                    int del = localTopicCounts[(size_t)m * K + oldTopic] == 0;
                    int jj = 0;
                    while (del && jj < M) { del = localTopicCounts[(size_t)jj * K + oldTopic] == 0; jj++; }
                    if (del) {
                        denseIndex = 0;
                        while (localTopicIndex[denseIndex] != oldTopic) { denseIndex++; if (denseIndex >= K) return 1; }
                        while (denseIndex < nonZeroTopics) { if (denseIndex < K - 1) localTopicIndex[denseIndex] = localTopicIndex[denseIndex + 1]; denseIndex++; }
                        nonZeroTopics--;
                    }
                }
                double mass = 0.0;
                for (denseIndex = 0; denseIndex < nonZeroTopics; denseIndex++) {
                    int topic = localTopicIndex[denseIndex];
                    int n = localTopicCounts[(size_t)m * K + topic];
                    double p_wt = (cnt[topic] + o->beta[m]) / (ls->nk_seg[(size_t)m * K + topic] + o->beta_sum[m]);
                    mass += (p[m * M + m] * n + totalMassOtherModalities[topic]) * p_wt;
                    topicDocWordMasses[denseIndex] = mass;
                }
                const double newTopicMass = (first_inactive < 0) ? 0 : newAll / K;
                double u1, u2;
                orc_token_uniforms(seed, sweep, doc_global, m, (uint32_t)position, &u1, &u2);
                const double root = ls->root[r];
                double sample = u1 * (newTopicMass + mass + root);
                int newTopic = -1;
                if (sample < newTopicMass) { st->new_mass_cnt++; newTopic = !ls->births ? first_inactive : (ls->b_pos < ls->b_n ? ls->b_list[ls->b_pos] : -1); }
                else {
                    sample -= newTopicMass;
                    if (sample < mass) {
                        st->topic_doc_mass_cnt++;
                        int lb = orc_lower_bound(topicDocWordMasses, sample, nonZeroTopics);
                        if (lb < 0) return 1;
                        newTopic = localTopicIndex[lb];
                    } else {
                        st->word_ftree_mass_cnt++;
                        if (!rows || (cell16 && ls->heavy[r])) newTopic = orc_ftree_sample(o->trees + r * 2 * K, K, u2);
                        else newTopic = orc_row_sample_live(cnt, ls->coef + (size_t)m * K, ls->smp + (size_t)m * K, K, cell16, (float)u2, (float)root, ls->mass0[r]);
                        if (newTopic == -2) return 1;
                    }
                }
                if (newTopic == -1) newTopic = K - 1;
                o->z[m][base + position] = newTopic;
                localTopicCounts[(size_t)m * K + newTopic]++;
                st->tokens++;
                if (newTopic != oldTopic) {
                    st->changed++;
                    ch_type[n_ch] = type; ch_old[n_ch] = oldTopic; ch_new[n_ch] = newTopic;
                    ch_key[n_ch] = (int64_t)(((uint64_t)doc_global << 34) | ((uint64_t)m << 31) | ((uint64_t)position << 11) | (uint64_t)newTopic);
                    n_ch++;
                }
            }
        chunk_end:
            if ((position & 63) == 63 || position == docLength[m] - 1) {
                for (int i = 0; i < n_ch; i++) {
                    const size_t rr = (size_t)(o->rowbase[m] + ch_type[i]) * K;
                    if (ch_old[i] != -1) { o->nwk[rr + ch_old[i]]--; ls->nk_delta[(size_t)m * K + ch_old[i]]--; }
                    o->nwk[rr + ch_new[i]]++; ls->nk_delta[(size_t)m * K + ch_new[i]]++;
                    if (o->inactive[ch_new[i]] && ch_key[i] < *act_key) *act_key = ch_key[i];
                    if (ls->births && o->inactive[ch_new[i]]) {
                        const int r = ls->b_rank[ch_new[i]];
                        if (r >= 0) { if (ch_key[i] < ls->b_key[r]) ls->b_key[r] = ch_key[i]; if (r + 1 > ls->b_pos) ls->b_pos = r + 1; }
                    }
                }
                n_ch = 0;
            }
        }
    }
    return 0;
}

int orc_sweep_live_seq(orc_model* o, uint32_t sweep_idx, uint64_t seed, int64_t doc_id_base, const double* p_in,
                       const int64_t* order, int64_t n_order, int nseg, int rows, int cell16, orc_stats* st)
{
    const int K = o->K, M = o->M;
    const int64_t nrows = o->rowbase[M];
    orc_stats local; memset(&local, 0, sizeof local);
    local.activated_topic = -1; local.activated_modality = -1; local.activation_key = INT64_MAX;
    double* p_own = NULL;
    const double* p = p_in;
    if (!p) { p_own = (double*)malloc((size_t)(o->D > 0 ? o->D : 1) * M * M * sizeof(double)); orc_draw_p_philox(o, seed, sweep_idx, doc_id_base, p_own); p = p_own; }
    live_state ls;
    ls.coef = (float*)calloc((size_t)M * K, sizeof(float)); ls.smp = (float*)calloc((size_t)M * K, sizeof(float));
    ls.root = (double*)calloc((size_t)(nrows > 0 ? nrows : 1), sizeof(double)); ls.mass0 = (float*)calloc((size_t)(nrows > 0 ? nrows : 1), sizeof(float)); ls.heavy = (uint8_t*)calloc((size_t)(nrows > 0 ? nrows : 1), 1);
    ls.nk_seg = (int32_t*)calloc((size_t)M * K, sizeof(int32_t)); ls.nk_delta = (int32_t*)calloc((size_t)M * K, sizeof(int32_t));
    ls.births = rows != 0;                                   /* (the live-rows form of the library; its stored-tree form activates one topic per segment) */
    ls.b_list = (int32_t*)calloc((size_t)K, sizeof(int32_t)); ls.b_rank = (int32_t*)calloc((size_t)K, sizeof(int32_t)); ls.b_key = (int64_t*)calloc((size_t)K, sizeof(int64_t));
    int32_t* localTopicCounts = (int32_t*)malloc((size_t)M * K * sizeof(int32_t));
    int32_t* localTopicIndex = (int32_t*)malloc((size_t)(K + 1) * sizeof(int32_t));
    double* topicDocWordMasses = (double*)malloc((size_t)(K + 1) * sizeof(double));
    double* totalMassOtherModalities = (double*)malloc((size_t)K * sizeof(double));
    if (nseg < 1) nseg = 1;
    for (int seg = 0; seg < nseg; seg++) {
        int first_inactive = -1;
        for (int k = 0; k < K; k++) if (o->inactive[k]) { first_inactive = k; break; }
        live_prepare(o, &ls, rows, cell16, seg == 0);
        int64_t act_key = INT64_MAX;
        ls.b_n = 0; ls.b_pos = 0;
        for (int k = 0; k < K; k++) { ls.b_rank[k] = -1; if (o->inactive[k]) { ls.b_rank[k] = ls.b_n; ls.b_key[ls.b_n] = INT64_MAX; ls.b_list[ls.b_n++] = k; } }
        for (int64_t q = seg; q < n_order; q += nseg) {
            const int64_t d = order[q];
            if (d < 0 || d >= o->D) continue;
            memset(localTopicIndex, 0, (size_t)(K + 1) * sizeof(int32_t));
            topicDocWordMasses[0] = 0;
            if (sample_one_doc_live(o, &ls, rows, cell16, d, doc_id_base + d, sweep_idx, seed, p + (size_t)d * M * M, first_inactive, &local, &act_key,
                                    localTopicCounts, localTopicIndex, topicDocWordMasses, totalMassOtherModalities)) local.aborted_docs++;
        }
        for (size_t i = 0; i < (size_t)M * K; i++) o->nk[i] += ls.nk_delta[i];          /* the block's private tokensPerTopic table lands */
        if (ls.births) {                                                               /* every topic a chunk's deltas reached, in index order */
            for (int r = 0; r < ls.b_pos; r++) {
                const int t = ls.b_list[r], mv = (int)((ls.b_key[r] >> 31) & 7);
                if (o->inactive[t]) { o->inactive[t] = 0; o->alpha[(size_t)mv * (K + 1) + t] = o->alpha[(size_t)mv * (K + 1) + K]; }
                if (local.activated_topic < 0) { local.activated_topic = t; local.activated_modality = mv; local.activation_key = ls.b_key[r]; }
            }
        } else
        if (act_key != INT64_MAX) {                                                    /* UPD:263-270 at the segment's end */
            const int t = (int)(act_key & 0x7ff), mv = (int)((act_key >> 31) & 7);
            if (o->inactive[t]) { o->inactive[t] = 0; o->alpha[(size_t)mv * (K + 1) + t] = o->alpha[(size_t)mv * (K + 1) + K]; }
            if (local.activated_topic < 0) { local.activated_topic = t; local.activated_modality = mv; local.activation_key = act_key; }
        }
    }
    free(ls.coef); free(ls.smp); free(ls.root); free(ls.mass0); free(ls.heavy); free(ls.nk_seg); free(ls.nk_delta); free(ls.b_list); free(ls.b_rank); free(ls.b_key);
    free(localTopicCounts); free(localTopicIndex); free(topicDocWordMasses); free(totalMassOtherModalities); free(p_own);
    if (st) *st = local;
    return 0;
}

void orc_apply_delta(orc_model* o, const int32_t* delta_nwk, const int32_t* delta_nk,
                     int32_t act_topic, int32_t act_modality)
{
    const int K = o->K, M = o->M;
    size_t n = (size_t)o->rowbase[M] * K;
    for (size_t i = 0; i < n; i++) o->nwk[i] += delta_nwk[i];
    for (size_t i = 0; i < (size_t)M * K; i++) o->nk[i] += delta_nk[i];
    if (act_topic >= 0 && act_topic < K && act_modality >= 0 && act_modality < M && o->inactive[act_topic]) {
        o->inactive[act_topic] = 0;                                                  /* UPD:266-268 */
        o->alpha[(size_t)act_modality * (K + 1) + act_topic] = o->alpha[(size_t)act_modality * (K + 1) + K];
    }
}

/* ======================================================================== */
/* SURVEY §8f "next" rows: hyper-parameter statistics and the log likelihood */
/* ======================================================================== */

/* MALLET 2.0.8 Dirichlet.logGammaStirling (class file only; arithmetic read from the jar's bytecode
 * with tools/javap_lite.py): shift z up to >= 2, Stirling series, subtract the logs back. */
double orc_log_gamma_stirling(double z)
{
    static const double HALF_LOG_TWO_PI_UNUSED = 0; (void)HALF_LOG_TWO_PI_UNUSED;
    const double HALF_LOG_TWO_PI = log(6.283185307179586) / 2.0;   /* Dirichlet.<clinit> */
    int shift = 0;
    while (z < 2.0) { z = z + 1; shift++; }
    double result = HALF_LOG_TWO_PI + (z - 0.5) * log(z) - z + 1 / (12.0 * z) - 1 / (360.0 * z * z * z)
                    + 1 / (1260.0 * z * z * z * z * z);
    while (shift > 0) { shift--; z = z - 1; result = result - log(z); }
    return result;
}

/* MALLET 2.0.8 Dirichlet.digamma as compiled: DIGAMMA_COEF_1..7 are written as integer quotients
 * (1/12, 1/120, ...) in that release and are therefore all 0 in the class file, so the series
 * reduces to log(z) - 0.5/z (the nested products are kept: they only produce signed zeros). */
double orc_mallet_digamma(double z)
{
    double psi = 0;
    if (z < 1e-06) { psi = -0.5772156649015329 - 1 / z; return psi; }
    while (z < 9.5) { psi = psi - 1 / z; z = z + 1; }
    double invZ = 1 / z;
    double invZSquared = invZ * invZ;
    psi = psi + (log(z) - 0.5 * invZ
          - invZSquared * (0.0 - invZSquared * (0.0 - invZSquared * (0.0 - invZSquared * (0.0 - invZSquared * (0.0 - invZSquared * (0.0 - invZSquared * 0.0)))))));
    return psi;
}

/* MALLET 2.0.8 Dirichlet.learnSymmetricConcentration(countHistogram, observationLengths, numDimensions,
 * currentValue), from the bytecode.  Kept quirks: previousLength is never advanced, and for lengths
 * <= 20 currentDigamma keeps accumulating across lengths. */
double orc_learn_symmetric_concentration(const int32_t* countHistogram, int n_count, const int32_t* observationLengths, int n_len,
                                         int numDimensions, double currentValue)
{
    double currentDigamma;
    int largestNonZeroCount = 0;
    int* nonZeroLengthIndex = (int*)malloc((size_t)(n_len > 0 ? n_len : 1) * sizeof(int));
    for (int index = 0; index < n_count; index++) if (countHistogram[index] > 0) largestNonZeroCount = index;
    int denseIndex = 0;
    for (int index = 0; index < n_len; index++)
        if (observationLengths[index] > 0) { nonZeroLengthIndex[denseIndex] = index; denseIndex++; }
    int denseIndexSize = denseIndex;
    for (int iteration = 1; iteration <= 200; iteration++) {
        double currentParameter = currentValue / numDimensions;
        currentDigamma = 0;
        double numerator = 0;
        for (int index = 1; index <= largestNonZeroCount; index++) {
            currentDigamma += 1.0 / (currentParameter + index - 1);
            numerator += countHistogram[index] * currentDigamma;
        }
        currentDigamma = 0;
        double denominator = 0;
        int previousLength = 0;
        double cachedDigamma = orc_mallet_digamma(currentValue);
        for (denseIndex = 0; denseIndex < denseIndexSize; denseIndex++) {
            int length = nonZeroLengthIndex[denseIndex];
            if (length - previousLength > 20) {
                currentDigamma = orc_mallet_digamma(currentValue + length) - cachedDigamma;
            } else {
                for (int index = previousLength; index < length; index++) currentDigamma += 1.0 / (currentValue + index);
            }
            denominator += currentDigamma * observationLengths[length];
        }
        currentValue = currentParameter * numerator / denominator;
    }
    free(nonZeroLengthIndex);
    return currentValue;
}

/* countHistogram of optimizeBeta (PTM:2295-2309): number of (type, topic) pairs holding each count > 0 */
void orc_count_histogram(const orc_model* o, int m, int32_t* hist, int32_t len)
{
    const int K = o->K;
    memset(hist, 0, (size_t)len * sizeof(int32_t));
    for (int type = 0; type < o->V[m]; type++) {
        const int32_t* counts = o->nwk + (size_t)(o->rowbase[m] + type) * K;
        for (int topic = 0; topic < K; topic++) {
            int count = counts[topic];
            if (count > 0 && count < len) hist[count]++;
        }
    }
}

/* optimizeBeta for one view (PTM:2293-2366) given maxTypeCount; returns the new beta, writes betaSum.
 * The catch (RuntimeException) arm cannot trigger in C; NaN handling is the reference's. */
double orc_optimize_beta(orc_model* o, int m, int maxTypeCount, double* betaSum_out)
{
    const int K = o->K;
    double prevBetaSum = o->beta_sum[m];
    int32_t* countHistogram = (int32_t*)calloc((size_t)maxTypeCount + 1, sizeof(int32_t));
    orc_count_histogram(o, m, countHistogram, maxTypeCount + 1);
    int maxTopicSize = 0;
    for (int topic = 0; topic < K; topic++) if (o->nk[(size_t)m * K + topic] > maxTopicSize) maxTopicSize = o->nk[(size_t)m * K + topic];
    int32_t* topicSizeHistogram = (int32_t*)calloc((size_t)maxTopicSize + 1, sizeof(int32_t));
    for (int topic = 0; topic < K; topic++) topicSizeHistogram[o->nk[(size_t)m * K + topic]]++;
    double betaSum = orc_learn_symmetric_concentration(countHistogram, maxTypeCount + 1, topicSizeHistogram, maxTopicSize + 1,
                                                       o->V[m], o->beta_sum[m]);
    double beta = o->beta[m];
    if (betaSum < o->V[m] * 0.0001) {                       /* PTM:2332-2335 */
        beta = 0.0001; betaSum = beta * o->V[m];
    } else if (isnan(betaSum)) {                            /* PTM:2337-2349 */
        if (o->beta[m] == 0.01) { beta = 0.0001; betaSum = beta * o->V[m]; }
        else { betaSum = prevBetaSum; beta = betaSum / o->V[m]; }
    } else {
        beta = betaSum / o->V[m];                           /* PTM:2351 */
    }
    free(countHistogram); free(topicSizeHistogram);
    *betaSum_out = betaSum;
    return beta;
}

/* optimizeP statistics (PTM:2706-2782): for every ordered pair the sum over entities of
 * pDistr_Mean[m][i][doc], accumulated in entity order exactly as PTM:2789-2792 does afterwards.
 * sums: [M][M].  The TreeMap keyed by view length (PTM:2717,2741) drops views of equal length. */
void orc_optimize_p_sums(const orc_model* o, double* sums)
{
    const int K = o->K, M = o->M;
    double* pd = (double*)calloc((size_t)M * M, sizeof(double));      /* pDistr_Mean[.][.][doc] of the current entity */
    int32_t* localTopicCounts = (int32_t*)malloc((size_t)M * K * sizeof(int32_t));
    for (int i = 0; i < M * M; i++) sums[i] = 0;
    for (int64_t doc = 0; doc < o->D; doc++) {
        int docLength[ORC_MAX_M];
        memset(localTopicCounts, 0, (size_t)M * K * sizeof(int32_t));
        memset(pd, 0, (size_t)M * M * sizeof(double));
        /* TreeMap<Integer,Byte>: key = length, later views overwrite equal keys */
        int keys[ORC_MAX_M], vals[ORC_MAX_M], nkeys = 0;
        for (int m = 0; m < M; m++) {
            int64_t b = o->doc_off[m][doc], e = o->doc_off[m][doc + 1];
            docLength[m] = (int)(e - b);
            for (int64_t t = b; t < e; t++) if (o->z[m][t] != -1) localTopicCounts[(size_t)m * K + o->z[m][t]]++;
            int found = -1;
            for (int q = 0; q < nkeys; q++) if (keys[q] == docLength[m]) found = q;
            if (found >= 0) vals[found] = m; else { keys[nkeys] = docLength[m]; vals[nkeys] = m; nkeys++; }
        }
        /* descending by key */
        for (int a = 0; a < nkeys; a++) for (int b2 = a + 1; b2 < nkeys; b2++)
            if (keys[b2] > keys[a]) { int t = keys[a]; keys[a] = keys[b2]; keys[b2] = t; t = vals[a]; vals[a] = vals[b2]; vals[b2] = t; }
        int previousViews[ORC_MAX_M], nprev = 0;
        previousViews[nprev++] = vals[0];                                 /* PTM:2747 */
        for (int q = 1; q < nkeys; q++) {                                 /* PTM:2751-2780 */
            int m = vals[q];
            if (docLength[m] > 0) {                                       /* Assignments[m] != null */
                int64_t b = o->doc_off[m][doc];
                for (int position = 0; position < docLength[m]; position++) {
                    int zt = o->z[m][b + position];
                    if (zt == -1) continue;
                    for (int pi = 0; pi < nprev; pi++) {
                        int i = previousViews[pi];
                        pd[m * M + i] += (localTopicCounts[(size_t)i * K + zt] > 0 ? 1.0 : 0.0) / (double)docLength[m];
                        pd[i * M + m] = pd[m * M + i];
                    }
                }
            }
            previousViews[nprev++] = m;
        }
        for (int i = 0; i < M * M; i++) sums[i] += pd[i];                 /* PTM:2790-2792, entity order */
    }
    free(pd); free(localTopicCounts);
}

/* modelLogLikelihood (PTM:3322-3452) per view.  docTopics.length is the LabelSequence backing array,
 * max(len, 2) (MALLET FeatureSequence allocates at least 2): entities of length 0/1 with the view
 * present count phantom tokens of topic 0.  A view is "present" when its span is
 * non-empty (CSR cannot express present-but-empty), so only the length-1 phantom survives here. */
void orc_model_log_likelihood(const orc_model* o, double* logLikelihood)
{
    const int K = o->K, M = o->M;
    int32_t* topicCounts = (int32_t*)calloc((size_t)K, sizeof(int32_t));
    double* topicLogGammas = (double*)malloc((size_t)K * sizeof(double));
    for (int m = 0; m < M; m++) {
        double ll = 0;
        const double* al = o->alpha + (size_t)m * (K + 1);
        for (int topic = 0; topic < K; topic++) topicLogGammas[topic] = orc_log_gamma_stirling(o->gamma[m] * al[topic]);
        int modalityCnt = 0;
        for (int64_t doc = 0; doc < o->D; doc++) {
            int64_t b = o->doc_off[m][doc], e = o->doc_off[m][doc + 1];
            if (e == b) continue;                                          /* Assignments[m] == null */
            int backing = (int)(e - b) > 2 ? (int)(e - b) : 2;
            for (int64_t t = b; t < e; t++) topicCounts[o->z[m][t] < 0 ? 0 : o->z[m][t]]++;
            for (int t = (int)(e - b); t < backing; t++) topicCounts[0]++;  /* phantom zeros of the backing array */
            for (int topic = 0; topic < K; topic++)
                if (topicCounts[topic] > 0)
                    ll += (orc_log_gamma_stirling(o->gamma[m] * al[topic] + topicCounts[topic]) - topicLogGammas[topic]);
            ll -= orc_log_gamma_stirling((double)o->gamma[m] * o->alpha_sum[m] + backing);
            modalityCnt++;
            memset(topicCounts, 0, (size_t)K * sizeof(int32_t));
        }
        ll += modalityCnt * orc_log_gamma_stirling((double)o->gamma[m] * o->alpha_sum[m]);
        if (isnan(ll) || isinf(ll)) { logLikelihood[m] = 0; continue; }
        int nonZeroTypeTopics = 0;
        int broke = 0;
        for (int type = 0; type < o->V[m] && !broke; type++) {
            const int32_t* cnt = o->nwk + (size_t)(o->rowbase[m] + type) * K;
            for (int index = 0; index < K; index++) {
                int count = cnt[index];
                if (count > 0) {
                    nonZeroTypeTopics++;
                    ll += (o->beta[m] + count) == 0 ? 0 : orc_log_gamma_stirling(o->beta[m] + count);
                    if (isnan(ll) || isinf(ll)) { ll = 0; break; }         /* PTM:3402-3410 breaks the inner loop only */
                }
            }
        }
        for (int topic = 0; topic < K; topic++) {
            int nk = o->nk[(size_t)m * K + topic];
            ll -= (o->beta[m] * o->V[m] + nk) == 0 ? 0 : orc_log_gamma_stirling((o->beta[m] * o->V[m]) + nk);
            if (isnan(ll) || isinf(ll)) ll = 0;
        }
        ll += (o->beta[m] * o->V[m]) == 0 ? 0 : orc_log_gamma_stirling(o->beta[m] * o->V[m]) * K;
        ll -= o->beta[m] == 0 ? 0 : orc_log_gamma_stirling(o->beta[m]) * nonZeroTypeTopics;
        if (isinf(ll)) ll = 0;
        logLikelihood[m] = ll;
    }
    free(topicCounts); free(topicLogGammas);
}
