/*
 * ref_threaded.c — CPU BASELINE (test/bench infrastructure, NOT product code).
 *
 * The reference's own execution topology restated in C11 + pthreads, to be
 * timed on host cores beside the GPU sweep (bench.py cpu_baseline, kind
 * "port"):
 *   T threads -> nst = 3T/4 samplers over contiguous doc slices and
 *   nut = T/4 updaters over type % nut stripes              PTM:1036-1101
 *   nst*nut unbounded queues of FastQDelta                  PTM:1042-1049, QD
 *   workers read n_wk / n_k / trees live (racy by design)   PTM:84-87, WRK:301-601
 *   updaters apply counts, histograms and two FTree.update
 *   per delta, then sleep 20 ms per polling pass            UPD:181-282
 *   one barrier per iteration                               PTM:1232, WRK:221, UPD:286
 * Nondeterministic like the reference (the reference's per-token uniforms come
 * from the unseedable ThreadLocalRandom, WRK:517,534; here a per-thread
 * splitmix64).  Slightly favourable to the reference: no GC, no object
 * allocation per delta, spinlocks instead of monitors.
 */
#define _GNU_SOURCE
#include "mvhdp_oracle_internal.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

/* ---- SPSC unbounded chunked queue (one worker -> one updater) ---- */
#define QCHUNK 4096
typedef struct { int32_t oldT, newT, type, mod, docOld, docNew; } qdelta; /* QD:12-36 */
typedef struct qchunk { qdelta v[QCHUNK]; _Atomic(struct qchunk*) next; } qchunk;
typedef struct {
    qchunk* head; size_t hpos;            /* consumer */
    qchunk* tail; size_t tpos;            /* producer */
    _Atomic size_t published;             /* total items published */
    size_t consumed;
    char pad[64];
} spsc;

static void q_init(spsc* q) { q->head = q->tail = (qchunk*)calloc(1, sizeof(qchunk)); q->hpos = q->tpos = 0; atomic_store(&q->published, 0); q->consumed = 0; }
static void q_put(spsc* q, qdelta d)
{
    if (q->tpos == QCHUNK) {
        qchunk* c = (qchunk*)calloc(1, sizeof(qchunk));
        atomic_store_explicit(&q->tail->next, c, memory_order_release);
        q->tail = c; q->tpos = 0;
    }
    q->tail->v[q->tpos++] = d;
    atomic_fetch_add_explicit(&q->published, 1, memory_order_release);
}
static int q_poll(spsc* q, qdelta* out)
{
    if (q->consumed == atomic_load_explicit(&q->published, memory_order_acquire)) return 0;
    if (q->hpos == QCHUNK) {
        qchunk* n = atomic_load_explicit(&q->head->next, memory_order_acquire);
        free(q->head); q->head = n; q->hpos = 0;
    }
    *out = q->head->v[q->hpos++];
    q->consumed++;
    return 1;
}
static void q_free(spsc* q) { while (q->head) { qchunk* n = atomic_load(&q->head->next); free(q->head); q->head = n; } }

typedef struct {
    orc_model* o;
    int nst, nut;
    spsc* queues;                 /* [nst*nut], index nst*utId + stId  (WRK:589, UPD:187) */
    atomic_flag* tree_lock;       /* one per (view,type): FTree methods are synchronized (FT:85-149) */
    _Atomic int32_t* hist;        /* topicDocCounts[m][k][c] */
    int64_t hist_off[ORC_MAX_M]; int32_t hist_len[ORC_MAX_M];
    pthread_barrier_t barrier;
    _Atomic int64_t newMassCnt, topicDocMassCnt, wordFTreeMassCnt, tokens, changed; /* WRK:33-35 */
    _Atomic int inactive_lock;
} shared_t;

typedef struct {
    shared_t* sh; int id; int64_t startDoc, numDocs;
    orc_jrand random;             /* the worker's own MALLET Randoms (PTM:1067-1072) */
    uint64_t tlr;                 /* stands in for ThreadLocalRandom */
} worker_t;

typedef struct { shared_t* sh; int id; } updater_t;

static inline double tlr_next_double(uint64_t* s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (double)(z >> 11) * 0x1.0p-53;
}

static inline void tlock(atomic_flag* f) { while (atomic_flag_test_and_set_explicit(f, memory_order_acquire)) { } }
static inline void tunlock(atomic_flag* f) { atomic_flag_clear_explicit(f, memory_order_release); }

static int first_inactive(const orc_model* o)
{
    for (int k = 0; k < o->K; k++) if (o->inactive[k]) return k;
    return -1;
}

/* WRK:301-601, live model */
static void worker_doc(worker_t* w, int64_t d, int32_t* localTopicCounts, int32_t* localTopicIndex,
                       double* topicDocWordMasses, double* totalMassOtherModalities)
{
    shared_t* sh = w->sh; orc_model* o = sh->o;
    const int K = o->K, M = o->M;
    int docLength[ORC_MAX_M];
    double p[ORC_MAX_M * ORC_MAX_M];
    memset(localTopicCounts, 0, (size_t)M * K * sizeof(int32_t));   /* new int[M][K] WRK:320 */
    memset(localTopicIndex, 0, (size_t)K * sizeof(int32_t));        /* WRK:313 */
    memset(topicDocWordMasses, 0, (size_t)K * sizeof(double));      /* WRK:314 */

    for (int m = 0; m < M; m++) {
        for (int j = m; j < M; j++) {                                /* WRK:329-337 */
            double pRand = m == j ? 1.0 : o->p_a[m][j] == 0 ? 0
                : ((double)orc_java_round(1000 * orc_mallet_next_beta(&w->random, o->p_a[m][j], o->p_b[m][j])) / (double)1000);
            p[m * M + j] = (j != 0 && o->beta[j] == 0.0001) ? 0 : pRand;
            p[j * M + m] = (m != 0 && o->beta[m] == 0.0001) ? 0 : pRand;
        }
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
        while (i < M && !found) {
            if (localTopicCounts[(size_t)i * K + topic] != 0) { localTopicIndex[denseIndex++] = topic; found = 1; }
            i++;
        }
    }
    int nonZeroTopics = denseIndex;
    int64_t n_tok = 0, n_chg = 0, c_new = 0, c_doc = 0, c_tree = 0;

    for (int m = 0; m < M; m++) {
        for (int k = 0; k < K; k++) totalMassOtherModalities[k] = 0;
        for (denseIndex = 0; denseIndex < nonZeroTopics; denseIndex++) {
            int topic = localTopicIndex[denseIndex];
            for (int i = 0; i < M; i++)
                if (i != m && docLength[i] != 0)
                    totalMassOtherModalities[topic] += p[m * M + i]
                        * (localTopicCounts[(size_t)i * K + topic] + o->gamma[i] * o->alpha[(size_t)i * (K + 1) + topic])
                        / (docLength[i] + (double)o->gamma[i] * o->alpha_sum[i]);
            totalMassOtherModalities[topic] *= (docLength[m] + (double)o->gamma[m] * o->alpha_sum[m]);
        }
        double newAll = 0;
        for (int i = 0; i < M; i++)
            newAll += p[m * M + i] * (o->gamma[i] * o->alpha[(size_t)i * (K + 1) + K]) / (docLength[i] + (double)o->gamma[i] * o->alpha_sum[i]);
        newAll *= (docLength[m] + (double)o->gamma[m] * o->alpha_sum[m]);
        if (docLength[m] == 0) continue;
        const int64_t base = o->doc_off[m][d];
        _Atomic int32_t* nk = (_Atomic int32_t*)(o->nk + (size_t)m * K);
        for (int position = 0; position < docLength[m]; position++) {
            int type = o->tokens[m][base + position];
            if (type >= o->V[m]) continue;
            int oldTopic = o->z[m][base + position];
            const volatile int32_t* cnt = o->nwk + (size_t)(o->rowbase[m] + type) * K;
            size_t trow = (size_t)(o->rowbase[m] + type);
            double* tree = o->trees + trow * 2 * K;
            if (oldTopic != -1) {
                localTopicCounts[(size_t)m * K + oldTopic]--;
                int del = localTopicCounts[(size_t)m * K + oldTopic] == 0;
                int jj = 0;
                while (del && jj < M) { del = localTopicCounts[(size_t)jj * K + oldTopic] == 0; jj++; }
                if (del) {
                    denseIndex = 0;
                    while (denseIndex < K && localTopicIndex[denseIndex] != oldTopic) denseIndex++;
                    if (denseIndex >= K) return;
                    while (denseIndex < nonZeroTopics) {
                        if (denseIndex < K - 1) localTopicIndex[denseIndex] = localTopicIndex[denseIndex + 1];
                        denseIndex++;
                    }
                    nonZeroTopics--;
                }
            }
            double mass = 0.0;
            for (denseIndex = 0; denseIndex < nonZeroTopics; denseIndex++) {
                int topic = localTopicIndex[denseIndex];
                int n = localTopicCounts[(size_t)m * K + topic];
                double p_wt = (cnt[topic] + o->beta[m]) / (atomic_load_explicit(&nk[topic], memory_order_relaxed) + o->beta_sum[m]);
                mass += (p[m * M + m] * n + totalMassOtherModalities[topic]) * p_wt;
                topicDocWordMasses[denseIndex] = mass;
            }
            int fi = first_inactive(o);
            double newTopicMass = fi < 0 ? 0 : newAll / K;
            double u1 = tlr_next_double(&w->tlr);
            double root = *(volatile double*)&tree[1];
            double sample = u1 * (newTopicMass + mass + root);
            int newTopic = -1;
            if (sample < newTopicMass) { c_new++; newTopic = fi; }
            else {
                sample -= newTopicMass;
                if (sample < mass) {
                    c_doc++;
                    int lb = orc_lower_bound(topicDocWordMasses, sample, nonZeroTopics);
                    if (lb < 0) return;
                    newTopic = localTopicIndex[lb];
                } else {
                    c_tree++;
                    double u2 = tlr_next_double(&w->tlr);
                    tlock(&sh->tree_lock[trow]);
                    newTopic = orc_ftree_sample(tree, K, u2);
                    tunlock(&sh->tree_lock[trow]);
                }
            }
            if (newTopic < 0) newTopic = K - 1;
            o->z[m][base + position] = newTopic;
            localTopicCounts[(size_t)m * K + newTopic]++;
            n_tok++;
            if (newTopic != oldTopic && sh->nut > 0) {
                n_chg++;
                qdelta dl = { oldTopic, newTopic, type, m,
                              oldTopic == -1 ? 0 : localTopicCounts[(size_t)m * K + oldTopic],
                              localTopicCounts[(size_t)m * K + newTopic] };
                q_put(&sh->queues[sh->nst * (type % sh->nut) + w->id], dl);   /* WRK:589 */
            }
        }
    }
    atomic_fetch_add(&sh->tokens, n_tok); atomic_fetch_add(&sh->changed, n_chg);
    atomic_fetch_add(&sh->newMassCnt, c_new); atomic_fetch_add(&sh->topicDocMassCnt, c_doc);
    atomic_fetch_add(&sh->wordFTreeMassCnt, c_tree);
}

static void* worker_main(void* arg)
{
    worker_t* w = (worker_t*)arg; shared_t* sh = w->sh; orc_model* o = sh->o;
    const int K = o->K, M = o->M;
    int32_t* ltc = (int32_t*)malloc((size_t)M * K * sizeof(int32_t));
    int32_t* lti = (int32_t*)malloc((size_t)(K + 1) * sizeof(int32_t));
    double* tdm = (double*)malloc((size_t)(K + 1) * sizeof(double));
    double* oth = (double*)malloc((size_t)K * sizeof(double));
    for (int64_t d = w->startDoc; d < o->D && d < w->startDoc + w->numDocs; d++)   /* WRK:192-194 */
        worker_doc(w, d, ltc, lti, tdm, oth);
    for (int ut = 0; ut < sh->nut; ut++) {                                           /* WRK:216-218 */
        qdelta s = { -1, -1, -1, -1, -1, -1 };
        q_put(&sh->queues[sh->nst * ut + w->id], s);
    }
    free(ltc); free(lti); free(tdm); free(oth);
    pthread_barrier_wait(&sh->barrier);                                              /* WRK:221 */
    return NULL;
}

static void* updater_main(void* arg)
{
    updater_t* u = (updater_t*)arg; shared_t* sh = u->sh; orc_model* o = sh->o;
    const int K = o->K;
    int finished_cnt = 0;
    char* finished = (char*)calloc((size_t)sh->nst, 1);
    int isFinished = 0;
    while (!isFinished) {                                                            /* UPD:181 */
        for (int st = 0; st < sh->nst; st++) {
            qdelta dl;
            while (q_poll(&sh->queues[u->id * sh->nst + st], &dl)) {                 /* UPD:187 */
                if (dl.mod == -1 && dl.newT == -1 && dl.oldT == -1 && dl.type == -1) {
                    if (!finished[st]) { finished[st] = 1; finished_cnt++; }
                    isFinished = finished_cnt == sh->nst;
                    continue;
                }
                int m = dl.mod;
                size_t trow = (size_t)(o->rowbase[m] + dl.type);
                int32_t* cnt = o->nwk + trow * K;
                _Atomic int32_t* nk = (_Atomic int32_t*)(o->nk + (size_t)m * K);
                _Atomic int32_t* h = sh->hist + sh->hist_off[m];
                int hl = sh->hist_len[m];
                if (dl.oldT != -1) cnt[dl.oldT]--;                                   /* UPD:199-206 */
                cnt[dl.newT]++;
                if (dl.oldT != -1) atomic_fetch_sub_explicit(&nk[dl.oldT], 1, memory_order_relaxed);
                atomic_fetch_add_explicit(&nk[dl.newT], 1, memory_order_relaxed);
                if (dl.oldT != -1) {                                                 /* UPD:220-227 */
                    if (dl.docOld + 1 < hl) atomic_fetch_sub_explicit(&h[(size_t)dl.oldT * hl + dl.docOld + 1], 1, memory_order_relaxed);
                    if (dl.docOld > 0 && dl.docOld < hl) atomic_fetch_add_explicit(&h[(size_t)dl.oldT * hl + dl.docOld], 1, memory_order_relaxed);
                }
                if (dl.docNew > 1 && dl.docNew - 1 < hl) atomic_fetch_sub_explicit(&h[(size_t)dl.newT * hl + dl.docNew - 1], 1, memory_order_relaxed);
                if (dl.docNew < hl) atomic_fetch_add_explicit(&h[(size_t)dl.newT * hl + dl.docNew], 1, memory_order_relaxed);
                double* tree = o->trees + trow * 2 * K;
                tlock(&sh->tree_lock[trow]);                                         /* UPD:242-260 */
                if (dl.oldT != -1) {
                    double p_wt = (cnt[dl.oldT] + o->beta[m]) / (atomic_load_explicit(&nk[dl.oldT], memory_order_relaxed) + o->beta_sum[m]);
                    orc_ftree_update(tree, K, dl.oldT, o->gamma[m] * o->alpha[(size_t)m * (K + 1) + dl.oldT] * p_wt);
                }
                double p_wt_new = (cnt[dl.newT] + o->beta[m]) / (atomic_load_explicit(&nk[dl.newT], memory_order_relaxed) + o->beta_sum[m]);
                orc_ftree_update(tree, K, dl.newT, o->gamma[m] * o->alpha[(size_t)m * (K + 1) + dl.newT] * p_wt_new);
                tunlock(&sh->tree_lock[trow]);
                if (o->inactive[dl.newT]) {                                          /* UPD:263-270 */
                    o->inactive[dl.newT] = 0;
                    o->alpha[(size_t)m * (K + 1) + dl.newT] = o->alpha[(size_t)m * (K + 1) + K];
                }
            }
        }
        usleep(20000);                                                               /* UPD:276-280 */
    }
    free(finished);
    pthread_barrier_wait(&sh->barrier);                                              /* UPD:286 */
    return NULL;
}

double orc_threaded_estimate(orc_model* o, int num_threads, int iters, uint64_t seed, orc_stats* st)
{
    const int K = o->K, M = o->M;
    int nst = 3 * num_threads / 4, nut = num_threads / 4;                            /* PTM:1036-1037 */
    if (nst < 1 || nut < 1) return -1.0;                                             /* Q10 */
    shared_t sh; memset(&sh, 0, sizeof sh);
    sh.o = o; sh.nst = nst; sh.nut = nut;
    sh.queues = (spsc*)calloc((size_t)nst * nut, sizeof(spsc));
    for (int i = 0; i < nst * nut; i++) q_init(&sh.queues[i]);
    int64_t sumV = o->rowbase[M];
    sh.tree_lock = (atomic_flag*)calloc((size_t)sumV, sizeof(atomic_flag));
    for (int64_t i = 0; i < sumV; i++) atomic_flag_clear(&sh.tree_lock[i]);
    /* topicDocCounts histograms (PTM:886-896, 647-649) */
    int64_t htot = 0;
    for (int m = 0; m < M; m++) {
        int mx = 0;
        for (int64_t d = 0; d < o->D; d++) { int l = (int)(o->doc_off[m][d + 1] - o->doc_off[m][d]); if (l > mx) mx = l; }
        sh.hist_len[m] = mx + 1; sh.hist_off[m] = htot; htot += (int64_t)K * (mx + 1);
    }
    sh.hist = (_Atomic int32_t*)calloc((size_t)htot, sizeof(int32_t));
    for (int m = 0; m < M; m++) {
        int32_t* tmp = (int32_t*)malloc((size_t)K * sh.hist_len[m] * sizeof(int32_t));
        orc_get_doc_topic_hist(o, m, tmp, sh.hist_len[m], NULL, 0);
        for (int64_t i = 0; i < (int64_t)K * sh.hist_len[m]; i++) atomic_store(&sh.hist[sh.hist_off[m] + i], tmp[i]);
        free(tmp);
    }
    orc_build_trees(o);                                                              /* PTM:531 */

    worker_t* ws = (worker_t*)calloc((size_t)nst, sizeof(worker_t));
    updater_t* us = (updater_t*)calloc((size_t)nut, sizeof(updater_t));
    int64_t docsPerThread = o->D / nst, offset = 0;                                  /* PTM:1051-1098 */
    for (int t = 0; t < nst; t++) {
        if (t == nst - 1) docsPerThread = o->D - offset;
        ws[t].sh = &sh; ws[t].id = t; ws[t].startDoc = offset; ws[t].numDocs = docsPerThread;
        orc_jrand_seed(&ws[t].random, (int64_t)seed);                                /* Q12: all seeded alike */
        ws[t].tlr = seed * 0x9E3779B97F4A7C15ULL + (uint64_t)t * 0xD1B54A32D192ED03ULL + 1;
        offset += docsPerThread;
    }
    for (int u = 0; u < nut; u++) { us[u].sh = &sh; us[u].id = u; }
    pthread_t* th = (pthread_t*)calloc((size_t)(nst + nut), sizeof(pthread_t));

    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int it = 1; it <= iters; it++) {                                            /* PTM:1146 */
        if (it < 200 && M > 1) {                                                     /* PTM:1166-1171 (burninPeriod default 200) */
            double v = fmin((double)it / 100 + 0.3, 1.1);
            for (int i = 0; i < M; i++) for (int j = 0; j < M; j++) o->p_a[i][j] = v;
        }
        pthread_barrier_init(&sh.barrier, NULL, (unsigned)(nst + nut + 1));          /* PTM:1038 */
        for (int u = 0; u < nut; u++) pthread_create(&th[nst + u], NULL, updater_main, &us[u]);  /* PTM:1213-1216 */
        for (int t = 0; t < nst; t++) pthread_create(&th[t], NULL, worker_main, &ws[t]);         /* PTM:1219-1229 */
        pthread_barrier_wait(&sh.barrier);                                           /* PTM:1232 */
        for (int t = 0; t < nst + nut; t++) pthread_join(th[t], NULL);
        pthread_barrier_destroy(&sh.barrier);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);

    if (st) {
        memset(st, 0, sizeof *st);
        st->tokens = atomic_load(&sh.tokens); st->changed = atomic_load(&sh.changed);
        st->new_mass_cnt = atomic_load(&sh.newMassCnt);
        st->topic_doc_mass_cnt = atomic_load(&sh.topicDocMassCnt);
        st->word_ftree_mass_cnt = atomic_load(&sh.wordFTreeMassCnt);
        st->activated_topic = -1; st->activated_modality = -1; st->activation_key = INT64_MAX;
    }
    for (int i = 0; i < nst * nut; i++) q_free(&sh.queues[i]);
    free(sh.queues); free(sh.tree_lock); free((void*)sh.hist); free(ws); free(us); free(th);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
