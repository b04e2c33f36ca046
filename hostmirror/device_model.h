// device_model.h — where the host class keeps its model: ONE mvhdp handle, or document shards behind an mvhdp_group
// (include/mvhdp.h "document shards").  Every call the host makes goes through here and is routed to the single-handle entry
// point or to its mvhdp_group_* counterpart; nothing of the reference is restated in this file.  With one shard the calls are
// exactly the ones the class made before this file existed.
//
// Shards are contiguous entity ranges balanced by token count, each created with doc_id_base = the global index of its first
// entity (so that every entity draws from the random streams of its global id: the sharded chain is the single handle's chain).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../include/mvhdp.h"

namespace mvtm {

class DeviceModel {
public:
    ~DeviceModel() { destroy(); }

    bool created() const { return !shards_.empty(); }
    int numShards() const { return (int)shards_.size(); }
    mvhdp_handle first() const { return shards_.empty() ? nullptr : shards_[0]; }     // the replicated model: any member's
    mvhdp_group group() const { return g_; }
    const std::vector<int64_t>& bounds() const { return lo_; }                         // [shards + 1] first entity of each shard, then D

    const char* lastError() const
    {
        if (!err_.empty()) return err_.c_str();
        if (g_ && group_err_) return mvhdp_group_last_error(g_);
        return mvhdp_last_error(err_handle_);
    }

    void destroy()
    {
        if (g_) { mvhdp_group_destroy(g_); g_ = nullptr; }
        for (mvhdp_handle h : shards_) mvhdp_destroy(h);
        shards_.clear(); lo_.clear();
    }

    // entity_tokens[d] = tokens of entity d over all views (what the cut is balanced by)
    int create(const mvhdp_config& cfg, int n_shards, const std::vector<int64_t>& entity_tokens)
    {
        destroy();
        err_.clear();
        const int64_t D = (int64_t)entity_tokens.size();
        if (n_shards < 1) n_shards = 1;
        if ((int64_t)n_shards > D && D > 0) n_shards = (int)D;
        lo_.assign(1, 0);
        if (n_shards > 1) {
            int64_t total = 0;
            for (int64_t t : entity_tokens) total += t;
            int64_t acc = 0, d = 0;
            for (int s = 1; s < n_shards; s++) {
                const int64_t want = total * s / n_shards;
                while (d < D - (n_shards - s) && acc < want) acc += entity_tokens[(size_t)d++];
                if (d <= lo_.back()) d = lo_.back() + 1;                               // every shard holds at least one entity
                lo_.push_back(d);
            }
        }
        lo_.push_back(D);
        for (int s = 0; s < n_shards; s++) {
            mvhdp_config c = cfg;
            c.doc_id_base = cfg.doc_id_base + lo_[(size_t)s];
            mvhdp_handle h = nullptr;
            int rc = mvhdp_create(&c, &h);
            if (rc != MVHDP_OK) { err_ = std::string("mvhdp_create: ") + mvhdp_last_error(nullptr); destroy(); return rc; }
            shards_.push_back(h);
        }
        if (n_shards > 1) {
            int rc = mvhdp_group_create(n_shards, shards_.data(), &g_);
            if (rc != MVHDP_OK) { err_ = std::string("mvhdp_group_create: ") + mvhdp_group_last_error(nullptr); destroy(); return rc; }
        }
        return MVHDP_OK;
    }

    // off [D+1], tok [off[D]]: the whole corpus of view m; every shard takes its slice (offsets rebased)
    int setCorpus(int m, int64_t D, const int64_t* off, const int32_t* tok)
    {
        if (shards_.size() == 1) return one(mvhdp_set_corpus(shards_[0], m, D, off, tok), shards_[0]);
        for (size_t s = 0; s < shards_.size(); s++) {
            const int64_t a = lo_[s], b = lo_[s + 1];
            std::vector<int64_t> o((size_t)(b - a) + 1);
            for (int64_t d = a; d <= b; d++) o[(size_t)(d - a)] = off[d] - off[a];
            int rc = one(mvhdp_set_corpus(shards_[s], m, b - a, o.data(), tok + off[a]), shards_[s]);
            if (rc) return rc;
        }
        view_off_[m].assign(lo_.size(), 0);
        for (size_t s = 0; s < lo_.size(); s++) view_off_[m][s] = off[lo_[s]];
        return MVHDP_OK;
    }
    int setAssignments(int m, const int32_t* z)
    {
        if (shards_.size() == 1) return one(mvhdp_set_assignments(shards_[0], m, z), shards_[0]);
        for (size_t s = 0; s < shards_.size(); s++) { int rc = one(mvhdp_set_assignments(shards_[s], m, z + view_off_[m][s]), shards_[s]); if (rc) return rc; }
        return MVHDP_OK;
    }
    int getAssignments(int m, int32_t* z)
    {
        if (shards_.size() == 1) return one(mvhdp_get_assignments(shards_[0], m, z), shards_[0]);
        for (size_t s = 0; s < shards_.size(); s++) { int rc = one(mvhdp_get_assignments(shards_[s], m, z + view_off_[m][s]), shards_[s]); if (rc) return rc; }
        return MVHDP_OK;
    }
    int setHyper(const mvhdp_hyper* hy) { return g_ ? grp(mvhdp_group_set_hyper(g_, hy)) : one(mvhdp_set_hyper(shards_[0], hy), shards_[0]); }
    int buildCounts() { return g_ ? grp(mvhdp_group_build_counts(g_)) : one(mvhdp_build_counts(shards_[0]), shards_[0]); }
    int buildTrees()
    {
        for (mvhdp_handle h : shards_) { int rc = one(mvhdp_build_trees(h), h); if (rc) return rc; }
        return MVHDP_OK;
    }
    int getCounts(int m, int32_t* nwk, int32_t* nk) { return one(mvhdp_get_counts(shards_[0], m, nwk, nk), shards_[0]); }
    int getAlpha(double* a, uint8_t* ina) { return one(mvhdp_get_alpha(shards_[0], a, ina), shards_[0]); }
    int docTopicHist(int m, int32_t* hist, int32_t hist_len, int32_t* lens, int32_t len_len)
    {
        return g_ ? grp(mvhdp_group_doc_topic_hist(g_, m, hist, hist_len, lens, len_len)) : one(mvhdp_get_doc_topic_hist(shards_[0], m, hist, hist_len, lens, len_len), shards_[0]);
    }
    int countHistogram(int m, int32_t* hist, int32_t len)
    {
        return g_ ? grp(mvhdp_group_count_histogram(g_, m, hist, len)) : one(mvhdp_get_count_histogram(shards_[0], m, hist, len), shards_[0]);
    }
    int viewOverlapSums(double* sums) { return g_ ? grp(mvhdp_group_view_overlap_sums(g_, sums)) : one(mvhdp_view_overlap_sums(shards_[0], sums), shards_[0]); }
    int gammaDocStatistics(int m, double gamma_m, uint64_t seed, uint32_t round, double* qs, double* qw)
    {
        return g_ ? grp(mvhdp_group_gamma_doc_statistics(g_, m, gamma_m, seed, round, qs, qw)) : one(mvhdp_gamma_doc_statistics(shards_[0], m, gamma_m, seed, round, qs, qw), shards_[0]);
    }
    int dpTableStatistics(int m, const int32_t* hist, int32_t len, const double* conc, uint64_t seed, uint32_t round, double* mk, uint8_t* active)
    {
        return one(mvhdp_dp_table_statistics(shards_[0], m, hist, len, conc, seed, round, mk, active), shards_[0]);   // (the histogram is the whole model's already)
    }
    int antoniakDraws(int n, const int32_t* items, const double* conc, uint64_t seed, uint32_t round, int32_t* tables)
    {
        return one(mvhdp_antoniak_draws(shards_[0], n, items, conc, seed, round, tables), shards_[0]);
    }
    int logLikelihood(double* ll) { return g_ ? grp(mvhdp_group_log_likelihood(g_, ll)) : one(mvhdp_model_log_likelihood(shards_[0], ll), shards_[0]); }

    // one Gibbs sweep of the whole model; st: the counters summed over the shards, times the slowest shard's
    int sweep(uint32_t sweep_idx, uint64_t seed, uint32_t flags, mvhdp_sweep_stats* st)
    {
        if (!g_) return one(mvhdp_sweep(shards_[0], sweep_idx, seed, flags, nullptr, nullptr, st), shards_[0]);
        std::vector<mvhdp_sweep_stats> v(shards_.size());
        int rc = grp(mvhdp_group_sweep(g_, sweep_idx, seed, flags, v.data()));
        if (rc) return rc;
        *st = v[0];
        for (size_t s = 1; s < v.size(); s++) {
            st->tokens += v[s].tokens; st->changed += v[s].changed; st->new_mass_cnt += v[s].new_mass_cnt;
            st->topic_doc_mass_cnt += v[s].topic_doc_mass_cnt; st->word_ftree_mass_cnt += v[s].word_ftree_mass_cnt;
            st->oov_skipped += v[s].oov_skipped; st->aborted_docs += v[s].aborted_docs; st->exact_fallbacks += v[s].exact_fallbacks;
            if (v[s].sweep_kernel_ms > st->sweep_kernel_ms) st->sweep_kernel_ms = v[s].sweep_kernel_ms;
            if (v[s].total_ms > st->total_ms) st->total_ms = v[s].total_ms;
        }
        return MVHDP_OK;
    }

private:
    int one(int rc, mvhdp_handle h) { if (rc) { err_.clear(); err_handle_ = h; group_err_ = false; } return rc; }
    int grp(int rc) { if (rc) { err_.clear(); group_err_ = true; } return rc; }

    std::vector<mvhdp_handle> shards_;
    mvhdp_group g_ = nullptr;
    std::vector<int64_t> lo_;
    std::vector<int64_t> view_off_[MVHDP_MAX_MODALITIES];     // per view: token offset of each shard's first entity
    std::string err_;
    mvhdp_handle err_handle_ = nullptr;
    bool group_err_ = false;
};

}  // namespace mvtm
