// knowceans_samplers.h — the parts of org.knowceans.util (vendored in source by the reference,
// src/main/java/org/knowceans/util/) that optimizeDP PTM:2440-2591 and optimizeGamma PTM:2369-2438 draw
// from, written against this build's own RNG plumbing:
//   * Cokus            — Cokus.java: MT19937 seeded by the 69069 LCG; the static generator seeds itself with
//                        4357 on first use (Cokus.java:158-159, `left` starts at -1 and is pre-decremented).
//   * StaticSamplers   — Samplers.java:1024-1110: the static Stirling-number cache `allss`, stirling(nn),
//                        randAntoniak(alpha, n), randMultDirect, binarySearch.  Quirk kept on purpose:
//                        randAntoniak scales and then prefix-sums the CACHED row in place
//                        (Samplers.java:1090-1095, 646-649), so later calls with the same n, and every row
//                        derived from it afterwards, see the modified numbers.
//   * RandomSamplers   — RandomSamplers.java:267-367,789-795 over any source with nextDouble().
// Host-side, off the GPU path: a few thousand draws per optimisation step.
#pragma once
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace mvtm {

class Cokus {
public:
    static constexpr int N = 624, M = 397;
    void seed(int32_t seed)
    {
        uint64_t x = (uint64_t)(uint32_t)(seed | 1);
        left_ = 0;
        state_[0] = (uint32_t)x;
        for (int j = 1; j <= N; j++) { x *= 69069ULL; state_[j] = (uint32_t)x; }
    }
    uint32_t rand()
    {
        if (--left_ < 0) return reload();
        return temper(state_[next_++]);
    }
    double randDouble() { return (double)rand() / 4294967296.0; }      // Cokus.java:217-220

private:
    static uint32_t temper(uint32_t y)
    {
        y ^= y >> 11;
        y ^= (y << 7) & 0x9D2C5680u;
        y ^= (y << 15) & 0xEFC60000u;
        return y ^ (y >> 18);
    }
    static uint32_t twist(uint32_t s0, uint32_t s1) { return (((s0 & 0x80000000u) | (s1 & 0x7FFFFFFFu)) >> 1) ^ ((s1 & 1u) ? 0x9908b0dfu : 0u); }
    uint32_t reload()
    {
        if (left_ < -1) seed(4357);
        left_ = N - 1;
        next_ = 1;
        for (int k = 0; k < N - M; k++) state_[k] = state_[k + M] ^ twist(state_[k], state_[k + 1]);
        for (int k = N - M; k < N - 1; k++) state_[k] = state_[k + M - N] ^ twist(state_[k], state_[k + 1]);
        state_[N - 1] = state_[M - 1] ^ twist(state_[N - 1], state_[0]);
        return temper(state_[0]);
    }
    uint32_t state_[N + 1] = {0};
    int next_ = 0;
    int left_ = -1;
};

// The static half of org.knowceans.util.Samplers.  One instance plays the role of the JVM-wide statics.
class StaticSamplers {
public:
    static constexpr int MAXSTIRLING = 20000;                           // Samplers.java:1024
    Cokus cokus;

    // Samplers.java:1051-1077.  Returns the cached row for nn (length nn), normalised to max 1.
    std::vector<double>& stirling(int nn)
    {
        if (nn < 1) throw std::out_of_range("stirling: index -1");      // allss[nn - 1]
        if (allss_.empty()) { allss_.resize(1); allss_[0] = {1.0}; logmaxss_ = {0.0}; }
        if (nn > maxnn_) {
            // rows maxnn..MAXSTIRLING-1 the failed Java loop would write are never read before being
            // rewritten, so raising at once is indistinguishable from the ArrayIndexOutOfBoundsException at row 20000
            if (nn > MAXSTIRLING) throw std::out_of_range("stirling: row beyond MAXSTIRLING");
            if ((int)allss_.size() < nn) { allss_.resize(nn); logmaxss_.resize(nn); }
            for (int mm = maxnn_; mm < nn; mm++) {
                const std::vector<double>& prev = allss_[mm - 1];
                const int len = (int)prev.size() + 1;
                std::vector<double> row((size_t)len, 0.0);
                for (int xx = 0; xx < len; xx++) {
                    row[xx] += (xx < len - 1) ? prev[xx] * mm : 0;
                    row[xx] += (xx == 0) ? 0 : prev[xx - 1];
                }
                double mss = row[0];
                for (int i = 1; i < len; i++) if (row[i] > mss) mss = row[i];
                const double inv = 1 / mss;
                for (int i = 0; i < len; i++) row[i] *= inv;
                allss_[mm] = std::move(row);
                logmaxss_[mm] = logmaxss_[mm - 1] + std::log(mss);
            }
            maxnn_ = nn;
        }
        lmss = logmaxss_[nn - 1];
        return allss_[nn - 1];
    }

    // Samplers.java:691-712
    static int binarySearch(const std::vector<double>& a, double p)
    {
        if (p < a[0]) return 0;
        int low = 0, high = (int)a.size() - 1;
        while (low <= high) {
            int mid = (low + high) >> 1;
            double midVal = a[mid];
            if (midVal < p) low = mid + 1;
            else if (midVal > p) {
                if (mid - 1 < 0) throw std::out_of_range("binarySearch: index -1");
                if (a[mid - 1] < p) return mid;
                high = mid - 1;
            } else return mid;
        }
        return (int)a.size();
    }

    // Samplers.java:644-661 (prefix sums in place, one Cokus draw)
    int randMultDirect(std::vector<double>& pp)
    {
        size_t i;
        for (i = 1; i < pp.size(); i++) pp[i] += pp[i - 1];
        double randNum = cokus.randDouble() * pp[i - 1];
        lastRand = randNum;
        return binarySearch(pp, randNum);
    }

    // Samplers.java:1088-1110
    int randAntoniak(double alpha, int n)
    {
        std::vector<double>& p = stirling(n);
        double aa = 1;
        for (size_t m = 0; m < p.size(); m++) { p[m] *= aa; aa *= alpha; }
        return randMultDirect(p) + 1;
    }

    double lmss = 0, lastRand = 0;

private:
    std::vector<std::vector<double>> allss_;
    std::vector<double> logmaxss_;
    int maxnn_ = 1;
};

// RandomSamplers.java over a java.util.Random-like source.  The reference constructs it over
// ThreadLocalRandom.current() (PTM:236), which cannot be seeded; this build injects a seeded stream.
template <class Rand>
class RandomSamplers {
public:
    explicit RandomSamplers(Rand* r) : rand_(r) {}
    double drand() { return rand_->nextDouble(); }

    double randGamma(double rr)                                         // RandomSamplers.java:294-335
    {
        if (rr <= 0.0) return 0.0;
        if (rr == 1.0) return -std::log(drand());
        if (rr < 1.0) {
            const double cc = 1.0 / rr, dd = 1.0 / (1.0 - rr);
            for (;;) {
                const double xx = std::pow(drand(), cc);
                const double yy = xx + std::pow(drand(), dd);
                if (yy <= 1.0) return -std::log(drand()) * xx / yy;
            }
        }
        const double bb = rr - 1.0, cc = 3.0 * rr - 0.75;
        for (;;) {
            const double uu = drand(), vv = drand();
            const double ww = uu * (1.0 - uu);
            const double yy = std::sqrt(cc / ww) * (uu - 0.5);
            const double xx = bb + yy;
            if (xx >= 0) {
                const double zz = 64.0 * ww * ww * ww * vv * vv;
                if ((zz <= (1.0 - 2.0 * yy * yy / xx)) || (std::log(zz) <= 2.0 * (bb * std::log(xx / bb) - yy))) return xx;
            }
        }
    }
    double randGamma(double shape, double scale) { return randGamma(shape) * scale; }   // :366-368

    double randBeta(double aa, double bb)                               // :267-271 via randDir :478-489
    {
        double w0 = randGamma(aa), w1 = randGamma(bb);
        double sum = 0;
        sum += w0; sum += w1;
        return w0 / sum;
    }
    int randBernoulli(double p) { return drand() < p ? 1 : 0; }         // :789-795

private:
    Rand* rand_;
};

}  // namespace mvtm
