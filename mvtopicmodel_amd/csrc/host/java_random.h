// java_random.h — java.util.Random (documented 48-bit LCG) and the parts of
// cc.mallet.util.Randoms (mallet 2.0.8) the host side of the hot path draws
// from: PTM:404-408,500-506 (initial assignments) and WRK:327-337 (view weights).
#pragma once
#include <cmath>
#include <cstdint>

namespace mvtm {

class JavaRandom {
public:
    explicit JavaRandom(int64_t seed) { setSeed(seed); }
    void setSeed(int64_t seed) { s_ = ((uint64_t)seed ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1); }
    int32_t next(int bits)
    {
        s_ = (s_ * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
        return (int32_t)(uint32_t)(s_ >> (48 - bits));
    }
    int32_t nextInt(int32_t bound)
    {
        int32_t r = next(31);
        int32_t m = bound - 1;
        if ((bound & m) == 0) return (int32_t)(((int64_t)bound * (int64_t)r) >> 31);
        for (int32_t u = r;; u = next(31)) {
            r = u % bound;
            if ((int32_t)((uint32_t)u - (uint32_t)r + (uint32_t)m) >= 0) return r;
        }
    }
    double nextDouble() { return (double)(((int64_t)next(26) << 27) + next(27)) * 0x1.0p-53; }

private:
    uint64_t s_;
};

// cc.mallet.util.Randoms extends java.util.Random
class Randoms : public JavaRandom {
public:
    explicit Randoms(int64_t seed) : JavaRandom(seed) {}
    double nextUniform() { return nextDouble(); }
    double nextGaussian()
    {
        if (!haveNextGaussian_) {
            double v1 = nextUniform(), v2 = nextUniform();
            double x1 = std::sqrt(-2 * std::log(v1)) * std::cos(2 * M_PI * v2);
            double x2 = std::sqrt(-2 * std::log(v1)) * std::sin(2 * M_PI * v2);
            nextGaussian_ = x2; haveNextGaussian_ = true;
            return x1;
        }
        haveNextGaussian_ = false;
        return nextGaussian_;
    }
    double nextBeta(double alpha, double beta)
    {
        if (alpha == 1 && beta == 1) return nextUniform();
        if (alpha >= 1 && beta >= 1) {
            double A = alpha - 1, B = beta - 1, C = A + B, L = C * std::log(C), mu = A / C, sigma = 0.5 / std::sqrt(C);
            double y = nextGaussian(), x = sigma * y + mu;
            while (x < 0 || x > 1) { y = nextGaussian(); x = sigma * y + mu; }
            double u = nextUniform();
            while (std::log(u) >= A * std::log(x / A) + B * std::log((1 - x) / B) + L + 0.5 * y * y) {
                y = nextGaussian(); x = sigma * y + mu;
                while (x < 0 || x > 1) { y = nextGaussian(); x = sigma * y + mu; }
                u = nextUniform();
            }
            return x;
        }
        double v1 = std::pow(nextUniform(), 1 / alpha), v2 = std::pow(nextUniform(), 1 / beta);
        while (v1 + v2 > 1) { v1 = std::pow(nextUniform(), 1 / alpha); v2 = std::pow(nextUniform(), 1 / beta); }
        return v1 / (v1 + v2);
    }

private:
    bool haveNextGaussian_ = false;
    double nextGaussian_ = 0;
};

}  // namespace mvtm
