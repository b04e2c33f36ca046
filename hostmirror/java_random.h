// java_random.h — java.util.Random (documented 48-bit LCG) and the parts of
// cc.mallet.util.Randoms (mallet 2.0.8) the host side of the hot path draws
// from: PTM:404-408,500-506 (initial assignments), WRK:327-337 (view weights) and
// PTM:2616 (sampleDirichlet's nextGamma).
#pragma once
#include <cmath>
#include <cstdint>
#include <stdexcept>

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

    // Randoms.nextGamma(alpha, beta, lambda) as compiled in mallet-2.0.8.jar (read with tools/javap_lite.py):
    // alpha < 1: rejection from the b = 1 + alpha/e envelope; alpha == 1: -log U; alpha > 1: Best's rejection.
    // Comparisons are written so that a NaN takes the branch the JVM's dcmpl/dcmpg encoding takes.
    double nextGamma(double alpha, double beta = 1, double lambda = 0)
    {
        double gamma = 0;
        if (alpha <= 0 || beta <= 0) throw std::invalid_argument("alpha and beta must be strictly positive.");
        if (alpha < 1) {
            const double b = 1 + alpha * std::exp(-1.0);
            bool flag = false;
            while (!flag) {
                const double p = b * nextUniform();
                if (p > 1) {
                    gamma = -std::log((b - p) / alpha);
                    if (nextUniform() <= std::pow(gamma, alpha - 1)) flag = true;
                } else {
                    gamma = std::pow(p, 1 / alpha);
                    if (nextUniform() <= std::exp(-gamma)) flag = true;
                }
            }
        } else if (alpha == 1) {
            gamma = -std::log(nextUniform());
        } else {
            const double b = alpha - 1, c = 3.0 * alpha - 0.75;
            bool flag = false;
            while (!flag) {
                const double u = nextUniform(), v = nextUniform();
                const double w = u * (1 - u);
                const double y = std::sqrt(c / w) * (u - 0.5);
                gamma = b + y;
                if (gamma >= 0) {
                    const double z = 64.0 * w * w * w * v * v;
                    flag = z <= 1 - 2.0 * y * y / gamma;
                    if (!flag) flag = std::log(z) <= 2.0 * (b * std::log(gamma / b) - y);
                }
            }
        }
        return beta * gamma + lambda;
    }

private:
    bool haveNextGaussian_ = false;
    double nextGaussian_ = 0;
};

}  // namespace mvtm
