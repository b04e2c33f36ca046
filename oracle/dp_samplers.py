"""TEST INFRASTRUCTURE ONLY (oracle): pure-Python restatement of the two randomised
hyper-parameter steps either side of the sweep and of the samplers they draw from.
Only tests/ may import this file; the product (mvtopicmodel_amd/) never does.

Follows, in the reference tree:
  optimizeDP        src/main/java/org/madgik/MVTopicModel/FastQMVWVParallelTopicModel.java:2440-2591
  sampleDirichlet   same file :2593-2634
  optimizeGamma     same file :2369-2438
  Cokus             src/main/java/org/knowceans/util/Cokus.java (MT19937; static, self-seeds with 4357)
  Samplers          src/main/java/org/knowceans/util/Samplers.java:644-712 (randMultDirect, binarySearch),
                    :1024-1110 (Stirling cache, stirling, randAntoniak)
  RandomSamplers    src/main/java/org/knowceans/util/RandomSamplers.java:267-271,294-335,366-368,478-489,789-795
  Randoms.nextGamma mallet-2.0.8.jar, cc/mallet/util/Randoms.class (bytecode, tools/javap_lite.py)
  java.util.Random  the documented 48-bit LCG

Parity status: the reference has no tests or vectors for any of this and cannot run here
(no JVM): *parity unpinned* beyond (a) the MT19937 stream, which is checked against
numpy's independent MT19937 in tests/test_dp_samplers.py, and (b) distributional checks.
Two of the three random streams are not reproducible in the reference itself
(`samp` runs over ThreadLocalRandom, `random` is an unseeded Randoms), so what is
compared bit-for-bit is the algorithm under an injected java.util.Random stream.
Python floats are IEEE doubles and math.log/pow/exp/sqrt call the same libm as the C++
host mirror, so equal operation order gives equal bits.
"""
import math

MASK48 = (1 << 48) - 1


class JavaRandom:
    def __init__(self, seed):
        self.s = (seed ^ 0x5DEECE66D) & MASK48

    def next(self, bits):
        self.s = (self.s * 0x5DEECE66D + 0xB) & MASK48
        return self.s >> (48 - bits)            # callers below only use the unsigned value (bits <= 27)

    def nextDouble(self):
        return ((self.next(26) << 27) + self.next(27)) * (1.0 / (1 << 53))

    nextUniform = nextDouble                    # cc.mallet.util.Randoms does not override it


def mallet_next_gamma(rnd, alpha, beta=1.0, lam=0.0):
    """Randoms.nextGamma(DDD)D.  NaN comparisons follow the dcmpl/dcmpg encodings."""
    gamma = 0.0
    if alpha <= 0 or beta <= 0:
        raise ValueError("alpha and beta must be strictly positive.")
    if alpha < 1:
        b = 1 + alpha * math.exp(-1.0)
        done = False
        while not done:
            p = b * rnd.nextUniform()
            if p > 1:
                gamma = -math.log((b - p) / alpha)
                if rnd.nextUniform() <= math.pow(gamma, alpha - 1):
                    done = True
            else:
                gamma = math.pow(p, 1 / alpha)
                if rnd.nextUniform() <= math.exp(-gamma):
                    done = True
    elif alpha == 1:
        gamma = -math.log(rnd.nextUniform())
    else:
        b = alpha - 1
        c = 3.0 * alpha - 0.75
        done = False
        while not done:
            u = rnd.nextUniform()
            v = rnd.nextUniform()
            w = u * (1 - u)
            y = math.sqrt(c / w) * (u - 0.5)
            gamma = b + y
            if gamma >= 0:
                z = 64.0 * w * w * w * v * v
                done = z <= 1 - 2.0 * y * y / gamma
                if not done:
                    done = math.log(z) <= 2.0 * (b * math.log(gamma / b) - y)
    return beta * gamma + lam


class Cokus:
    """MT19937 with the 1998 seeding (x <- 69069 x mod 2^32), first use seeds 4357."""
    N, M = 624, 397

    def __init__(self):
        self.mt = None
        self.idx = 0

    def seed(self, seed):
        x = (seed | 1) & 0xFFFFFFFF
        self.mt = [x]
        for _ in range(self.N - 1):
            x = (x * 69069) & 0xFFFFFFFF
            self.mt.append(x)
        self.idx = self.N

    def rand(self):
        if self.mt is None:
            self.seed(4357)
        if self.idx >= self.N:
            mt, N, M = self.mt, self.N, self.M
            for k in range(N):
                y = (mt[k] & 0x80000000) | (mt[(k + 1) % N] & 0x7FFFFFFF)
                mt[k] = mt[(k + M) % N] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            self.idx = 0
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    def randDouble(self):
        return self.rand() / 4294967296.0


class StaticSamplers:
    """The static state of org.knowceans.util.Samplers: Cokus + the Stirling cache.
    randAntoniak modifies the cached row in place (scale by alpha^m, then prefix sums)."""
    MAXSTIRLING = 20000

    def __init__(self):
        self.cokus = Cokus()
        self.allss = {}
        self.logmaxss = {}
        self.maxnn = 1

    def stirling(self, nn):
        if nn < 1:
            raise IndexError(nn - 1)
        if 0 not in self.allss:
            self.allss[0] = [1.0]
            self.logmaxss[0] = 0.0
        if nn > self.maxnn:
            if nn > self.MAXSTIRLING:
                raise IndexError(self.MAXSTIRLING)      # rows written before the throw are never read again
            for mm in range(self.maxnn, nn):
                prev = self.allss[mm - 1]
                ln = len(prev) + 1
                row = [0.0] * ln
                for xx in range(ln):
                    row[xx] += prev[xx] * mm if xx < ln - 1 else 0
                    row[xx] += 0 if xx == 0 else prev[xx - 1]
                mss = row[0]
                for x in row[1:]:
                    if x > mss:
                        mss = x
                inv = 1 / mss
                self.allss[mm] = [x * inv for x in row]
                self.logmaxss[mm] = self.logmaxss[mm - 1] + math.log(mss)
            self.maxnn = nn
        return self.allss[nn - 1]

    @staticmethod
    def binary_search(a, p):
        if p < a[0]:
            return 0
        low, high = 0, len(a) - 1
        while low <= high:
            mid = (low + high) >> 1
            v = a[mid]
            if v < p:
                low = mid + 1
            elif v > p:
                if mid - 1 < 0:
                    raise IndexError(-1)
                if a[mid - 1] < p:
                    return mid
                high = mid - 1
            else:
                return mid
        return len(a)

    def rand_mult_direct(self, pp):
        for i in range(1, len(pp)):
            pp[i] += pp[i - 1]
        r = self.cokus.randDouble() * pp[-1]
        return self.binary_search(pp, r)

    def rand_antoniak(self, alpha, n):
        p = self.stirling(n)
        aa = 1.0
        for m in range(len(p)):
            p[m] *= aa
            aa *= alpha
        return self.rand_mult_direct(p) + 1


class RandomSamplers:
    def __init__(self, rnd):
        self.rnd = rnd

    def drand(self):
        return self.rnd.nextDouble()

    def rand_gamma(self, rr, scale=None):
        if scale is not None:
            return self.rand_gamma(rr) * scale
        if rr <= 0.0:
            return 0.0
        if rr == 1.0:
            return -math.log(self.drand())
        if rr < 1.0:
            cc = 1.0 / rr
            dd = 1.0 / (1.0 - rr)
            while True:
                xx = math.pow(self.drand(), cc)
                yy = xx + math.pow(self.drand(), dd)
                if yy <= 1.0:
                    return -math.log(self.drand()) * xx / yy
        bb = rr - 1.0
        cc = 3.0 * rr - 0.75
        while True:
            uu = self.drand()
            vv = self.drand()
            ww = uu * (1.0 - uu)
            yy = math.sqrt(cc / ww) * (uu - 0.5)
            xx = bb + yy
            if xx >= 0:
                zz = 64.0 * ww * ww * ww * vv * vv
                if zz <= (1.0 - 2.0 * yy * yy / xx) or math.log(zz) <= 2.0 * (bb * math.log(xx / bb) - yy):
                    return xx

    def rand_beta(self, aa, bb):
        w = [self.rand_gamma(aa), self.rand_gamma(bb)]
        s = 0.0
        for x in w:
            s += x
        return w[0] / s

    def rand_bernoulli(self, p):
        return 1 if self.drand() < p else 0


def sample_dirichlet(random, p):
    magnitude = 0.0
    for x in p:
        magnitude += x
    partition = [x / magnitude for x in p]
    dist = [0.0] * len(p)
    total = 0.0
    for i in range(len(p)):
        if partition[i] * magnitude > 0:
            dist[i] = mallet_next_gamma(random, partition[i] * magnitude, 1)
            if dist[i] <= 0:
                dist[i] = 0.0001
        else:
            dist[i] = 0.0001
        total += dist[i]
    return [x / total for x in dist]


class DPState:
    """The fields the two steps read and write (PTM:136-139 defaults)."""

    def __init__(self, K, M, alpha, gamma):
        self.K, self.M = K, M
        self.alpha = [list(map(float, a)) for a in alpha]         # [M][K+1]
        self.alphaSum = [0.0] * M
        self.gamma = list(map(float, gamma))
        self.gammaRoot = 10.0
        self.gammaView = [0.0] * M
        self.tablesCnt = [0.0] * M
        self.rootTablesCnt = 0.0
        self.inactive = set()


def optimize_dp(st, topic_doc_counts, statics, random):
    """topic_doc_counts[m][t][i] = entities with i tokens of topic t in view m (i = 0..histogramSize[m])."""
    K, M = st.K, st.M
    mk = [[0.0] * (K + 1) for _ in range(M)]
    mk_root = [0.0] * (K + 1)
    st.tablesCnt = [0.0] * M
    st.inactive |= set(range(K))
    for m in range(M):
        for t in range(K):
            tdc = topic_doc_counts[m][t]
            for i in range(len(tdc)):
                c = int(tdc[i])
                if c > 0 and i > 1:
                    st.inactive.discard(t)
                    try:
                        cur = statics.rand_antoniak(st.gamma[m] * st.alpha[m][t], i)
                    except Exception:
                        cur = 1
                    mk[m][t] += c * cur
                elif c > 0 and i == 1:
                    st.inactive.discard(t)
                    mk[m][t] += c
    for t in range(K):
        for m in range(M):
            if mk[m][t] > 1:
                try:
                    cur = statics.rand_antoniak(st.gammaRoot, min(int(math.ceil(mk[m][t])), 2147483647))
                except Exception:
                    cur = 1
                mk_root[t] += cur
            elif mk[m][t] == 1:
                mk_root[t] += 1
    v = [0.0] * (K + 1)
    mk_root[K] = st.gammaRoot
    s = 0.0
    for x in mk_root:
        s += x
    st.rootTablesCnt = s
    num_samples = 10
    for _ in range(num_samples):
        tt = sample_dirichlet(random, mk_root)
        for kk in range(K + 1):
            v[kk] += tt[kk] / float(num_samples)
    for m in range(M):
        for t in range(K):
            mk[m][t] += v[t] * st.gammaRoot
        st.alpha[m] = [0.0] * (K + 1)
        st.alphaSum[m] = 0.0
        mk[m][K] = st.gammaView[m] + v[K] * st.gammaRoot
        s = 0.0
        for x in mk[m]:
            s += x
        st.tablesCnt[m] = s
        for _ in range(num_samples):
            tt = sample_dirichlet(random, mk[m])
            for kk in range(K + 1):
                a = tt[kk] / float(num_samples)
                st.alpha[m][kk] += a
                st.alphaSum[m] += a
    return st


def optimize_gamma(st, doc_length_counts, samp):
    K, M = st.K, st.M
    aalpha, balpha, agamma, bgamma = 5.0, 0.1, 5.0, 0.1
    R = 10
    for _ in range(R):
        eta = samp.rand_beta(st.gammaRoot + 1, st.rootTablesCnt)
        bloge = bgamma - math.log(eta)
        pie = 1. / (1. + (st.rootTablesCnt * bloge / (agamma + K - 1)))
        u = samp.rand_bernoulli(pie)
        st.gammaRoot = samp.rand_gamma(agamma + K - 1 + u, 1. / bloge)
    for m in range(M):
        for _ in range(R):
            prev = st.gamma[m]
            eta = samp.rand_beta(st.gammaView[m] + 1, st.tablesCnt[m])
            bloge = bgamma - math.log(eta)
            pie = 1. / (1. + (st.tablesCnt[m] * bloge / (agamma + K - 1)))
            u = samp.rand_bernoulli(pie)
            st.gammaView[m] = samp.rand_gamma(agamma + K - 1 + u, 1. / bloge)
            qs = 0.0
            qw = 0.0
            for j in range(len(doc_length_counts[m])):
                for _i in range(int(doc_length_counts[m][j])):
                    qs += samp.rand_bernoulli(j / (j + st.gamma[m]))
                    qw += math.log(samp.rand_beta(st.gamma[m] + 1, float(j)))
            st.gamma[m] = samp.rand_gamma(aalpha + st.tablesCnt[m] - qs, 1. / (balpha - qw))
            if st.gamma[m] == 0:
                st.gamma[m] = prev
    return st
