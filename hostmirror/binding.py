"""ctypes view of the C++ host mirror of FastQMVWVParallelTopicModel / FastQMVWVTopicInferencer
(hostmirror/, hooks declared in include/mvtm_host.h).  The mirror is its own library, hostmirror/lib/libmvtm_host.so: it stands in
for the Java host classes that cannot be compiled in this image and calls the product (libmvhdp.so) through the C ABI."""
import atexit
import ctypes as C
import os
import sys
import weakref

import numpy as np

from mvtopicmodel_amd._lib import SweepStatsC, load_library

HOST_SYMBOLS = [
    "mvtm_last_error", "mvtm_model_new", "mvtm_model_delete", "mvtm_model_configure",
    "mvtm_model_add_instances", "mvtm_model_estimate", "mvtm_model_num_entities",
    "mvtm_model_view_tokens", "mvtm_model_get_view", "mvtm_model_get_counts",
    "mvtm_model_get_log", "mvtm_model_native_handle", "mvtm_init_assignments",
    "mvtm_model_print_state", "mvtm_model_display_top_words", "mvtm_number_format5", "mvtm_model_print_document_topics", "mvtm_java_double_to_string", "mvtm_model_optimize_p", "mvtm_model_optimize_beta", "mvtm_model_log_likelihood", "mvtm_model_get_perplexities",
    "mvtm_model_seed_host_samplers", "mvtm_model_optimize_dp", "mvtm_model_optimize_gamma",
    "mvtm_cokus_stream", "mvtm_rand_antoniak_seq", "mvtm_random_samplers_stream", "mvtm_mallet_next_gamma_stream",
    "mvtm_model_set_live_updates", "mvtm_model_set_shards", "mvtm_model_set_device_gamma_statistics", "mvtm_model_get_inferencer", "mvtm_inferencer_delete", "mvtm_inferencer_configure",
    "mvtm_inferencer_infer", "mvtm_inferencer_num_entities", "mvtm_inferencer_view_tokens", "mvtm_inferencer_get_view",
    "mvtm_inferencer_doc_topics", "mvtm_inferencer_print_document_topics", "mvtm_inferencer_get_stats",
]

HOST_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmvtm_host.so")
_host = None

_ready = False
_live = weakref.WeakSet()          # open models, closed at exit while the HIP runtime is still alive


@atexit.register
def _close_all():
    for m in list(_live):
        try:
            m.close()
        except Exception:
            pass


def _lib():
    global _ready, _host
    if _host is None:
        load_library()                       # the product library first: fails loudly when it has not been built
        if not os.path.exists(HOST_LIB_PATH):
            raise ImportError(f"{HOST_LIB_PATH} is missing: build it with `make -C hostmirror`")
        _host = C.CDLL(HOST_LIB_PATH)
    L = _host
    if not _ready:
        vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
        L.mvtm_last_error.restype = C.c_char_p
        L.mvtm_model_new.argtypes = [i32, i32, dbl, dbl]; L.mvtm_model_new.restype = vp
        L.mvtm_model_delete.argtypes = [vp]; L.mvtm_model_delete.restype = None
        L.mvtm_model_configure.argtypes = [vp, i32, i32, i32, i32, i32, i64]
        L.mvtm_model_add_instances.argtypes = [vp, i32, vp, vp, vp, vp, vp]
        L.mvtm_model_estimate.argtypes = [vp]
        L.mvtm_model_num_entities.argtypes = [vp]; L.mvtm_model_num_entities.restype = i64
        L.mvtm_model_view_tokens.argtypes = [vp, i32]; L.mvtm_model_view_tokens.restype = i64
        L.mvtm_model_get_view.argtypes = [vp, i32, vp, vp, vp, vp]
        L.mvtm_model_get_counts.argtypes = [vp, i32, vp, vp]
        L.mvtm_model_get_log.argtypes = [vp, i32, C.POINTER(dbl), C.POINTER(SweepStatsC)]
        L.mvtm_model_native_handle.argtypes = [vp]; L.mvtm_model_native_handle.restype = vp
        L.mvtm_init_assignments.argtypes = [i32, i32, i64, vp, i64, vp]
        L.mvtm_model_print_state.argtypes = [vp, C.c_char_p]
        L.mvtm_model_display_top_words.argtypes = [vp, i32, i32, C.c_char_p, i32]
        L.mvtm_number_format5.argtypes = [dbl, C.c_char_p, i32]
        L.mvtm_model_print_document_topics.argtypes = [vp, C.c_char_p, dbl, i32, vp, vp]
        L.mvtm_java_double_to_string.argtypes = [dbl, C.c_char_p, i32]
        L.mvtm_model_optimize_p.argtypes = [vp, vp, vp]
        L.mvtm_model_optimize_beta.argtypes = [vp, vp, vp]
        L.mvtm_model_log_likelihood.argtypes = [vp, vp]
        L.mvtm_model_get_perplexities.argtypes = [vp, i32, vp, i32]
        L.mvtm_model_seed_host_samplers.argtypes = [vp, i64, i64]
        L.mvtm_model_optimize_dp.argtypes = [vp, vp, vp, vp, vp]
        L.mvtm_model_optimize_gamma.argtypes = [vp, vp, vp, vp]
        L.mvtm_cokus_stream.argtypes = [i32, vp]
        L.mvtm_rand_antoniak_seq.argtypes = [i32, vp, vp, vp]
        L.mvtm_random_samplers_stream.argtypes = [i64, i32, dbl, dbl, i32, vp]
        L.mvtm_mallet_next_gamma_stream.argtypes = [i64, dbl, dbl, i32, vp]
        L.mvtm_model_set_live_updates.argtypes = [vp, i32, i32]
        L.mvtm_model_set_device_gamma_statistics.argtypes = [vp, i32]
        L.mvtm_model_set_shards.argtypes = [vp, i32]
        L.mvtm_model_get_inferencer.argtypes = [vp, vp, vp]; L.mvtm_model_get_inferencer.restype = vp
        L.mvtm_inferencer_delete.argtypes = [vp]; L.mvtm_inferencer_delete.restype = None
        L.mvtm_inferencer_configure.argtypes = [vp, i32, i32, i32]
        L.mvtm_inferencer_infer.argtypes = [vp, i32, vp, vp, vp, vp, vp, i64]; L.mvtm_inferencer_infer.restype = i64
        L.mvtm_inferencer_num_entities.argtypes = [vp]; L.mvtm_inferencer_num_entities.restype = i64
        L.mvtm_inferencer_view_tokens.argtypes = [vp, i32]; L.mvtm_inferencer_view_tokens.restype = i64
        L.mvtm_inferencer_get_view.argtypes = [vp, i32, vp, vp, vp, vp]
        L.mvtm_inferencer_doc_topics.argtypes = [vp, vp]
        L.mvtm_inferencer_print_document_topics.argtypes = [vp, dbl, i32, vp, i64]; L.mvtm_inferencer_print_document_topics.restype = i64
        L.mvtm_inferencer_get_stats.argtypes = [vp, i32, C.POINTER(SweepStatsC)]
        _ready = True
    return L


def number_format5(v):
    """java.text.NumberFormat.getInstance() with at most 5 fraction digits (PTM:221-222)."""
    buf = C.create_string_buffer(512)
    n = _lib().mvtm_number_format5(float(v), buf, 512)
    return buf.value[:n].decode()


def java_double_to_string(v):
    L = _lib()
    buf = C.create_string_buffer(64)
    n = L.mvtm_java_double_to_string(float(v), buf, 64)
    return buf.value[:n].decode()


def init_assignments(K, doc_off, seed):
    """PTM:465-515 draw order on CSR arrays (java.util.Random(seed)); returns z per view."""
    L = _lib()
    M = len(doc_off)
    offs = [np.ascontiguousarray(o, dtype=np.int64) for o in doc_off]
    D = len(offs[0]) - 1
    z = [np.empty(int(o[-1]), dtype=np.int32) for o in offs]
    op = (C.c_void_p * M)(*[o.ctypes.data for o in offs])
    zp = (C.c_void_p * M)(*[a.ctypes.data for a in z])
    rc = L.mvtm_init_assignments(int(K), M, D, C.cast(op, C.c_void_p), int(seed), C.cast(zp, C.c_void_p))
    if rc:
        raise RuntimeError("mvtm_init_assignments failed")
    return z


def cokus_stream(n):
    """First n 32-bit outputs of a fresh knowceans Cokus generator (self-seeded with 4357)."""
    out = np.empty(n, dtype=np.uint32)
    _lib().mvtm_cokus_stream(int(n), out.ctypes.data)
    return out


def rand_antoniak_seq(alpha, n):
    """Samplers.randAntoniak(alpha[i], n[i]) called in order on fresh statics; -1 where the call threw."""
    a = np.ascontiguousarray(alpha, dtype=np.float64); nn = np.ascontiguousarray(n, dtype=np.int32)
    out = np.empty(len(a), dtype=np.int32)
    _lib().mvtm_rand_antoniak_seq(len(a), a.ctypes.data, nn.ctypes.data, out.ctypes.data)
    return out


def random_samplers_stream(seed, kind, a, b, n):
    kinds = {"gamma": 0, "beta": 1, "bernoulli": 2, "gamma_scale": 3}
    out = np.empty(n, dtype=np.float64)
    if _lib().mvtm_random_samplers_stream(int(seed), kinds[kind], float(a), float(b), int(n), out.ctypes.data):
        raise ValueError(kind)
    return out


def mallet_next_gamma_stream(seed, alpha, beta, n):
    out = np.empty(n, dtype=np.float64)
    if _lib().mvtm_mallet_next_gamma_stream(int(seed), float(alpha), float(beta), int(n), out.ctypes.data):
        raise ValueError(_lib().mvtm_last_error().decode())
    return out


class FastQMVWVParallelTopicModel:
    """Same verbs as the reference class: ctor, set*, addInstances, estimate."""

    def __init__(self, numberOfTopics, numModalities, alpha, beta):
        self.L = _lib()
        self.K, self.M = int(numberOfTopics), int(numModalities)
        self.p = self.L.mvtm_model_new(self.K, self.M, float(alpha), float(beta))
        if not self.p:
            raise ValueError(self.L.mvtm_last_error().decode())
        _live.add(self)
        self._cfg = dict(numIterations=1000, burninPeriod=200, optimizeInterval=50, randomSeed=-1, device=0, docIdBase=0)
        self.V = None

    def close(self):
        if getattr(self, "p", None):
            self.L.mvtm_model_delete(self.p)
            self.p = None

    def __del__(self):
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def _push_cfg(self):
        c = self._cfg
        self.L.mvtm_model_configure(self.p, c["numIterations"], c["burninPeriod"], c["optimizeInterval"],
                                    c["randomSeed"], c["device"], c["docIdBase"])

    def setNumIterations(self, n): self._cfg["numIterations"] = int(n)
    def setBurninPeriod(self, n): self._cfg["burninPeriod"] = int(n)
    def setOptimizeInterval(self, n): self._cfg["optimizeInterval"] = int(n)
    def setRandomSeed(self, s): self._cfg["randomSeed"] = int(s)
    def setNumThreads(self, n): pass
    def setDevice(self, d): self._cfg["device"] = int(d)

    def setLiveUpdates(self, live, tree_rebuilds_per_sweep=0):
        """False: deferred sweeps (the parity contract); True: MVHDP_SWEEP_LIVE, the reference's update discipline."""
        self.L.mvtm_model_set_live_updates(self.p, int(bool(live)), int(tree_rebuilds_per_sweep))

    def setSegmentedUpdates(self, segments=0):
        """MVHDP_SWEEP_SEGMENT_APPLY: deterministic sweeps in `segments` segments with the deltas applied in between."""
        self.L.mvtm_model_set_live_updates(self.p, 2, int(segments))

    def setNumShards(self, n):
        """Before addInstances: the model as n document shards behind an mvhdp_group; every step of estimate() runs over the group."""
        self.L.mvtm_model_set_shards(self.p, int(n))

    def setDeviceGammaStatistics(self, on):
        """optimizeGamma's per-entity sums on the device instead of the reference's sequential host loop."""
        self._dev_stats = (getattr(self, "_dev_stats", 0) & ~1) | int(bool(on))
        self.L.mvtm_model_set_device_gamma_statistics(self.p, self._dev_stats)

    def setDeviceTableStatistics(self, on):
        """optimizeDP's view-table simulation (the Antoniak draws, PTM:2454-2488) on the device instead of the reference's host loop."""
        self._dev_stats = (getattr(self, "_dev_stats", 0) & ~2) | (2 if on else 0)
        self.L.mvtm_model_set_device_gamma_statistics(self.p, self._dev_stats)

    def getInferencer(self, discr_weight=None, p_mean=None):
        """PTM:3457-3463."""
        dw = None if discr_weight is None else np.ascontiguousarray(discr_weight, dtype=np.float64)
        pm = None if p_mean is None else np.ascontiguousarray(p_mean, dtype=np.float64)
        q = self.L.mvtm_model_get_inferencer(self.p, None if dw is None else dw.ctypes.data, None if pm is None else pm.ctypes.data)
        if not q:
            raise RuntimeError(self.L.mvtm_last_error().decode())
        return FastQMVWVTopicInferencer(self.L, q, self.K, self.M)

    def addInstances(self, training):
        """training: per view (name_ids int64[n], off int64[n+1], tokens int32[N], alphabetSize)."""
        M = self.M
        assert len(training) == M
        self._push_cfg()
        names = [np.ascontiguousarray(t[0], dtype=np.int64) for t in training]
        offs = [np.ascontiguousarray(t[1], dtype=np.int64) for t in training]
        toks = [np.ascontiguousarray(t[2], dtype=np.int32) for t in training]
        n_inst = np.array([len(n) for n in names], dtype=np.int64)
        alphabet = np.array([int(t[3]) for t in training], dtype=np.int32)
        self.V = [int(a) for a in alphabet]
        arr = lambda xs: C.cast((C.c_void_p * M)(*[x.ctypes.data for x in xs]), C.c_void_p)
        rc = self.L.mvtm_model_add_instances(self.p, M, n_inst.ctypes.data_as(C.c_void_p), arr(names), arr(offs), arr(toks),
                                             alphabet.ctypes.data_as(C.c_void_p))
        if rc:
            raise RuntimeError(self.L.mvtm_last_error().decode())

    def estimate(self):
        self._push_cfg()
        if self.L.mvtm_model_estimate(self.p):
            raise RuntimeError(self.L.mvtm_last_error().decode())

    def num_entities(self):
        return int(self.L.mvtm_model_num_entities(self.p))

    def get_view(self, m):
        D = self.num_entities()
        N = int(self.L.mvtm_model_view_tokens(self.p, m))
        ids = np.empty(D, dtype=np.int64); off = np.empty(D + 1, dtype=np.int64)
        tok = np.empty(N, dtype=np.int32); top = np.empty(N, dtype=np.int32)
        self.L.mvtm_model_get_view(self.p, m, ids.ctypes.data, off.ctypes.data, tok.ctypes.data, top.ctypes.data)
        return ids, off, tok, top

    def get_counts(self, m):
        nwk = np.empty((self.V[m], self.K), dtype=np.int32); nk = np.empty(self.K, dtype=np.int32)
        self.L.mvtm_model_get_counts(self.p, m, nwk.ctypes.data, nk.ctypes.data)
        return nwk, nk

    def printState(self, filename):
        if self.L.mvtm_model_print_state(self.p, str(filename).encode()):
            raise RuntimeError(self.L.mvtm_last_error().decode())

    def displayTopWords(self, numWords, usingNewLines=False):
        n = self.L.mvtm_model_display_top_words(self.p, int(numWords), int(bool(usingNewLines)), None, 0)
        if n < 0:
            raise RuntimeError(self.L.mvtm_last_error().decode())
        buf = C.create_string_buffer(n + 1)
        self.L.mvtm_model_display_top_words(self.p, int(numWords), int(bool(usingNewLines)), buf, n + 1)
        return buf.value.decode()

    def printDocumentTopics(self, filename, threshold, max_topics, discr_weight=None, p_mean=None):
        dw = None if discr_weight is None else np.ascontiguousarray(discr_weight, dtype=np.float64)
        pm = None if p_mean is None else np.ascontiguousarray(p_mean, dtype=np.float64)
        if self.L.mvtm_model_print_document_topics(self.p, str(filename).encode(), float(threshold), int(max_topics),
                                                   None if dw is None else dw.ctypes.data, None if pm is None else pm.ctypes.data):
            raise RuntimeError(self.L.mvtm_last_error().decode())

    def optimizeP(self):
        pa = np.zeros((self.M, self.M)); pm = np.zeros((self.M, self.M))
        if self.L.mvtm_model_optimize_p(self.p, pa.ctypes.data, pm.ctypes.data):
            raise RuntimeError(self.L.mvtm_last_error().decode())
        return pa, pm

    def optimizeBeta(self):
        b = np.zeros(self.M); bs = np.zeros(self.M)
        if self.L.mvtm_model_optimize_beta(self.p, b.ctypes.data, bs.ctypes.data):
            raise RuntimeError(self.L.mvtm_last_error().decode())
        return b, bs

    def seedHostSamplers(self, samp_seed, random_seed):
        self.L.mvtm_model_seed_host_samplers(self.p, int(samp_seed), int(random_seed))

    def optimizeDP(self):
        alpha = np.zeros((self.M, self.K + 1)); asum = np.zeros(self.M)
        ina = np.zeros(self.K, dtype=np.uint8); tables = np.zeros(self.M + 1)
        if self.L.mvtm_model_optimize_dp(self.p, alpha.ctypes.data, asum.ctypes.data, ina.ctypes.data, tables.ctypes.data):
            raise RuntimeError(self.L.mvtm_last_error().decode())
        return alpha, asum, ina, tables

    def optimizeGamma(self):
        g = np.zeros(self.M); gv = np.zeros(self.M); gr = C.c_double()
        if self.L.mvtm_model_optimize_gamma(self.p, g.ctypes.data, gv.ctypes.data, C.addressof(gr)):
            raise RuntimeError(self.L.mvtm_last_error().decode())
        return g, gv, gr.value

    def modelLogLikelihood(self):
        ll = np.zeros(self.M)
        if self.L.mvtm_model_log_likelihood(self.p, ll.ctypes.data):
            raise RuntimeError(self.L.mvtm_last_error().decode())
        return ll

    def perplexities(self, m, cap=1024):
        out = np.zeros(cap)
        n = self.L.mvtm_model_get_perplexities(self.p, m, out.ctypes.data, cap)
        return out[:n]

    def iteration_log(self):
        out = []
        i = 0
        while True:
            ms = C.c_double(); st = SweepStatsC()
            if self.L.mvtm_model_get_log(self.p, i, C.byref(ms), C.byref(st)):
                break
            out.append((ms.value, {f: getattr(st, f) for f, _ in SweepStatsC._fields_}))
            i += 1
        return out


class FastQMVWVTopicInferencer:
    """org.madgik.MVTopicModel.FastQMVWVTopicInferencer (INF:114-330) through the host mirror; made by
    FastQMVWVParallelTopicModel.getInferencer()."""

    def __init__(self, L, q, K, M):
        self.L, self.q, self.K, self.M = L, q, K, M
        self._cfg = dict(numIterations=10, randomSeed=-1, device=0)       # INF:73
        _live.add(self)

    def close(self):
        if getattr(self, "q", None):
            self.L.mvtm_inferencer_delete(self.q)
            self.q = None

    def __del__(self):
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def setRandomSeed(self, s): self._cfg["randomSeed"] = int(s)
    def setNumIterations(self, n): self._cfg["numIterations"] = int(n)
    def setDevice(self, d): self._cfg["device"] = int(d)

    def inferTopicDistributionsOnNewDocs(self, training):
        """training: per view (name_ids int64[n], off int64[n+1], tokens int32[N]).  Returns the printDocumentTopics text."""
        M = self.M
        assert len(training) == M
        c = self._cfg
        self.L.mvtm_inferencer_configure(self.q, c["numIterations"], c["randomSeed"], c["device"])
        names = [np.ascontiguousarray(t[0], dtype=np.int64) for t in training]
        offs = [np.ascontiguousarray(t[1], dtype=np.int64) for t in training]
        toks = [np.ascontiguousarray(t[2], dtype=np.int32) for t in training]
        n_inst = np.array([len(n) for n in names], dtype=np.int64)
        arr = lambda xs: C.cast((C.c_void_p * M)(*[x.ctypes.data for x in xs]), C.c_void_p)
        n = self.L.mvtm_inferencer_infer(self.q, M, n_inst.ctypes.data_as(C.c_void_p), arr(names), arr(offs), arr(toks), None, 0)
        if n < 0:
            raise RuntimeError(self.L.mvtm_last_error().decode())
        return self.printDocumentTopics(0.03, -1)

    def printDocumentTopics(self, threshold, max_topics):
        n = self.L.mvtm_inferencer_print_document_topics(self.q, float(threshold), int(max_topics), None, 0)
        if n < 0:
            raise RuntimeError(self.L.mvtm_last_error().decode())
        buf = C.create_string_buffer(n + 1)
        self.L.mvtm_inferencer_print_document_topics(self.q, float(threshold), int(max_topics), buf, n + 1)
        return buf.raw[:n].decode()

    def num_entities(self):
        return int(self.L.mvtm_inferencer_num_entities(self.q))

    def get_view(self, m):
        D = self.num_entities()
        N = int(self.L.mvtm_inferencer_view_tokens(self.q, m))
        ids = np.empty(D, dtype=np.int64); off = np.empty(D + 1, dtype=np.int64)
        tok = np.empty(max(N, 1), dtype=np.int32); top = np.empty(max(N, 1), dtype=np.int32)
        self.L.mvtm_inferencer_get_view(self.q, m, ids.ctypes.data, off.ctypes.data, tok.ctypes.data, top.ctypes.data)
        return ids, off, tok[:N], top[:N]

    def doc_topics(self):
        out = np.zeros((self.num_entities(), self.K), dtype=np.float64)
        if self.L.mvtm_inferencer_doc_topics(self.q, out.ctypes.data):
            raise RuntimeError(self.L.mvtm_last_error().decode())
        return out

    def iteration_stats(self):
        out = []
        i = 0
        while True:
            st = SweepStatsC()
            if self.L.mvtm_inferencer_get_stats(self.q, i, C.byref(st)):
                break
            out.append({f: getattr(st, f) for f, _ in SweepStatsC._fields_})
            i += 1
        return out
