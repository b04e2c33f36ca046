"""ctypes binding of the CPU ORACLE (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker.  Nothing under mvtopicmodel_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmvhdp_oracle.so")

MAX_M = 8


class Stats(C.Structure):
    _fields_ = [
        ("tokens", C.c_int64), ("changed", C.c_int64),
        ("new_mass_cnt", C.c_int64), ("topic_doc_mass_cnt", C.c_int64),
        ("word_ftree_mass_cnt", C.c_int64), ("oov_skipped", C.c_int64),
        ("aborted_docs", C.c_int64),
        ("activated_topic", C.c_int32), ("activated_modality", C.c_int32),
        ("activation_key", C.c_int64),
    ]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


class JRand(C.Structure):
    _fields_ = [("s", C.c_uint64), ("have_gauss", C.c_int), ("next_gauss", C.c_double)]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in
            ("mvhdp_oracle.c", "ref_threaded.c", "mvhdp_oracle.h", "mvhdp_oracle_internal.h")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    vp, i32, i64, u32, u64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_double
    P = C.POINTER
    L.orc_philox4x32_10.argtypes = [P(u32), P(u32), P(u32)]
    L.orc_token_uniforms.argtypes = [u64, u32, i64, C.c_int, u32, P(dbl), P(dbl)]
    L.orc_jrand_seed.argtypes = [P(JRand), i64]
    L.orc_jrand_next.argtypes = [P(JRand), C.c_int]; L.orc_jrand_next.restype = i32
    L.orc_jrand_next_int.argtypes = [P(JRand), i32]; L.orc_jrand_next_int.restype = i32
    for f in ("orc_mallet_next_uniform", "orc_mallet_next_gaussian"):
        getattr(L, f).argtypes = [P(JRand)]; getattr(L, f).restype = dbl
    L.orc_mallet_next_beta.argtypes = [P(JRand), dbl, dbl]; L.orc_mallet_next_beta.restype = dbl
    L.orc_ftree_construct.argtypes = [vp, C.c_int, vp]
    L.orc_ftree_sample.argtypes = [vp, C.c_int, dbl]; L.orc_ftree_sample.restype = C.c_int
    L.orc_ftree_update.argtypes = [vp, C.c_int, C.c_int, dbl]
    L.orc_lower_bound.argtypes = [vp, dbl, C.c_int]; L.orc_lower_bound.restype = C.c_int
    L.orc_java_round.argtypes = [dbl]; L.orc_java_round.restype = i64
    L.orc_create.argtypes = [C.c_int, C.c_int, vp]; L.orc_create.restype = vp
    L.orc_destroy.argtypes = [vp]
    L.orc_set_corpus.argtypes = [vp, C.c_int, i64, vp, vp]
    L.orc_set_assignments.argtypes = [vp, C.c_int, vp]
    L.orc_get_assignments.argtypes = [vp, C.c_int, vp]
    L.orc_set_hyper.argtypes = [vp] + [vp] * 8
    L.orc_get_alpha.argtypes = [vp, vp]
    L.orc_get_inactive.argtypes = [vp, vp]
    L.orc_init_assignments.argtypes = [vp, i64]
    L.orc_build_counts.argtypes = [vp]
    L.orc_build_trees.argtypes = [vp]
    L.orc_build_inference_trees.argtypes = [vp]
    L.orc_init_assignments_from_trees.argtypes = [vp, u64, i64]
    L.orc_get_counts.argtypes = [vp, C.c_int, vp, vp]
    L.orc_set_counts.argtypes = [vp, C.c_int, vp, vp]
    L.orc_get_tree.argtypes = [vp, C.c_int, C.c_int, vp]
    L.orc_get_doc_topic_hist.argtypes = [vp, C.c_int, vp, i32, vp, i32]
    L.orc_draw_p_mallet.argtypes = [vp, P(JRand), vp]
    L.orc_draw_p_philox.argtypes = [vp, u64, u32, i64, vp]
    L.orc_sweep.argtypes = [vp, u32, u64, i64, vp, u32, P(Stats), vp, vp, vp, C.c_int, vp, vp, vp, vp]
    L.orc_sweep.restype = C.c_int
    L.orc_sweep_list.argtypes = [vp, u32, u64, i64, vp, u32, P(Stats), vp, vp, vp, i64]
    L.orc_sweep_list.restype = C.c_int
    L.orc_sweep_live_seq.argtypes = [vp, u32, u64, i64, vp, vp, i64, C.c_int, C.c_int, C.c_int, P(Stats)]
    L.orc_sweep_live_seq.restype = C.c_int
    L.orc_row_sample_live.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
    L.orc_row_sample_live.restype = C.c_int
    L.orc_apply_delta.argtypes = [vp, vp, vp, i32, i32]
    L.orc_log_gamma_stirling.argtypes = [dbl]; L.orc_log_gamma_stirling.restype = dbl
    L.orc_mallet_digamma.argtypes = [dbl]; L.orc_mallet_digamma.restype = dbl
    L.orc_learn_symmetric_concentration.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, dbl]
    L.orc_learn_symmetric_concentration.restype = dbl
    L.orc_count_histogram.argtypes = [vp, C.c_int, vp, i32]
    L.orc_optimize_beta.argtypes = [vp, C.c_int, C.c_int, P(dbl)]; L.orc_optimize_beta.restype = dbl
    L.orc_optimize_p_sums.argtypes = [vp, vp]
    L.orc_model_log_likelihood.argtypes = [vp, vp]
    L.orc_threaded_estimate.argtypes = [vp, C.c_int, C.c_int, u64, P(Stats)]
    L.orc_threaded_estimate.restype = dbl
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


SWEEP_REUSE_TREES = 1
SWEEP_NO_APPLY = 2
SWEEP_FROZEN = 16


class Oracle:
    """Thin object wrapper; same verbs as the product's C ABI."""

    def __init__(self, K, V):
        self.L = lib()
        self.K = int(K)
        self.V = [int(v) for v in V]
        self.M = len(self.V)
        varr = np.asarray(self.V, dtype=np.int32)
        self.h = self.L.orc_create(self.K, self.M, _ptr(varr))
        if not self.h:
            raise ValueError("orc_create failed")
        self.N = [0] * self.M
        self.D = 0

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_corpus(self, m, doc_off, tokens):
        doc_off = np.ascontiguousarray(doc_off, dtype=np.int64)
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        self.D = len(doc_off) - 1
        self.N[m] = int(doc_off[-1])
        rc = self.L.orc_set_corpus(self.h, m, self.D, _ptr(doc_off), _ptr(tokens))
        if rc:
            raise ValueError(f"orc_set_corpus rc={rc}")

    def set_assignments(self, m, z):
        z = np.ascontiguousarray(z, dtype=np.int32)
        assert len(z) == self.N[m]
        self.L.orc_set_assignments(self.h, m, _ptr(z))

    def get_assignments(self, m):
        z = np.empty(self.N[m], dtype=np.int32)
        self.L.orc_get_assignments(self.h, m, _ptr(z))
        return z

    def set_hyper(self, alpha, alpha_sum, beta, beta_sum, gamma, p_a, p_b, inactive=None):
        M, K = self.M, self.K
        a = np.ascontiguousarray(alpha, dtype=np.float64).reshape(M, K + 1)
        vs = [np.ascontiguousarray(x, dtype=np.float64).reshape(M) for x in (alpha_sum, beta, beta_sum, gamma)]
        pa = np.ascontiguousarray(p_a, dtype=np.float64).reshape(M, M)
        pb = np.ascontiguousarray(p_b, dtype=np.float64).reshape(M, M)
        ina = None if inactive is None else np.ascontiguousarray(inactive, dtype=np.uint8).reshape(K)
        self.L.orc_set_hyper(self.h, _ptr(a), *[_ptr(v) for v in vs], _ptr(pa), _ptr(pb), _ptr(ina))

    def get_alpha(self):
        a = np.empty((self.M, self.K + 1), dtype=np.float64)
        self.L.orc_get_alpha(self.h, _ptr(a))
        return a

    def get_inactive(self):
        a = np.empty(self.K, dtype=np.uint8)
        self.L.orc_get_inactive(self.h, _ptr(a))
        return a

    def init_assignments(self, seed):
        self.L.orc_init_assignments(self.h, int(seed))

    def build_counts(self):
        self.L.orc_build_counts(self.h)

    def build_trees(self):
        self.L.orc_build_trees(self.h)

    def build_inference_trees(self):
        self.L.orc_build_inference_trees(self.h)

    def init_assignments_from_trees(self, seed, doc_id_base=0):
        self.L.orc_init_assignments_from_trees(self.h, int(seed), int(doc_id_base))

    def get_counts(self, m):
        nwk = np.empty((self.V[m], self.K), dtype=np.int32)
        nk = np.empty(self.K, dtype=np.int32)
        self.L.orc_get_counts(self.h, m, _ptr(nwk), _ptr(nk))
        return nwk, nk

    def set_counts(self, m, nwk, nk):
        nwk = np.ascontiguousarray(nwk, dtype=np.int32)
        nk = np.ascontiguousarray(nk, dtype=np.int32)
        self.L.orc_set_counts(self.h, m, _ptr(nwk), _ptr(nk))

    def get_tree(self, m, w):
        t = np.empty(2 * self.K, dtype=np.float64)
        self.L.orc_get_tree(self.h, m, w, _ptr(t))
        return t

    def get_doc_topic_hist(self, m, hist_len, len_len=0):
        hist = np.empty((self.K, hist_len), dtype=np.int32)
        dl = np.empty(max(len_len, 1), dtype=np.int32)
        self.L.orc_get_doc_topic_hist(self.h, m, _ptr(hist), hist_len, _ptr(dl), len_len)
        return hist, dl[:len_len]

    def draw_p_mallet(self, seed):
        r = JRand()
        self.L.orc_jrand_seed(C.byref(r), int(seed))
        p = np.empty((self.D, self.M, self.M), dtype=np.float64)
        self.L.orc_draw_p_mallet(self.h, C.byref(r), _ptr(p))
        return p

    def draw_p_philox(self, seed, sweep, doc_id_base=0):
        p = np.empty((self.D, self.M, self.M), dtype=np.float64)
        self.L.orc_draw_p_philox(self.h, int(seed), int(sweep), int(doc_id_base), _ptr(p))
        return p

    def sweep(self, sweep_idx, seed, p=None, flags=0, doc_id_base=0, want_delta=False,
              want_dbg=False, trace=None):
        """Returns dict(stats=..., delta_nwk, delta_nk, dbg=[per-view (N,4)], trace=(n,K+1))."""
        st = Stats()
        sumV = sum(self.V)
        dn = np.zeros((sumV, self.K), dtype=np.int32) if want_delta else None
        dk = np.zeros((self.M, self.K), dtype=np.int32) if want_delta else None
        dbg = None
        dbg_ptrs = None
        if want_dbg:
            dbg = [np.zeros((max(self.N[m], 1), 4), dtype=np.float64) for m in range(self.M)]
            dbg_ptrs = (C.c_void_p * self.M)(*[d.ctypes.data for d in dbg])
        if p is not None:
            p = np.ascontiguousarray(p, dtype=np.float64)
            assert p.shape == (self.D, self.M, self.M)
        nt, td, tv, tp, tout = 0, None, None, None, None
        if trace is not None and len(trace) > 0:
            nt = len(trace)
            td = np.ascontiguousarray([t[0] for t in trace], dtype=np.int64)
            tv = np.ascontiguousarray([t[1] for t in trace], dtype=np.int32)
            tp = np.ascontiguousarray([t[2] for t in trace], dtype=np.int32)
            tout = np.zeros((nt, self.K + 1), dtype=np.float64)
        rc = self.L.orc_sweep(self.h, int(sweep_idx), int(seed), int(doc_id_base), _ptr(p), int(flags),
                              C.byref(st), _ptr(dn), _ptr(dk),
                              C.cast(dbg_ptrs, C.c_void_p) if dbg_ptrs is not None else None,
                              nt, _ptr(td), _ptr(tv), _ptr(tp), _ptr(tout))
        if rc:
            raise RuntimeError(f"orc_sweep rc={rc}")
        return dict(stats=st.as_dict(), delta_nwk=dn, delta_nk=dk,
                    dbg=[d[: self.N[m]] for m, d in enumerate(dbg)] if dbg else None, trace=tout)

    def sweep_list(self, sweep_idx, seed, docs, p=None, flags=0, doc_id_base=0, want_delta=False):
        """Deferred sweep over the listed entities only (local indices, list order)."""
        st = Stats()
        docs = np.ascontiguousarray(docs, dtype=np.int64)
        dn = np.zeros((sum(self.V), self.K), dtype=np.int32) if want_delta else None
        dk = np.zeros((self.M, self.K), dtype=np.int32) if want_delta else None
        if p is not None:
            p = np.ascontiguousarray(p, dtype=np.float64)
        rc = self.L.orc_sweep_list(self.h, int(sweep_idx), int(seed), int(doc_id_base), _ptr(p), int(flags), C.byref(st),
                                   _ptr(dn), _ptr(dk), _ptr(docs), len(docs))
        if rc:
            raise RuntimeError(f"orc_sweep_list rc={rc}")
        return dict(stats=st.as_dict(), delta_nwk=dn, delta_nk=dk)

    def sweep_live_seq(self, sweep_idx, seed, order, nseg=1, rows=1, cell16=1, p=None, doc_id_base=0):
        """MVHDP_SWEEP_LIVE as the sequential algorithm it is with one resident wavefront (mvhdp_tuning.single_wave): entities in
        `order`, n_wk deltas landing at every 64-token chunk end, n_k / coefficients / roots of the segment start."""
        st = Stats()
        order = np.ascontiguousarray(order, dtype=np.int64)
        if p is not None:
            p = np.ascontiguousarray(p, dtype=np.float64)
        rc = self.L.orc_sweep_live_seq(self.h, int(sweep_idx), int(seed), int(doc_id_base), _ptr(p), _ptr(order), len(order), int(nseg), int(rows), int(cell16), C.byref(st))
        if rc:
            raise RuntimeError(f"orc_sweep_live_seq rc={rc}")
        return dict(stats=st.as_dict())

    def apply_delta(self, dn, dk, act_topic=-1, act_modality=-1):
        dn = np.ascontiguousarray(dn, dtype=np.int32)
        dk = np.ascontiguousarray(dk, dtype=np.int32)
        self.L.orc_apply_delta(self.h, _ptr(dn), _ptr(dk), int(act_topic), int(act_modality))

    def count_histogram(self, m, length):
        h = np.zeros(length, dtype=np.int32)
        self.L.orc_count_histogram(self.h, m, _ptr(h), length)
        return h

    def optimize_beta(self, m, max_type_count):
        bs = C.c_double()
        b = self.L.orc_optimize_beta(self.h, m, int(max_type_count), C.byref(bs))
        return b, bs.value

    def optimize_p_sums(self):
        s = np.zeros((self.M, self.M), dtype=np.float64)
        self.L.orc_optimize_p_sums(self.h, _ptr(s))
        return s

    def model_log_likelihood(self):
        ll = np.zeros(self.M, dtype=np.float64)
        self.L.orc_model_log_likelihood(self.h, _ptr(ll))
        return ll

    def threaded_estimate(self, num_threads, iters, seed):
        st = Stats()
        secs = self.L.orc_threaded_estimate(self.h, int(num_threads), int(iters), int(seed), C.byref(st))
        return secs, st.as_dict()
