"""NativeSampler — Python mirror of the JNI host class
``org.madgik.MVTopicModel.NativeSampler`` shown in INTEGRATION.md: a thin,
numpy-typed wrapper over the C ABI (include/mvhdp.h).  One instance = one model
shard on one MI355X.  All compute happens in libmvhdp.so on the GPU.
"""
import atexit
import ctypes as C
import sys
import weakref
from dataclasses import dataclass, field

import numpy as np

from ._lib import (MAX_M, UNIQUE_ID_BYTES, Config, DebugC, GroupInfoC, HyperC, MvhdpError, SweepStatsC, TuningC, load_library)

SWEEP_REUSE_TREES = 0x1
SWEEP_NO_APPLY = 0x2
SWEEP_EXACT_CHAIN = 0x4
SWEEP_GENERIC_KERNEL = 0x8
SWEEP_FROZEN = 0x10
SWEEP_LIVE = 0x20
SWEEP_SEGMENT_APPLY = 0x40
SWEEP_SEGMENT_OVERLAP = 0x80
SWEEP_ASYNC_EXCHANGE = 0x100


def SWEEP_LIVE_SEGMENTS(n):
    return (int(n) & 0xFF) << 16

def SWEEP_ONLY_SEGMENT(s):
    return ((int(s) + 1) & 0xFF) << 24


BUF_COUNTS = 0
BUF_DELTA = 1


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


_live = weakref.WeakSet()          # open samplers, closed at exit while the HIP runtime is still alive


@atexit.register
def _close_all():
    for s in list(_live):
        try:
            s.close()
        except Exception:
            pass


@dataclass
class Hyper:
    """The hyper-parameters the sampler reads (PTM:79-83,95,130-131)."""
    alpha: np.ndarray          # [M][K+1]
    alpha_sum: np.ndarray      # [M]
    beta: np.ndarray           # [M]
    beta_sum: np.ndarray       # [M]
    gamma: np.ndarray          # [M]
    p_a: np.ndarray            # [M][M]
    p_b: np.ndarray            # [M][M]
    inactive: np.ndarray = None  # [K] uint8 or None

    @staticmethod
    def defaults(K, V, alpha=0.1, beta=0.01, p_a=0.31, p_b=1.0, inactive=None):
        """PTM:207-214 (alpha[m][.]=alpha, alphaSum=K*alpha, gamma=1, beta) + betaSum=beta*V_m PTM:420.
        p_a = 0.31 is iteration 1 of the burn-in schedule PTM:1168."""
        M = len(V)
        return Hyper(alpha=np.full((M, K + 1), alpha, dtype=np.float64),
                     alpha_sum=np.full(M, K * alpha, dtype=np.float64),
                     beta=np.full(M, beta, dtype=np.float64),
                     beta_sum=np.array([beta * v for v in V], dtype=np.float64),
                     gamma=np.ones(M, dtype=np.float64),
                     p_a=np.full((M, M), p_a, dtype=np.float64),
                     p_b=np.full((M, M), p_b, dtype=np.float64),
                     inactive=None if inactive is None else np.asarray(inactive, dtype=np.uint8))


@dataclass
class SweepStats:
    tokens: int = 0
    changed: int = 0
    new_mass_cnt: int = 0
    topic_doc_mass_cnt: int = 0
    word_ftree_mass_cnt: int = 0
    oov_skipped: int = 0
    aborted_docs: int = 0
    exact_fallbacks: int = 0
    activated_topic: int = -1
    activated_modality: int = -1
    activation_key: int = 0
    sweep_kernel_ms: float = 0.0
    total_ms: float = 0.0
    activations: int = 0
    reserved: int = 0
    dbg: list = field(default=None, repr=False)
    trace: np.ndarray = field(default=None, repr=False)


class NativeSampler:
    def __init__(self, K, V, device=0, doc_id_base=0):
        self.L = load_library()
        self.K = int(K)
        self.V = [int(v) for v in V]
        self.M = len(self.V)
        if self.M > MAX_M:
            raise ValueError("too many modalities")
        cfg = Config()
        cfg.num_topics = self.K
        cfg.num_modalities = self.M
        for m, v in enumerate(self.V):
            cfg.num_types[m] = v
        cfg.device = int(device)
        cfg.doc_id_base = int(doc_id_base)
        self.h = C.c_void_p()
        rc = self.L.mvhdp_create(C.byref(cfg), C.byref(self.h))
        _live.add(self)
        if rc != 0:
            msg = self.L.mvhdp_last_error(None).decode()
            self.h = None
            raise MvhdpError(rc, msg)
        self.N = [0] * self.M
        self.D = 0

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.L.mvhdp_destroy(self.h)
            self.h = None

    def __del__(self):
        # at interpreter shutdown the HIP runtime may already be gone: handles still open then were closed by
        # the atexit hook below, while the runtime was alive
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc):
        if rc != 0:
            raise MvhdpError(rc, self.L.mvhdp_last_error(self.h).decode())

    # -- corpus / assignments ----------------------------------------------
    def set_corpus(self, m, doc_off, tokens):
        doc_off = np.ascontiguousarray(doc_off, dtype=np.int64)
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        if doc_off.ndim != 1 or len(doc_off) < 1 or len(tokens) != int(doc_off[-1]):
            raise ValueError("doc_off/tokens shape mismatch")
        self._ck(self.L.mvhdp_set_corpus(self.h, m, len(doc_off) - 1, _ptr(doc_off), _ptr(tokens)))
        self.D = len(doc_off) - 1
        self.N[m] = int(doc_off[-1])

    def set_assignments(self, m, z):
        z = np.ascontiguousarray(z, dtype=np.int32)
        if len(z) != self.N[m]:
            raise ValueError("z length mismatch")
        self._ck(self.L.mvhdp_set_assignments(self.h, m, _ptr(z)))

    def set_view_presence(self, m, present):
        """present: uint8 [D] (1 = the entity has the view even without tokens) or None (inferred from the spans)."""
        if present is None:
            self._ck(self.L.mvhdp_set_view_presence(self.h, m, None))
            return
        p = np.ascontiguousarray(present, dtype=np.uint8)
        if len(p) != self.D:
            raise ValueError("presence mask length mismatch")
        self._ck(self.L.mvhdp_set_view_presence(self.h, m, _ptr(p)))

    def get_assignments(self, m):
        z = np.empty(self.N[m], dtype=np.int32)
        self._ck(self.L.mvhdp_get_assignments(self.h, m, _ptr(z)))
        return z

    # -- model state --------------------------------------------------------
    def set_hyper(self, hy: Hyper):
        M, K = self.M, self.K
        hc = HyperC()
        self._alpha_keep = np.ascontiguousarray(hy.alpha, dtype=np.float64).reshape(M, K + 1)
        hc.alpha = self._alpha_keep.ctypes.data
        for m in range(M):
            hc.alpha_sum[m] = float(hy.alpha_sum[m]); hc.beta[m] = float(hy.beta[m])
            hc.beta_sum[m] = float(hy.beta_sum[m]); hc.gamma[m] = float(hy.gamma[m])
            for j in range(M):
                hc.p_a[m][j] = float(hy.p_a[m][j]); hc.p_b[m][j] = float(hy.p_b[m][j])
        self._inactive_keep = None
        if hy.inactive is not None:
            self._inactive_keep = np.ascontiguousarray(hy.inactive, dtype=np.uint8).reshape(K)
            hc.inactive = self._inactive_keep.ctypes.data
        self._ck(self.L.mvhdp_set_hyper(self.h, C.byref(hc)))

    def get_alpha(self):
        a = np.empty((self.M, self.K + 1), dtype=np.float64)
        ina = np.empty(self.K, dtype=np.uint8)
        self._ck(self.L.mvhdp_get_alpha(self.h, _ptr(a), _ptr(ina)))
        return a, ina

    def build_counts(self):
        self._ck(self.L.mvhdp_build_counts(self.h))

    def build_trees(self):
        self._ck(self.L.mvhdp_build_trees(self.h))

    def build_inference_trees(self):
        self._ck(self.L.mvhdp_build_inference_trees(self.h))

    def init_assignments_from_trees(self, seed):
        self._ck(self.L.mvhdp_init_assignments_from_trees(self.h, int(seed)))

    def get_counts(self, m):
        nwk = np.empty((self.V[m], self.K), dtype=np.int32)
        nk = np.empty(self.K, dtype=np.int32)
        self._ck(self.L.mvhdp_get_counts(self.h, m, _ptr(nwk), _ptr(nk)))
        return nwk, nk

    def set_counts(self, m, nwk, nk):
        nwk = np.ascontiguousarray(nwk, dtype=np.int32)
        nk = np.ascontiguousarray(nk, dtype=np.int32)
        assert nwk.shape == (self.V[m], self.K) and nk.shape == (self.K,)
        self._ck(self.L.mvhdp_set_counts(self.h, m, _ptr(nwk), _ptr(nk)))

    def get_tree(self, m, w):
        t = np.empty(2 * self.K, dtype=np.float64)
        self._ck(self.L.mvhdp_get_tree(self.h, m, w, _ptr(t)))
        return t

    def get_doc_topic_hist(self, m, hist_len, len_len=0):
        hist = np.empty((self.K, hist_len), dtype=np.int32)
        dl = np.empty(max(len_len, 1), dtype=np.int32) if len_len > 0 else None
        self._ck(self.L.mvhdp_get_doc_topic_hist(self.h, m, _ptr(hist), hist_len, _ptr(dl), len_len))
        return hist, (dl[:len_len] if dl is not None else None)

    # -- statistics either side of the sweep ------------------------
    def get_count_histogram(self, m, length):
        h = np.zeros(length, dtype=np.int32)
        self._ck(self.L.mvhdp_get_count_histogram(self.h, m, _ptr(h), length))
        return h

    def view_overlap_sums(self):
        s = np.zeros((self.M, self.M), dtype=np.float64)
        self._ck(self.L.mvhdp_view_overlap_sums(self.h, _ptr(s)))
        return s

    def model_log_likelihood(self):
        ll = np.zeros(self.M, dtype=np.float64)
        self._ck(self.L.mvhdp_model_log_likelihood(self.h, _ptr(ll)))
        return ll

    def gamma_doc_statistics(self, m, gamma_m, seed, round_idx):
        """PTM:2415-2433: (qs, qw) = (sum Bernoulli(j/(j+gamma)), sum log Beta(gamma+1, j)) over the entities with view m."""
        qs, qw = C.c_double(), C.c_double()
        self._ck(self.L.mvhdp_gamma_doc_statistics(self.h, int(m), float(gamma_m), int(seed), int(round_idx), C.byref(qs), C.byref(qw)))
        return qs.value, qw.value

    def doc_topic_proportions(self, view_weights, d0=0, d1=None):
        """PTM:2871-2899: [d1-d0][K] topic proportions, view_weights[m] = (m==0 ? 1 : discrWeight[m]) * pMean[0][m]."""
        d1 = self.D if d1 is None else int(d1)
        w = np.ascontiguousarray(view_weights, dtype=np.float64)
        out = np.zeros((max(d1 - int(d0), 0), self.K), dtype=np.float64)
        self._ck(self.L.mvhdp_doc_topic_proportions(self.h, _ptr(w), int(d0), d1, _ptr(out)))
        return out

    # -- the hot path ---------------------------------------------------------
    def sweep(self, sweep_idx, seed, flags=0, p=None, want_dbg=False, trace=None) -> SweepStats:
        st = SweepStatsC()
        if p is not None:
            p = np.ascontiguousarray(p, dtype=np.float64)
            if p.shape != (self.D, self.M, self.M):
                raise ValueError("p override must be [D][M][M]")
        dbgc = None
        dbg_arrays = None
        tout = None
        keep = []
        if want_dbg or trace:
            dbgc = DebugC()
            if want_dbg:
                dbg_arrays = [np.zeros((max(self.N[m], 1), 4), dtype=np.float64) for m in range(self.M)]
                for m in range(self.M):
                    dbgc.tok_dbg[m] = dbg_arrays[m].ctypes.data
            if trace:
                td = np.ascontiguousarray([t[0] for t in trace], dtype=np.int64)
                tv = np.ascontiguousarray([t[1] for t in trace], dtype=np.int32)
                tp = np.ascontiguousarray([t[2] for t in trace], dtype=np.int32)
                tout = np.zeros((len(trace), self.K + 1), dtype=np.float64)
                keep += [td, tv, tp]
                dbgc.n_trace = len(trace)
                dbgc.trace_doc = td.ctypes.data; dbgc.trace_view = tv.ctypes.data
                dbgc.trace_pos = tp.ctypes.data; dbgc.trace_out = tout.ctypes.data
        rc = self.L.mvhdp_sweep(self.h, int(sweep_idx), int(seed), int(flags), _ptr(p),
                                C.byref(dbgc) if dbgc is not None else None, C.byref(st))
        self._ck(rc)
        out = SweepStats(**{f: getattr(st, f) for f, _ in SweepStatsC._fields_})
        if dbg_arrays is not None:
            out.dbg = [a[: self.N[m]] for m, a in enumerate(dbg_arrays)]
        out.trace = tout
        return out

    def sweep_many(self, first_idx, n, seed, flags=0):
        """n sweeps enqueued back to back, one synchronisation (mvhdp_sweep_many): the list of their statistics."""
        arr = (SweepStatsC * max(int(n), 1))()
        self._ck(self.L.mvhdp_sweep_many(self.h, int(first_idx), int(n), int(seed), int(flags), C.cast(arr, C.c_void_p)))
        return [SweepStats(**{f: getattr(arr[i], f) for f, _ in SweepStatsC._fields_}) for i in range(int(n))]

    # -- tuning (never changes a result) --------------------------------------------
    def get_tuning(self):
        t = TuningC()
        self._ck(self.L.mvhdp_get_tuning(self.h, C.byref(t)))
        return t

    def set_tuning(self, t=None, **kw):
        """set_tuning(force_primary=2, walk_theta=[0.5, 0], narrow=0, ...): fields not named keep their value."""
        if t is None:
            t = self.get_tuning()
        for k, v in kw.items():
            if k in ("walk_theta", "tree_branch_share", "learnt_walk_step"):
                arr = getattr(t, k)
                for i, x in enumerate(v):
                    arr[i] = x
                if k == "walk_theta":
                    for i in range(len(v), MAX_M):
                        arr[i] = 0.0
            else:
                setattr(t, k, v)
        self._ck(self.L.mvhdp_set_tuning(self.h, C.byref(t)))

    def dp_table_statistics(self, m, hist, conc, seed, round_idx):
        """optimizeDP's view-table simulation on the device (mvhdp_dp_table_statistics): (mk [K], active [K])."""
        hist = np.ascontiguousarray(hist, dtype=np.int32)
        conc = np.ascontiguousarray(conc, dtype=np.float64)
        mk = np.zeros(self.K, dtype=np.float64); act = np.zeros(self.K, dtype=np.uint8)
        self._ck(self.L.mvhdp_dp_table_statistics(self.h, int(m), _ptr(hist), int(hist.shape[1]), _ptr(conc), int(seed), int(round_idx), _ptr(mk), _ptr(act)))
        return mk, act

    def antoniak_draws(self, items, conc, seed, round_idx):
        """n independent draws of the number of tables a CRP(conc[j]) makes of items[j] items (mvhdp_antoniak_draws)."""
        items = np.ascontiguousarray(items, dtype=np.int32); conc = np.ascontiguousarray(conc, dtype=np.float64)
        out = np.zeros(len(items), dtype=np.int32)
        self._ck(self.L.mvhdp_antoniak_draws(self.h, len(items), _ptr(items), _ptr(conc), int(seed), int(round_idx), _ptr(out)))
        return out

    def apply_delta(self, activated_topic=-1, activated_modality=-1):
        self._ck(self.L.mvhdp_apply_delta(self.h, int(activated_topic), int(activated_modality)))

    def apply_delta_begin(self):
        self._ck(self.L.mvhdp_apply_delta_begin(self.h))

    def apply_delta_rows(self, row_begin, row_end):
        self._ck(self.L.mvhdp_apply_delta_rows(self.h, int(row_begin), int(row_end)))

    def apply_delta_end(self, activated_topic=-1, activated_modality=-1):
        self._ck(self.L.mvhdp_apply_delta_end(self.h, int(activated_topic), int(activated_modality)))

    def trees_current(self):
        rc = self.L.mvhdp_trees_current(self.h)
        if rc < 0:
            self._ck(rc)
        return bool(rc)

    def get_view_weights(self):
        p = np.empty((self.D, self.M, self.M), dtype=np.float64)
        self._ck(self.L.mvhdp_get_view_weights(self.h, _ptr(p)))
        return p

    # -- interop ----------------------------------------------------------------
    def device_buffer(self, which):
        ptr = C.c_void_p()
        nbytes = C.c_size_t()
        self._ck(self.L.mvhdp_device_buffer(self.h, int(which), C.byref(ptr), C.byref(nbytes)))
        return ptr.value, nbytes.value

    def counts_written(self):
        self._ck(self.L.mvhdp_counts_written(self.h))

    def set_stream(self, hip_stream):
        self._ck(self.L.mvhdp_set_stream(self.h, C.c_void_p(hip_stream) if hip_stream else None))

    def synchronize(self):
        self._ck(self.L.mvhdp_synchronize(self.h))


class NativeGroup:
    """Document shards on several GPUs, exchange step inside the library (include/mvhdp.h mvhdp_group_*): Python mirror of what
    the Java host of INTEGRATION.md calls.  Two ways to form one:

      NativeGroup(samplers)                            one process, one NativeSampler per GPU (or several on one GPU: tests)
      NativeGroup.from_rank(sampler, id, rank, n)      one process per GPU; `id` = NativeGroup.unique_id() of one rank, handed to
                                                       the others by the launcher (torch.distributed's store in bench.py)
    """

    def __init__(self, samplers=None, _handle=None, _members=None):
        self.L = load_library()
        if _handle is not None:
            self.g, self.members = _handle, list(_members)
            return
        self.members = list(samplers)
        arr = (C.c_void_p * len(self.members))(*[s.h for s in self.members])
        self.g = C.c_void_p()
        rc = self.L.mvhdp_group_create(len(self.members), arr, C.byref(self.g))
        if rc != 0:
            self.g = None
            raise MvhdpError(rc, self.L.mvhdp_group_last_error(None).decode())

    @staticmethod
    def unique_id():
        L = load_library()
        buf = (C.c_uint8 * UNIQUE_ID_BYTES)()
        rc = L.mvhdp_group_unique_id(C.cast(buf, C.c_void_p))
        if rc != 0:
            raise MvhdpError(rc, L.mvhdp_group_last_error(None).decode())
        return bytes(buf)

    @classmethod
    def from_rank(cls, sampler, unique_id, rank, nranks):
        L = load_library()
        assert len(unique_id) == UNIQUE_ID_BYTES
        buf = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        g = C.c_void_p()
        rc = L.mvhdp_group_create_rank(sampler.h, C.cast(buf, C.c_void_p), int(rank), int(nranks), C.byref(g))
        if rc != 0:
            raise MvhdpError(rc, L.mvhdp_group_last_error(None).decode())
        return cls(_handle=g, _members=[sampler])

    def _ck(self, rc):
        if rc != 0:
            raise MvhdpError(rc, self.L.mvhdp_group_last_error(self.g).decode())

    def close(self):
        if getattr(self, "g", None):
            self.L.mvhdp_group_destroy(self.g)
            self.g = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def info(self):
        t = GroupInfoC()
        self._ck(self.L.mvhdp_group_get_info(self.g, C.byref(t)))
        return t

    def set_exchange_chunks(self, n):
        self._ck(self.L.mvhdp_group_set_exchange_chunks(self.g, int(n)))

    def build_counts(self):
        self._ck(self.L.mvhdp_group_build_counts(self.g))

    def drain(self):
        """Lands the exchange a SWEEP_ASYNC_EXCHANGE sweep left on the wire: every replica is the global model again."""
        self._ck(self.L.mvhdp_group_drain(self.g))

    def abort(self):
        """This rank cannot go on: its next sweep contributes nothing and fails on every rank together."""
        self._ck(self.L.mvhdp_group_abort(self.g))

    # -- the steps either side of the sweep, for the sharded model (mvhdp_group_*: include/mvhdp.h) --
    def set_hyper(self, hy):
        for s in self.members:
            s.set_hyper(hy)                     # (keeps the arrays alive per member; same effect as mvhdp_group_set_hyper)

    def model_log_likelihood(self):
        M = self.members[0].M
        ll = np.zeros(M, dtype=np.float64)
        self._ck(self.L.mvhdp_group_log_likelihood(self.g, _ptr(ll)))
        return ll

    def get_doc_topic_hist(self, m, hist_len, len_len=0):
        hist = np.empty((self.members[0].K, hist_len), dtype=np.int32)
        dl = np.empty(max(len_len, 1), dtype=np.int32) if len_len > 0 else None
        self._ck(self.L.mvhdp_group_doc_topic_hist(self.g, m, _ptr(hist), hist_len, _ptr(dl), len_len))
        return hist, (dl[:len_len] if dl is not None else None)

    def get_count_histogram(self, m, length):
        h = np.zeros(length, dtype=np.int32)
        self._ck(self.L.mvhdp_group_count_histogram(self.g, m, _ptr(h), length))
        return h

    def view_overlap_sums(self):
        M = self.members[0].M
        s = np.zeros((M, M), dtype=np.float64)
        self._ck(self.L.mvhdp_group_view_overlap_sums(self.g, _ptr(s)))
        return s

    def gamma_doc_statistics(self, m, gamma_m, seed, round_idx):
        qs, qw = C.c_double(), C.c_double()
        self._ck(self.L.mvhdp_group_gamma_doc_statistics(self.g, int(m), float(gamma_m), int(seed), int(round_idx), C.byref(qs), C.byref(qw)))
        return qs.value, qw.value

    def sweep(self, sweep_idx, seed, flags=0):
        """One sweep of the whole model; the list of the local members' statistics."""
        n = len(self.members)
        arr = (SweepStatsC * n)()
        self._ck(self.L.mvhdp_group_sweep(self.g, int(sweep_idx), int(seed), int(flags), C.cast(arr, C.c_void_p)))
        return [SweepStats(**{f: getattr(arr[i], f) for f, _ in SweepStatsC._fields_}) for i in range(n)]
