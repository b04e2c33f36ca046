"""The N>1 path on CPU: world_size-2 `gloo`, each rank holding a contiguous document shard.
The sharding / all-reduce host logic is the product's (mvtopicmodel_amd.dist); the per-shard
sampler is the test oracle here because there is no GPU (on the GPU box the same functions
drive NativeSampler over RCCL).  A sharded sweep must equal the single-shard sweep bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleShard:
    """Adapter giving the oracle the interface mvtopicmodel_amd.dist expects of a shard."""

    def __init__(self, oracle):
        from oracle import binding
        self.o = oracle
        self.binding = binding
        n = sum(oracle.V) * oracle.K + oracle.M * oracle.K
        self.counts = torch.zeros(n, dtype=torch.int32)
        self.delta = torch.zeros(n, dtype=torch.int32)

    def _split(self, t):
        o = self.o
        nw = sum(o.V) * o.K
        a = t.numpy()
        return a[:nw].reshape(sum(o.V), o.K), a[nw:].reshape(o.M, o.K)

    def build_counts_local(self):
        self.o.build_counts()
        nwk, nk = self._split(self.counts)
        r = 0
        for m in range(self.o.M):
            a, b = self.o.get_counts(m)
            nwk[r:r + self.o.V[m]] = a; nk[m] = b
            r += self.o.V[m]

    def sync(self):
        # counts may have been all-reduced: push them into the oracle
        nwk, nk = self._split(self.counts)
        r = 0
        for m in range(self.o.M):
            self.o.set_counts(m, nwk[r:r + self.o.V[m]], nk[m])
            r += self.o.V[m]

    def sweep_local(self, sweep_idx, seed, flags=0):
        res = self.o.sweep(sweep_idx, seed, flags=self.binding.SWEEP_NO_APPLY, doc_id_base=self.doc_id_base, want_delta=True)
        nwk, nk = self._split(self.delta)
        nwk[:] = res["delta_nwk"]; nk[:] = res["delta_nk"]
        class S: pass
        s = S()
        for k, v in res["stats"].items():
            setattr(s, k, v)
        return s

    def on_stream(self):
        import contextlib
        return contextlib.nullcontext()

    def has_inactive(self):
        return bool(self.o.get_inactive().any())

    def counts_written(self):
        pass

    def apply(self, topic, modality):
        dn, dk = self._split(self.delta)
        self.o.apply_delta(dn, dk, topic, modality)
        # keep the mirror current for the next sync()
        nwk, nk = self._split(self.counts)
        nwk += dn; nk += dk


def _build_case(with_inactive):
    sys.path.insert(0, ROOT)
    from mvtopicmodel_amd import synth
    from mvtopicmodel_amd.native import Hyper
    K, V = 30, [400, 50, 60]
    c = synth.generate(K, V, 90, [25, 4, 6], seed=77, chunk_docs=4096)
    inactive = None
    if with_inactive:
        inactive = np.zeros(K, dtype=np.uint8); inactive[[25, 28]] = 1
    hy = Hyper.defaults(K, V, inactive=inactive)
    if with_inactive:
        hy.alpha[:, K] = 30.0
    return c, hy


def _make_oracle(c, hy, z=None):
    from oracle.binding import Oracle
    o = Oracle(c.K, c.V)
    for m in range(c.M):
        o.set_corpus(m, c.doc_off[m], c.tokens[m])
    o.set_hyper(hy.alpha, hy.alpha_sum, hy.beta, hy.beta_sum, hy.gamma, hy.p_a, hy.p_b, hy.inactive)
    if z is not None:
        for m in range(c.M):
            o.set_assignments(m, z[m])
    return o


def _init_z(c, hy, with_inactive):
    o = _make_oracle(c, hy)
    o.init_assignments(1)
    z = [o.get_assignments(m) for m in range(c.M)]
    if with_inactive:
        for m in range(c.M):
            z[m][np.isin(z[m], [25, 28])] = 2
    return z


def _worker(rank, world, port, with_inactive, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mvtopicmodel_amd import synth
    from mvtopicmodel_amd.dist import build_counts_all_reduce, sweep_all_reduce
    c, hy = _build_case(with_inactive)
    z = _init_z(c, hy, with_inactive)
    tot = sum(np.diff(c.doc_off[m]) for m in range(c.M))
    lo, hi = synth.shard_bounds(tot, world)[rank]
    sub = c.slice_docs(lo, hi)
    zsub = [z[m][c.doc_off[m][lo]:c.doc_off[m][hi]] for m in range(c.M)]
    shard = OracleShard(_make_oracle(sub, hy, zsub))
    shard.doc_id_base = lo
    build_counts_all_reduce(shard)
    acts = []
    for it in range(3):
        sweep_all_reduce(shard, it, 4242)          # default arguments: the key reduction is decided from the hyper-parameters
        acts.append((int(shard.o.get_inactive().sum())))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=lo, hi=hi,
             **{f"z{m}": shard.o.get_assignments(m) for m in range(c.M)},
             **{f"nwk{m}": shard.o.get_counts(m)[0] for m in range(c.M)},
             **{f"nk{m}": shard.o.get_counts(m)[1] for m in range(c.M)},
             alpha=shard.o.get_alpha(), inactive=shard.o.get_inactive())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("with_inactive", [False, True])
def test_two_shards_equal_one_shard(tmp_path, with_inactive):
    world = 2
    port = 29500 + (os.getpid() % 500) + (7 if with_inactive else 0)
    mp.spawn(_worker, args=(world, port, with_inactive, str(tmp_path)), nprocs=world, join=True)
    c, hy = _build_case(with_inactive)
    z = _init_z(c, hy, with_inactive)
    ref = _make_oracle(c, hy, z)
    ref.build_counts()
    for it in range(3):
        ref.sweep(it, 4242)
    parts = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    for m in range(c.M):
        zcat = np.concatenate([p[f"z{m}"] for p in parts])
        assert np.array_equal(zcat, ref.get_assignments(m)), f"assignments differ in view {m}"
        nwk, nk = ref.get_counts(m)
        for p in parts:                      # every replica holds the same global counts
            assert np.array_equal(p[f"nwk{m}"], nwk) and np.array_equal(p[f"nk{m}"], nk)
    for p in parts:
        assert np.array_equal(p["alpha"], ref.get_alpha())
        assert np.array_equal(p["inactive"], ref.get_inactive())
    if with_inactive:
        assert ref.get_inactive().sum() < 2      # at least one topic got activated along the way


def test_shard_bounds_balance_tokens():
    from mvtopicmodel_amd import synth
    rng = np.random.RandomState(0)
    t = rng.poisson(150, 10000)
    b = synth.shard_bounds(t, 8)
    assert b[0][0] == 0 and b[-1][1] == 10000
    assert all(b[i][1] == b[i + 1][0] for i in range(7))
    loads = [t[lo:hi].sum() for lo, hi in b]
    assert max(loads) - min(loads) < 2 * t.max()


def test_activation_key_decoding():
    from mvtopicmodel_amd.dist import KEY_NONE, decode_activation
    assert decode_activation(KEY_NONE) == (-1, -1)
    key = (123456 << 34) | (3 << 31) | (77 << 11) | 1999
    assert decode_activation(key) == (1999, 3)
