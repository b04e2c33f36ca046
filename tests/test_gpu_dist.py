"""GPU side of the sharded path: library-owned HBM buffers aliased as torch tensors and
handed to the RCCL ("nccl") backend.  One GPU only here (the driver runs 2/4/8)."""
import os

import numpy as np
import pytest

from mvtopicmodel_amd.native import Hyper
from tests.helpers import assert_same_state, make_native, make_oracle, small_corpus

pytestmark = pytest.mark.gpu


def test_device_buffers_alias_and_single_rank_rccl_all_reduce():
    import torch
    import torch.distributed as dist
    from mvtopicmodel_amd.dist import GpuShard, build_counts_all_reduce, sweep_all_reduce
    K, V = 50, [600, 80]
    c = small_corpus(K, V, 120, [40, 6], 91)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(2)])
    shard = GpuShard(s, "cuda:0")
    # the tensors alias the library's buffers: counts visible without a copy
    nwk0, nk0 = s.get_counts(0)
    nw = sum(V) * K
    assert np.array_equal(shard.counts[: V[0] * K].cpu().numpy().reshape(V[0], K), nwk0)
    assert np.array_equal(shard.counts[nw: nw + K].cpu().numpy(), nk0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        build_counts_all_reduce(shard)
        t = shard.delta.clone()
        dist.all_reduce(shard.delta)            # RCCL on library-owned memory
        torch.cuda.synchronize()
        assert torch.equal(t, shard.delta)
        for it in range(2):
            o.sweep(it, 5)
            sweep_all_reduce(shard, it, 5)
            assert_same_state(o, s, 2)
        # the exchange step itself, forced although the group has one rank: plain sequence, then the chunked pipeline
        # (all-reduce by row ranges, apply + tree rebuild per range, trees reused by the next sweep)
        o.sweep(2, 5); sweep_all_reduce(shard, 2, 5, pipeline=False, force_exchange=True); assert_same_state(o, s, 2)
        ph = {}
        for it in range(3, 7):
            o.sweep(it, 5)
            sweep_all_reduce(shard, it, 5, force_exchange=True, timings=ph)
            assert_same_state(o, s, 2)
            assert s.trees_current()
            for m, w in [(0, 3), (1, 79)]:
                o.build_trees()
                assert np.array_equal(o.get_tree(m, w), s.get_tree(m, w))
        assert ph["n"] == 4 and "exchange_device" in ph
        # live sweeps through the same exchange (each replica live for its own entities, AD-LDA across replicas): the
        # shard's delta is counts_after - counts_before, the snapshot is restored, the pipeline applies the sum
        from mvtopicmodel_amd.native import SWEEP_LIVE, SWEEP_LIVE_SEGMENTS
        for it in range(7, 10):
            st = sweep_all_reduce(shard, it, 5, flags=SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(3), force_exchange=True, pipeline=(it != 8))
            assert st.tokens == c.total_tokens
            for m in range(2):
                z = s.get_assignments(m)
                nwk, nk = s.get_counts(m)
                ref = np.zeros_like(nwk); np.add.at(ref, (c.tokens[m], z), 1)
                assert nwk.min() >= 0 and np.array_equal(ref, nwk) and np.array_equal(ref.sum(axis=0), nk)
    finally:
        dist.destroy_process_group()
    shard.close()
    s.close()


def test_pipelined_exchange_with_inactive_topics_and_row_range_errors():
    import torch
    import torch.distributed as dist
    from mvtopicmodel_amd._lib import MvhdpError
    from mvtopicmodel_amd.dist import GpuShard, sweep_all_reduce
    K, V = 30, [400, 50, 60]
    c = small_corpus(K, V, 90, [25, 4, 6], 93)
    inactive = np.zeros(K, dtype=np.uint8); inactive[[25, 28]] = 1
    hy = Hyper.defaults(K, V, inactive=inactive)
    hy.alpha[:, K] = 30.0
    o = make_oracle(c, hy)
    z = [o.get_assignments(m) for m in range(3)]
    for m in range(3):
        z[m][np.isin(z[m], [25, 28])] = 2
        o.set_assignments(m, z[m])
    o.build_counts()
    s = make_native(c, hy, z)
    shard = GpuShard(s, "cuda:0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29612")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        acts = 0
        for it in range(4):
            ro = o.sweep(it, 8)
            st = sweep_all_reduce(shard, it, 8, force_exchange=True)
            assert st.activated_topic == ro["stats"]["activated_topic"]
            acts += st.activated_topic >= 0
            assert s.trees_current() == (st.activated_topic < 0)          # an activation changes alpha: trees rebuilt next sweep
            assert_same_state(o, s, 3)
            assert np.array_equal(s.get_alpha()[0], o.get_alpha()) and np.array_equal(s.get_alpha()[1], o.get_inactive())
        assert acts >= 1
        # bracket discipline
        st = shard.sweep_local(9, 8)
        with pytest.raises(MvhdpError):
            s.apply_delta_rows(0, 10)                      # no begin
        s.apply_delta_begin()
        s.apply_delta_rows(0, 100)
        with pytest.raises(MvhdpError):
            s.apply_delta_end(-1, -1)                      # rows missing
        s.apply_delta_begin()                              # (the n_k part is zero by now: applying it again adds nothing)
        s.apply_delta_rows(100, sum(V)); s.apply_delta_rows(0, 100)
        s.apply_delta_end(-1, -1)
    finally:
        dist.destroy_process_group()
    shard.close()
    s.close()


def test_shards_on_one_gpu_equal_single_shard():
    """Two NativeSampler shards (as two ranks would hold them) + a host-side sum of their deltas
    == one sampler over all documents: doc sharding with a global doc id base is exact."""
    from mvtopicmodel_amd import synth
    from mvtopicmodel_amd.native import SWEEP_NO_APPLY
    K, V = 60, [700, 90, 70]
    c = small_corpus(K, V, 150, [40, 5, 6], 92)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    z = [o.get_assignments(m) for m in range(3)]
    tot = sum(np.diff(c.doc_off[m]) for m in range(3))
    bounds = synth.shard_bounds(tot, 2)
    shards = []
    for lo, hi in bounds:
        sub = c.slice_docs(lo, hi)
        zs = [z[m][c.doc_off[m][lo]:c.doc_off[m][hi]] for m in range(3)]
        shards.append(make_native(sub, hy, zs, doc_id_base=lo))
    # global counts on every shard
    glob = [o.get_counts(m) for m in range(3)]
    for sh in shards:
        for m in range(3):
            sh.set_counts(m, *glob[m])
    import torch
    from mvtopicmodel_amd.dist import GpuShard
    gs = [GpuShard(sh, "cuda:0") for sh in shards]
    for it in range(2):
        o.sweep(it, 11)
        for g in gs:
            g.sweep_local(it, 11)
        total = gs[0].delta + gs[1].delta
        for g in gs:
            g.delta.copy_(total)
            torch.cuda.synchronize()
            g.apply(-1, -1)
        for m in range(3):
            zcat = np.concatenate([sh.get_assignments(m) for sh in shards])
            assert np.array_equal(zcat, o.get_assignments(m))
            for sh in shards:
                a, b = sh.get_counts(m)
                assert np.array_equal(a, o.get_counts(m)[0]) and np.array_equal(b, o.get_counts(m)[1])
    for sh in shards:
        sh.close()


def test_sampler_first_torch_second_in_a_fresh_process():
    """a process that creates a NativeSampler BEFORE anything touches torch.cuda, then a GpuShard and a
    1-rank RCCL group: the loader keeps one HIP runtime mapped (mvtopicmodel_amd/_lib.py), so torch's lazy init works."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys\n"
        f"sys.path.insert(0, {root!r})\n"
        "import numpy as np\n"
        "from mvtopicmodel_amd import NativeSampler, synth, _lib\n"
        "from mvtopicmodel_amd.native import Hyper\n"
        "K, V = 20, [200, 30]\n"
        "c = synth.generate(K, V, 60, [20, 4], seed=5)\n"
        "s = NativeSampler(K, V, device=0)\n"
        "rng = np.random.default_rng(1)\n"
        "for m in range(2):\n"
        "    s.set_corpus(m, c.doc_off[m], c.tokens[m]); s.set_assignments(m, rng.integers(0, K, len(c.tokens[m]), dtype=np.int32))\n"
        "s.set_hyper(Hyper.defaults(K, V)); s.build_counts(); s.sweep(0, 1)\n"
        "assert 'torch' not in sys.modules\n"
        "import torch, torch.distributed as dist\n"
        "from mvtopicmodel_amd.dist import GpuShard, sweep_all_reduce\n"
        "assert torch.cuda.is_available()\n"
        "shard = GpuShard(s, 'cuda:0')\n"
        "os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29631')\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))\n"
        "st = sweep_all_reduce(shard, 1, 1, force_exchange=True)\n"
        "assert st.tokens == c.total_tokens\n"
        "m = _lib.mapped_runtime_libraries()\n"
        "assert len(m['libamdhip64']) == 1 and len(m['libhsa-runtime64']) == 1, m\n"
        "dist.destroy_process_group(); shard.close(); s.close()\n"
        "print('order ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "order ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
