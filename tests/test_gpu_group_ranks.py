"""The library's one-process-per-GPU path with MORE THAN ONE rank (mvhdp_group_create_rank, nranks = 2 and 4): the code that meets an
8-GPU node first, run here as several processes on ONE GPU.  The real RCCL refuses two ranks on one device, so the collective is
tests/native/fake_rccl.c -- the nine entry points the library resolves at run time, over POSIX shared memory, ordered on the HIP stream
like the real one -- selected through MVHDP_RCCL_LIB.  Test infrastructure only: the product never falls back to it.

Every rank is a fresh child process (tests/rank_worker.py); this process holds the references: the oracle (deferred sweeps: bit for
bit), the same shards as members of a one-process group (the segmented sweep across shards: bit for bit), a single handle (the
statistics either side of the sweep), and the invariants of live sweeps (the counts are the counts of the assignments, on every rank).
Failures: a rank whose sweep is refused, a rank whose host raises mvhdp_group_abort, a rank that dies -- the peers must return an error
from the same call, never hang (every run has a timeout)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from mvtopicmodel_amd import NativeGroup, NativeSampler, synth
from mvtopicmodel_amd.native import SWEEP_LIVE_SEGMENTS, SWEEP_SEGMENT_APPLY
from tests import rank_worker as W
from tests.helpers import make_native

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "native", "fake_rccl.c"), "-o", out,
                           "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"])
    return out


def run_ranks(tmp_path, fake, nranks, scenario, timeout=240, slot_bytes=None, collective_timeout_ms=20000, exchange16=None):
    env = dict(os.environ)
    env["MVHDP_RCCL_LIB"] = fake
    if exchange16 is not None:
        env["MVHDP_EXCHANGE16"] = str(exchange16)
    env["FAKE_RCCL_TIMEOUT_MS"] = str(collective_timeout_ms)
    if slot_bytes:
        env["FAKE_RCCL_SLOT_BYTES"] = str(slot_bytes)        # (smaller than a row range: the collective goes in pieces)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rank_worker.py"), str(tmp_path), str(r), str(nranks), scenario],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(nranks)]
    outs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=timeout)
            outs.append(o)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()                                    # (the exact processes started here)
    logs, arrs = [], []
    for r, p in enumerate(procs):
        jp = os.path.join(str(tmp_path), f"rank{r}.json")
        if os.path.exists(jp):
            logs.append(json.load(open(jp))); arrs.append(dict(np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))))
        else:
            logs.append(None); arrs.append(None)
    return procs, outs, logs, arrs


def make_oracle_from(c, hy, z):
    from oracle.binding import Oracle
    o = Oracle(c.K, c.V)
    for m in range(c.M):
        o.set_corpus(m, c.doc_off[m], c.tokens[m]); o.set_assignments(m, z[m])
    o.set_hyper(hy.alpha, hy.alpha_sum, hy.beta, hy.beta_sum, hy.gamma, hy.p_a, hy.p_b, hy.inactive)
    o.build_counts()
    return o


def concat_z(arrs, tag, M):
    return [np.concatenate([a[f"{tag}_z{m}"] for a in arrs]) for m in range(M)]


def assert_all_ok(procs, outs, logs):
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and logs[r] is not None, f"rank {r} failed:\n{o[-3000:]}"
    n = len(procs)
    assert all(lg["ranks"] == n and lg["rccl"] == 1 and lg["rccl_version"] == 1 for lg in logs)      # (version 1: the stand-in, never mistaken for a release)


def assert_replicas_hold(arrs, tag, c, z=None):
    """every rank's counts are the same and are the counts of the concatenated assignments"""
    z = z or concat_z(arrs, tag, c.M)
    for m in range(c.M):
        want = np.zeros((c.V[m], c.K), dtype=np.int64)
        np.add.at(want, (c.tokens[m], z[m]), 1)
        for r, a in enumerate(arrs):
            assert a[f"{tag}_nwk{m}"].min() >= 0
            assert np.array_equal(a[f"{tag}_nwk{m}"].astype(np.int64), want), f"rank {r}: n_wk of view {m} is not the count of the assignments ({tag})"
            assert np.array_equal(a[f"{tag}_nk{m}"].astype(np.int64), want.sum(axis=0)), f"rank {r}: n_k of view {m} ({tag})"


@pytest.mark.parametrize("nranks,scenario,slot,x16", [(2, "deferred", None, None), (4, "deferred", 4096, None), (2, "deferred", None, 0),
                                                      (2, "inactive_deferred", None, None), (4, "inactive_deferred", None, 0)])
def test_deferred_sweeps_and_statistics_across_ranks_equal_the_single_handle(tmp_path, fake_rccl, nranks, scenario, slot, x16):
    """Both widths of the exchange: the first sweep behind a count travels at full width, the second carries every rank's proposal of the
    packed layout, from the third on the deltas of the small rows (at most 32767 tokens of their type) travel two to a word -- fewer
    bytes, the same integers; MVHDP_EXCHANGE16=0 keeps the full width."""
    procs, outs, logs, arrs = run_ranks(tmp_path, fake_rccl, nranks, scenario, slot_bytes=slot, exchange16=x16)
    assert_all_ok(procs, outs, logs)
    for lg in logs:
        assert [e["packed"] for e in lg["events"]] == ([0, 0, 0] if x16 == 0 else [0, 1, 1]), lg["events"]       # (agreed on at the end of the second sweep)
        b = [e["bytes"] for e in lg["events"]]
        assert b[0] == b[1] and (b[2] == b[0] if x16 == 0 else b[2] < 0.6 * b[0]), b
    c, z0 = W.corpus()
    hy = W.hyper(scenario)
    if scenario.startswith("inactive"):
        for m in range(c.M):
            z0[m][np.isin(z0[m], [41, 45])] = 1
    o = make_oracle_from(c, hy, z0)
    acts = []
    for it in range(3):
        acts.append(o.sweep(it, W.SEED)["stats"])
    for r, lg in enumerate(logs):
        assert [e["ok"] for e in lg["events"]] == [True] * 3
        assert [(e["activated_topic"], e["activated_modality"]) for e in lg["events"]] == [(a["activated_topic"], a["activated_modality"]) for a in acts], f"rank {r}"
        assert lg["exchange_ms"] > 0
    if scenario.startswith("inactive"):
        assert any(a["activated_topic"] >= 0 for a in acts)
    z = concat_z(arrs, "end", c.M)
    for m in range(c.M):
        assert np.array_equal(z[m], o.get_assignments(m)), f"assignments of view {m} differ from the oracle's"
        nwk, nk = o.get_counts(m)
        for r, a in enumerate(arrs):
            assert np.array_equal(a[f"end_nwk{m}"], nwk) and np.array_equal(a[f"end_nk{m}"], nk), f"rank {r}: a replica's counts differ in view {m}"
    for a in arrs:
        assert np.array_equal(a["alpha"], o.get_alpha()) and np.array_equal(a["inactive"], o.get_inactive())
    # the statistics: integers exact, sums to rounding (the ranks' partial sums are added in rank order)
    s = make_native(c, hy, z)
    s.set_hyper(_hyper_after(hy, o))
    s.build_counts()
    ll = s.model_log_likelihood()
    maxlen = int(max(np.diff(c.doc_off[m]).max() for m in range(c.M))) + 1
    for r, a in enumerate(arrs):
        assert np.allclose(a["ll"], ll, rtol=1e-12, atol=0), f"rank {r}: LL {a['ll']} against {ll}"
        assert np.allclose(a["overlap"], s.view_overlap_sums(), rtol=1e-12, atol=0)
        for m in range(c.M):
            hist, dl = s.get_doc_topic_hist(m, maxlen, maxlen)
            assert np.array_equal(a[f"hist{m}"], hist) and np.array_equal(a[f"doclen{m}"], dl), f"rank {r}: histograms of view {m}"
            assert np.array_equal(a[f"chist{m}"], s.get_count_histogram(m, 64))
            assert np.allclose(a[f"gamma{m}"], np.array(s.gamma_doc_statistics(m, 1.0, 5, 0)), rtol=1e-12, atol=0)
    s.close()


def _hyper_after(hy, o):
    """the hyper-parameters as the sweeps left them (an activated topic took alpha[m][K], UPD:263-270)"""
    from mvtopicmodel_amd.native import Hyper
    return Hyper(alpha=o.get_alpha(), alpha_sum=hy.alpha_sum, beta=hy.beta, beta_sum=hy.beta_sum, gamma=hy.gamma, p_a=hy.p_a, p_b=hy.p_b,
                 inactive=o.get_inactive() if hy.inactive is not None else None)


@pytest.mark.parametrize("nranks", [2, 4])
def test_segmented_sweep_across_ranks_equals_the_same_shards_in_one_process(tmp_path, fake_rccl, nranks):
    procs, outs, logs, arrs = run_ranks(tmp_path, fake_rccl, nranks, "segmented")
    assert_all_ok(procs, outs, logs)
    c, z0 = W.corpus()
    hy = W.hyper("segmented")
    tot = sum(np.diff(c.doc_off[m]) for m in range(c.M))
    shards = []
    for lo, hi in synth.shard_bounds(tot, nranks):
        sub = c.slice_docs(lo, hi)
        shards.append(make_native(sub, hy, [z0[m][c.doc_off[m][lo]:c.doc_off[m][hi]] for m in range(c.M)], doc_id_base=lo))
    with NativeGroup(shards) as g:
        g.build_counts()
        for it in range(2):
            g.sweep(it, W.SEED, flags=SWEEP_SEGMENT_APPLY | SWEEP_LIVE_SEGMENTS(3))
        for r, a in enumerate(arrs):
            for m in range(c.M):
                assert np.array_equal(a[f"end_z{m}"], shards[r].get_assignments(m)), f"rank {r}: assignments of view {m}"
                nwk, nk = shards[r].get_counts(m)
                assert np.array_equal(a[f"end_nwk{m}"], nwk) and np.array_equal(a[f"end_nk{m}"], nk)
    for s in shards:
        s.close()
    assert_replicas_hold(arrs, "end", c)


@pytest.mark.parametrize("nranks,scenario", [(2, "live"), (4, "live"), (2, "async"), (4, "async")])
def test_live_sweeps_across_ranks_keep_every_replica_the_count_of_the_assignments(tmp_path, fake_rccl, nranks, scenario):
    """MVHDP_SWEEP_LIVE over ranks (AD-LDA across replicas), also with the exchange one sweep behind (ASYNC_EXCHANGE) and a drain."""
    procs, outs, logs, arrs = run_ranks(tmp_path, fake_rccl, nranks, scenario)
    assert_all_ok(procs, outs, logs)
    c, _ = W.corpus()
    for lg in logs:
        assert [e["ok"] for e in lg["events"]] == [True] * 3 and all(e["changed"] > 0 for e in lg["events"])
    assert sum(lg["events"][0]["tokens"] for lg in logs) == c.total_tokens
    assert_replicas_hold(arrs, "end", c)
    if scenario == "async":
        for a in arrs[1:]:
            assert np.array_equal(a["ll"], arrs[0]["ll"])


@pytest.mark.parametrize("nranks,scenario", [(2, "fail_sweep"), (4, "fail_sweep"), (2, "abort"), (4, "abort")])
def test_a_failing_rank_fails_the_sweep_on_every_rank_and_a_recount_recovers(tmp_path, fake_rccl, nranks, scenario):
    """... and the failed rank's delta buffer, which the in-place collectives filled with the peers' sums, does not leak into the next
    sweep: after the recount two deferred sweeps equal the oracle's from the recovered assignments."""
    procs, outs, logs, arrs = run_ranks(tmp_path, fake_rccl, nranks, scenario)
    assert_all_ok(procs, outs, logs)
    c, _ = W.corpus()
    hy = W.hyper(scenario)
    for r, lg in enumerate(logs):
        ev = {e["sweep"]: e for e in lg["events"]}
        assert ev[0]["ok"] and not ev[1]["ok"] and lg["failed_together"], f"rank {r}: {lg}"
        assert lg["recount_ok"] and ev[2]["ok"] and ev[3]["ok"], f"rank {r}: {lg}"
    assert_replicas_hold(arrs, "recovered", c)
    z = concat_z(arrs, "recovered", c.M)
    o = make_oracle_from(c, hy, z)
    o.sweep(2, W.SEED); o.sweep(3, W.SEED)
    ze = concat_z(arrs, "end", c.M)
    for m in range(c.M):
        assert np.array_equal(ze[m], o.get_assignments(m)), f"assignments of view {m} after the recovery differ from the oracle's"
        nwk, nk = o.get_counts(m)
        for r, a in enumerate(arrs):
            assert np.array_equal(a[f"end_nwk{m}"], nwk) and np.array_equal(a[f"end_nk{m}"], nk), f"rank {r}: counts of view {m} after the recovery"


def test_a_rank_that_dies_gives_its_peers_an_error_not_a_hang(tmp_path, fake_rccl):
    procs, outs, logs, arrs = run_ranks(tmp_path, fake_rccl, 3, "die", timeout=120, collective_timeout_ms=3000)
    for r in (0, 1):
        assert procs[r].returncode == 0 and logs[r] is not None, outs[r][-3000:]
        assert logs[r]["events"][0]["ok"] and logs[r]["peer_error"], logs[r]
        assert logs[r]["seconds"] < 60


def _bench_line(args, env, timeout=420):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                 # (rank 0 prints ONE JSON line)
    return json.loads(lines[0])


@pytest.mark.parametrize("nranks", [2, 4])
def test_bench_with_several_ranks_takes_the_native_group_and_ends_at_the_single_gpu_counts(fake_rccl, nranks):
    """`python bench.py --gpus N` (N = 2, 4) as the driver launches it (torch.distributed.run, one process per rank), all ranks on cuda:0 with the
    test-side collective: the N > 1 leg of the bench -- the id broadcast, mvhdp_group_create_rank, the exchange inside the library -- runs
    for real instead of ending in the fallback, and since deferred sweeps do not depend on the sharding the final counts are those of
    the one-GPU run of the same command."""
    common = ["--workload", "C3", "--docs", "24000", "--steps", "4", "--warmup", "3", "--no-cpu-baseline", "--live-steps", "0"]
    env = dict(os.environ)
    one = _bench_line(["--gpus", "1"] + common, env)
    env2 = dict(env, MVHDP_RCCL_LIB=fake_rccl, MVHDP_RCCL_LIB_FIRST="1", FAKE_RCCL_TIMEOUT_MS="60000")   # (FIRST: torch has mapped the real RCCL in these processes)
    two = _bench_line(["--gpus", str(nranks), "--rehearse-native"] + common, env2)
    assert two["n_gpus"] == nranks and two["exchange"]["kind"].startswith("native"), two["exchange"]
    assert "fallback_reason" not in two["exchange"]
    assert two["final_nk_fingerprint"] == one["final_nk_fingerprint"]
    assert two["config"]["tokens"] == one["config"]["tokens"] and two["value"] > 0
