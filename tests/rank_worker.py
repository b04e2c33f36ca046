"""One rank of a multi-process run of the library's one-process-per-GPU path (mvhdp_group_create_rank with nranks > 1), for
tests/test_gpu_group_ranks.py: a FRESH process per rank, all on cuda:0, the collective being tests/native/fake_rccl.c
(MVHDP_RCCL_LIB: the real RCCL refuses two ranks on one device).  Test infrastructure.

  python tests/rank_worker.py <workdir> <rank> <nranks> <scenario>

Every rank builds the same synthetic corpus, keeps its document shard (balanced by token count), forms the group from the id rank 0
leaves in <workdir>/uid, runs the scenario and leaves <workdir>/rank<r>.npz (arrays) + rank<r>.json (what happened).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

K, V, D, LAM, CSEED = 48, [700, 90, 70], 260, [40, 5, 6], 97
SEED = 11


def corpus():
    from mvtopicmodel_amd import synth
    from mvtopicmodel_amd.java_init import init_assignments
    c = synth.generate(K, V, D, LAM, CSEED, chunk_docs=4096)
    z = init_assignments(K, c.doc_off, seed=1)
    return c, z


def hyper(scenario):
    from mvtopicmodel_amd.native import Hyper
    if scenario.startswith("inactive"):
        inactive = np.zeros(K, dtype=np.uint8); inactive[[41, 45]] = 1
        hy = Hyper.defaults(K, V, inactive=inactive); hy.alpha[:, K] = 25.0
        return hy
    return Hyper.defaults(K, V)


def main():
    workdir, rank, nranks, scenario = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    from mvtopicmodel_amd import NativeGroup, NativeSampler, synth
    from mvtopicmodel_amd._lib import MvhdpError
    from mvtopicmodel_amd.native import SWEEP_ASYNC_EXCHANGE, SWEEP_LIVE, SWEEP_LIVE_SEGMENTS, SWEEP_SEGMENT_APPLY
    c, z = corpus()
    hy = hyper(scenario)
    if scenario.startswith("inactive"):
        for m in range(c.M):
            z[m][np.isin(z[m], [41, 45])] = 1
    tot = sum(np.diff(c.doc_off[m]) for m in range(c.M))
    lo, hi = synth.shard_bounds(tot, nranks)[rank]
    sub = c.slice_docs(lo, hi)
    s = NativeSampler(K, V, device=0, doc_id_base=lo)
    for m in range(c.M):
        s.set_corpus(m, sub.doc_off[m], sub.tokens[m])
        s.set_assignments(m, z[m][c.doc_off[m][lo]:c.doc_off[m][hi]])
    s.set_hyper(hy)
    s.build_counts()
    uid_path = os.path.join(workdir, "uid")
    if rank == 0:
        uid = NativeGroup.unique_id()
        with open(uid_path + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(uid_path + ".tmp", uid_path)
    else:
        t0 = time.time()
        while not os.path.exists(uid_path):
            if time.time() - t0 > 120:
                raise SystemExit("no id from rank 0")
            time.sleep(0.02)
        uid = open(uid_path, "rb").read()
    g = NativeGroup.from_rank(s, uid, rank, nranks)
    info = g.info()
    log = {"rank": rank, "ranks": int(info.ranks), "rccl": int(info.rccl), "rccl_version": int(info.rccl_version), "events": []}
    out = {}

    def snap(tag):
        for m in range(c.M):
            out[f"{tag}_z{m}"] = s.get_assignments(m)
            nwk, nk = s.get_counts(m)
            out[f"{tag}_nwk{m}"] = nwk; out[f"{tag}_nk{m}"] = nk

    def sweep(idx, flags=0):
        try:
            st = g.sweep(idx, SEED, flags)[0]
            gi = g.info()
            log["events"].append({"sweep": idx, "ok": True, "tokens": int(st.tokens), "changed": int(st.changed),
                                  "activated_topic": int(st.activated_topic), "activated_modality": int(st.activated_modality),
                                  "packed": int(gi.exchange_packed), "bytes": int(gi.last_exchange_bytes)})
            return True
        except MvhdpError as e:
            log["events"].append({"sweep": idx, "ok": False, "code": e.code, "msg": str(e)})
            return False

    g.build_counts()
    snap("start")
    if scenario in ("deferred", "inactive_deferred"):
        for it in range(3):
            sweep(it)
        snap("end")
        a, ina = s.get_alpha()
        out["alpha"], out["inactive"] = a, ina
        # the statistics either side of the sweep, put together across the ranks
        out["ll"] = g.model_log_likelihood()
        maxlen = int(max(np.diff(c.doc_off[m]).max() for m in range(c.M))) + 1
        for m in range(c.M):
            hist, dl = g.get_doc_topic_hist(m, maxlen, maxlen)
            out[f"hist{m}"] = hist; out[f"doclen{m}"] = dl
            out[f"chist{m}"] = g.get_count_histogram(m, 64)
            out[f"gamma{m}"] = np.array(g.gamma_doc_statistics(m, 1.0, 5, 0))
        out["overlap"] = g.view_overlap_sums()
        log["exchange_ms"] = float(g.info().last_exchange_ms)
    elif scenario == "segmented":
        for it in range(2):
            sweep(it, SWEEP_SEGMENT_APPLY | SWEEP_LIVE_SEGMENTS(3))
        snap("end")
    elif scenario == "live":
        for it in range(3):
            sweep(it, SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(2))
        snap("end")
    elif scenario == "async":
        for it in range(3):
            sweep(it, SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(1) | SWEEP_ASYNC_EXCHANGE)
        g.drain()
        snap("end")
        out["ll"] = g.model_log_likelihood()
    elif scenario in ("fail_sweep", "abort"):
        sweep(0)
        bad = nranks - 1
        if rank == bad:
            if scenario == "abort":
                g.abort()
            else:
                s.set_assignments(0, s.get_assignments(0))        # the counts no longer vouch for the assignments: this rank's sweep is refused
        ok = sweep(1)                                             # must fail on EVERY rank, from the same call
        log["failed_together"] = not ok
        try:
            g.build_counts()                                      # the recount the header prescribes (collective)
            log["recount_ok"] = True
        except MvhdpError as e:
            log["recount_ok"] = False; log["recount_msg"] = str(e)
        snap("recovered")
        sweep(2)
        sweep(3)
        snap("end")
    elif scenario == "die":
        sweep(0)
        if rank == nranks - 1:
            os._exit(0)                                           # this rank is gone before the next sweep
        t0 = time.time()
        ok = sweep(1)                                             # the peers must get an error, not hang
        log["peer_error"] = not ok
        log["seconds"] = time.time() - t0
        try:
            g.build_counts()
            log["recount_ok"] = True
        except MvhdpError as e:
            log["recount_ok"] = False
    else:
        raise SystemExit("unknown scenario " + scenario)
    np.savez(os.path.join(workdir, f"rank{rank}.npz"), **out)
    with open(os.path.join(workdir, f"rank{rank}.json"), "w") as f:
        json.dump(log, f)
    if scenario != "die":
        g.close()
        s.close()
    else:
        os._exit(0)                                               # (a broken communicator: nothing to tear down in order)


if __name__ == "__main__":
    main()
