"""The sweep's planner and its walk-threshold search (mvtopicmodel_amd/csrc/mvhdp_plan.h) on recorded inputs, without a GPU:
mvhdp_plan_probe / mvhdp_tuner_probe are pure functions of their arguments (include/mvhdp.h)."""
import ctypes as C

import numpy as np
import pytest

from mvtopicmodel_amd import _lib

REGS = [(70, 72, 96), (72, 72, 104), (125, 128, 160), (207, 226, 256), (256, 256, 256), (96, 96, 128)]   # VGPRs as hipcc allocates them
SWEEP_REUSE_TREES, SWEEP_NO_APPLY, SWEEP_FROZEN, SWEEP_LIVE, SWEEP_SEGMENT_APPLY = 0x1, 0x2, 0x10, 0x20, 0x40


def probe(K=400, M=3, D=1_000_000, mdt=250, longer=(1_000_000, 900_000, 40, 0, 0), tok=None, ent=None, flags=0, tuning=None,
          batch=0, debug=0, trees_current=0, inactive=0):
    L = _lib.load_library()
    pi = _lib.PlanInputC()
    pi.num_topics, pi.num_modalities, pi.num_entities, pi.max_entity_tokens = K, M, D, mdt
    for i, v in enumerate(longer):
        pi.entities_longer_than[i] = v
    for i, v in enumerate(tok or []):
        pi.tokens_by_list_rounds[i] = v
    for i, v in enumerate(ent or []):
        pi.entities_by_class[i] = v
    pi.flags, pi.debug, pi.batch, pi.trees_current, pi.num_cus = flags, debug, batch, trees_current, 256
    pi.inactive_topics = inactive
    for c in range(6):
        for f in range(3):
            pi.kernel_registers[c][f] = REGS[c][f]
    po = _lib.PlanOutputC()
    t = None
    if tuning:
        t = _lib.TuningC()
        t.narrow = -1
        t.live16 = -1
        t.live_rows = -1
        t.live_overlap = -1
        for g in range(4):
            t.learnt_walk_step[g] = -1
        for k, v in tuning.items():
            if isinstance(v, (list, tuple)):
                for i, x in enumerate(v):
                    getattr(t, k)[i] = x
            else:
                setattr(t, k, v)
    assert L.mvhdp_plan_probe(C.byref(pi), C.byref(t) if t is not None else None, C.byref(po)) == 0
    return po


def test_c4_early_chain_two_round_primary_and_a_four_round_class_beside_it():
    # sweep 5 of a C4 chain from a random start: 4 % of the tokens in lists of <= 64 topics, 95 % in 65..128, 1 % beyond
    po = probe(tok=[6_000_000, 140_000_000, 1_000_000, 200_000], ent=[40_000, 950_000, 10_000, 0, 0, 0, 0, 0])
    assert po.status == 0 and po.register_resident == 1
    assert po.primary_class == 1 and po.dominant_class == 1                 # 4 % is not worth a kernel of its own (primary_min_share 0.10)
    assert list(po.class_used) == [0, 1, 1, 0, 0, 0]
    assert list(po.class_map)[:3] == [1, 1, 2]
    assert po.routed_prefix == 900_000                                      # entities with more than 128 tokens
    assert po.class_stream[1] == 0 and po.class_stream[2] == 3              # primary on the handle's stream, the wider class beside it
    assert po.class_narrow[1] == po.class_walk[1]                           # the mirror goes with the walk flavour, whatever the variant
    assert po.class_grid[1] == 256 * 7                                      # 72 VGPRs: 7 waves per SIMD = 7 blocks of 4 waves per CU


def test_c4_settling_chain_one_round_primary_with_the_mirror():
    po = probe(tok=[80_000_000, 66_000_000, 1_000_000], ent=[550_000, 449_000, 1_000, 0, 0, 0, 0, 0],
               tuning=dict(walk_fixed=1, walk_theta=[0.5, 0.0, 0.0]))
    assert po.primary_class == 0 and po.dominant_class == 0
    assert list(po.class_used) == [1, 1, 1, 0, 0, 0]
    assert po.routed_prefix == 1_000_000                                    # every entity has more than 64 tokens
    assert po.class_walk[0] == 1 and po.class_narrow[0] == 1 and po.class_theta0[0] == 0.5
    assert po.class_narrow[1] == po.class_walk[1] and po.class_narrow[2] == po.class_walk[2]
    p1 = probe(tok=[80_000_000, 66_000_000, 1_000_000], ent=[550_000, 449_000, 1_000, 0, 0, 0, 0, 0],
               tuning=dict(walk_fixed=1, walk_theta=[0.5, 0.0, 0.0], narrow=1))
    assert p1.class_narrow[0] == 1 and p1.class_narrow[1] == 0              # narrow = 1: the mirror for the 1-round variant only
    assert po.class_stream[0] == 0 and po.class_stream[1] == 3 and po.class_stream[2] == 3      # (0.7 % of the tokens in the 4-round class: behind the 2-round one)
    assert po.class_grid[0] == 256 * 7                                      # 72 VGPRs: 7 waves per SIMD
    # a live sweep cannot use the snapshot mirror; nor a sweep that re-uses trees which are not current
    po = probe(tok=[80_000_000, 66_000_000], ent=[550_000, 450_000], flags=SWEEP_LIVE, tuning=dict(walk_fixed=1, walk_theta=[0.5], live16=0))
    assert po.class_narrow[0] == 0
    # ... unless the live sweep keeps the mirror current itself (live16): then EVERY kernel of the sweep reads and updates the mirror
    po = probe(tok=[80_000_000, 66_000_000], ent=[550_000, 450_000], flags=SWEEP_LIVE)
    assert [po.class_narrow[c] for c in range(2)] == [1, 1] and [po.class_walk[c] for c in range(2)] == [1, 1]
    assert probe(K=100, tok=[80, 66], ent=[55, 45], flags=SWEEP_LIVE).class_narrow[0] == 0           # short rows: not worth it by default
    assert probe(tok=[80_000_000, 66_000_000], ent=[550_000, 450_000], flags=SWEEP_REUSE_TREES, tuning=dict(walk_fixed=1, walk_theta=[0.5])).class_narrow[0] == 0
    assert probe(tok=[80_000_000, 66_000_000], ent=[550_000, 450_000], flags=SWEEP_REUSE_TREES, trees_current=1, tuning=dict(walk_fixed=1, walk_theta=[0.5])).class_narrow[0] == 1
    assert probe(tok=[80_000_000, 66_000_000], ent=[550_000, 450_000], tuning=dict(walk_fixed=1, walk_theta=[0.5], narrow=0)).class_narrow[0] == 0


def test_every_kernel_flavour_has_its_own_walk_threshold():
    """The 1-round kernel on the 16-bit mirror, the 1-round kernel on 32-bit rows and the wider variants are bound by different
    things: each starts from its own measured threshold and keeps what the search finds for it."""
    known = dict(learnt_walk_step=[-1, -1, 11, -1], tree_branch_share=[0.2, 0.6, 0.6])       # view 0 is steered, the side views are always walked
    kw = dict(tok=[80_000_000, 66_000_000], ent=[550_000, 450_000])
    po = probe(tuning=known, **kw)
    assert po.class_narrow[0] == 1 and abs(po.class_theta0[0] - 0.30) < 1e-12                # default of the mirror flavour
    assert abs(po.class_theta0[1] - 0.55) < 1e-12                                            # what the search had found for the wider variants
    po = probe(tuning=dict(narrow=0, **known), **kw)
    assert po.class_narrow[0] == 0 and po.class_theta0[0] == 1.0                             # 32-bit rows: every walk on demand
    po = probe(K=200, tuning=dict(learnt_walk_step=[-1, -1, 0, -1], tree_branch_share=[0.2, 0.6, 0.6]), **kw)
    assert po.class_theta0[0] == 0.0 and po.class_walk[0] == 0                               # short rows: latency-bound, plain flavour


def test_long_rows_take_the_mirror_even_when_no_view_is_steered():
    """C5: K = 1000, the tree branch takes 44 % of the tokens, so no view has a walk threshold -- but a row of the counts is 4 KB, and
    the 16-bit mirror (the walk flavour's) halves the lines of every gather: 51.9 against 64.4 ms per sweep."""
    known = dict(tree_branch_share=[0.44, 0.45, 0.45, 0.45, 0.45])
    kw = dict(K=1000, M=5, tok=[30_000_000, 40_000_000, 20_000_000, 20_000_000], ent=[400_000, 400_000, 150_000, 50_000, 0, 0, 0, 0])
    po = probe(tuning=known, **kw)
    used = [c for c in range(5) if po.class_used[c]]
    assert used and all(po.class_walk[c] == 1 and po.class_narrow[c] == 1 and po.class_theta0[c] == 0.0 for c in used)
    po = probe(tuning=dict(narrow=0, **known), **kw)
    assert all(po.class_narrow[c] == 0 for c in used)


def test_empty_classes_are_not_launched_unless_the_sizes_can_move():
    kw = dict(tok=[146_000_000, 1_000_000], ent=[990_000, 10_000, 0, 0, 0, 0, 0, 0])
    po = probe(**kw)
    assert list(po.class_used) == [1, 1, 0, 0, 0, 0] and list(po.class_map)[:3] == [0, 1, -1]
    po = probe(batch=1, **kw)                                               # mvhdp_sweep_many: lists grow between the sweeps of a batch
    assert list(po.class_used) == [1, 1, 1, 0, 0, 0] and list(po.class_map)[:3] == [0, 1, 2]   # 250 tokens: at most 4 rounds
    po = probe(tok=[146_000_000, 1_000_000], ent=[990_000, 9_999, 0, 0, 0, 0, 1, 0])            # one entity of unknown size
    assert list(po.class_used) == [1, 1, 1, 0, 0, 0]


def test_c5_power_law_every_class_on_its_stream():
    po = probe(K=1000, M=5, D=1_000_000, mdt=2080, longer=(600_000, 200_000, 60_000, 9_000, 2_000),
               tok=[50_000_000, 30_000_000, 10_000_000, 6_000_000, 2_000_000, 2_000_000, 1_000_000, 1_000_000] + [1_000_000] * 8,
               ent=[800_000, 150_000, 40_000, 8_000, 1_500, 0, 0, 0])
    assert po.primary_class == 0
    assert list(po.class_used) == [1, 1, 1, 1, 1, 0]
    assert [po.class_stream[c] for c in range(5)] == [0, 3, 4, 2, 1]               # a stream per class (2 = B, 1 = A: created at high priority -- a hardware-queue pool of their own;
    #                                  the 4-round class carries 14 % of the tokens here: its own stream D)
    assert [probe(K=1000, M=5, mdt=2080, longer=(600_000, 200_000, 60_000, 9_000, 2_000), tok=[50, 30, 10, 6, 2, 2, 1, 1] + [1] * 8,
                  ent=[800, 150, 40, 8, 2, 0, 0, 0], tuning=dict(single_stream=1)).class_stream[c] for c in range(5)] == [0] * 5
    assert po.need_full_trees == 0
    # K = 2048 with eight views: the 16-round variant's slot state does not fit the LDS next to nothing: generic kernel for that class
    po = probe(K=2048, M=8, D=1000, mdt=5000, longer=(1000, 900, 800, 500, 100), tok=[1, 1, 1, 1] + [100] * 13, ent=[1, 1, 1, 1, 1, 995, 0, 0])
    assert po.status == 0 and po.class_used[5] == 1 and po.need_full_trees == 1


def test_forced_variants_flags_and_errors():
    assert probe(mdt=600, tok=[1, 100], ent=[1, 100], tuning=dict(force_primary=8)).primary_class == 3
    assert probe(mdt=250, tok=[1, 100], ent=[1, 100], tuning=dict(force_primary=8)).primary_class == 2     # no wider than the longest entity can need
    po = probe(tok=[1, 100], ent=[1, 100], tuning=dict(force_primary=32))
    assert po.register_resident == 0 and list(po.class_used) == [0, 0, 0, 0, 0, 1] and po.need_full_trees == 1
    assert probe(tok=[1, 100], ent=[1, 100], flags=0x8).register_resident == 0                     # MVHDP_SWEEP_GENERIC_KERNEL
    assert probe(flags=SWEEP_SEGMENT_APPLY | (8 << 16)).segments == 8 and probe().segments == 1
    # a live sweep: its live-rows form (the tree branch reads the word's live count row: one segment is enough) wherever every kernel is
    # register-resident; stored trees rebuilt at four segment borders otherwise
    po = probe(flags=SWEEP_LIVE)
    assert (po.segments, po.live_rows) == (1, 1) and all(po.class_walk[c] for c in range(6) if po.class_used[c])
    po = probe(flags=SWEEP_LIVE, tuning=dict(live_rows=0))
    assert (po.segments, po.live_rows) == (4, 0)
    assert probe(flags=SWEEP_LIVE | (3 << 16)).segments == 3 and probe(flags=SWEEP_LIVE | (3 << 16)).live_rows == 1
    assert probe(flags=SWEEP_LIVE, debug=1).live_rows == 0 and probe(flags=SWEEP_LIVE | SWEEP_REUSE_TREES, trees_current=1).live_rows == 0
    assert probe(K=2048, M=2, D=1000, mdt=5000, longer=(1000, 900, 800, 500, 100), flags=SWEEP_LIVE).live_rows == 0      # (the generic kernel can be reached)
    # ... also over a truncated HDP (inActiveTopicIndex not empty): the live-rows form gives birth to topics chunk by chunk, the stored-tree
    # form one per segment border
    assert probe(flags=SWEEP_LIVE, inactive=1).segments == 1 and probe(flags=SWEEP_LIVE, inactive=1).live_rows == 1
    assert probe(flags=SWEEP_LIVE, inactive=1, tuning=dict(live_rows=0)).segments == 4
    assert probe(D=3, longer=(3, 3, 0, 0, 0), flags=SWEEP_LIVE | (200 << 16)).segments == 3        # never more segments than entities
    assert probe(flags=SWEEP_SEGMENT_APPLY | SWEEP_NO_APPLY).status == -1
    assert probe(flags=SWEEP_LIVE | SWEEP_FROZEN).status == -1
    assert probe(flags=0x8000).status == -1
    # a view beyond 65535 tokens cannot sit in the 16-bit slot counts of the 8- and 16-round variants: generic class, always launched
    po = probe(K=1000, M=1, D=10, mdt=70_000, longer=(10, 10, 10, 10, 1), tok=[100], ent=[9, 0, 0, 0, 0, 1, 0, 0], tuning=dict(force_primary=8))
    assert po.class_used[5] == 1 and po.routed_prefix >= 1


def test_walk_threshold_search_finds_the_minimum_of_a_smooth_cost_curve():
    L = _lib.load_library()
    f = np.array([0.2, 0.6, 0.5], dtype=np.float64)                        # only view 0 is steered (tree-branch share < 0.35)
    u1 = np.full(20, 0.001, dtype=np.float64)
    steps = np.zeros(120, dtype=np.int32)
    for best in (0, 9, 20):
        ns = np.array([1.0 + 0.004 * abs(i - best) for i in range(21)], dtype=np.float64)
        assert L.mvhdp_tuner_probe(3, f.ctypes.data, u1.ctypes.data, ns.ctypes.data, len(steps), 0, steps.ctypes.data) == 0
        assert abs(int(np.median(steps[-30:])) - best) <= 2, (best, steps)
    # a flat curve: the search must not wander off
    ns = np.ones(21)
    assert L.mvhdp_tuner_probe(3, f.ctypes.data, u1.ctypes.data, ns.ctypes.data, len(steps), 1, steps.ctypes.data) == 0
    assert steps.max() <= 8


def test_sixteen_bit_delta_cells_are_for_plain_deferred_sweeps():
    """mvhdp_plan_output.delta16: the n_wk deltas of rows that cannot overflow 16 bits go to a table half the size -- when this call
    applies them itself and in one piece; document shards (NO_APPLY), segments, live and frozen sweeps keep the 32-bit table."""
    kw = dict(tok=[80_000_000, 66_000_000, 1_000_000], ent=[550_000, 449_000, 1_000, 0, 0, 0, 0, 0])
    assert probe(**kw).delta16 == 1
    assert probe(flags=SWEEP_NO_APPLY, **kw).delta16 == 0
    assert probe(flags=SWEEP_LIVE, **kw).delta16 == 0
    seg = probe(flags=SWEEP_SEGMENT_APPLY | (8 << 16), **kw)
    assert seg.status == 0 and seg.segments == 8 and seg.delta16 == 0                    # (tried: the fold at every segment border costs what the cells save)
    assert probe(flags=SWEEP_FROZEN | SWEEP_REUSE_TREES, trees_current=1, **kw).delta16 == 0
    assert probe(debug=1, **kw).delta16 == 0                                             # the debug flavour does not read the mirror
    assert probe(tuning=dict(narrow=0), **kw).delta16 == 0                               # nor a sweep pinned to the 32-bit rows
