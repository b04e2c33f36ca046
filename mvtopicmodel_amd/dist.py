"""Document shards across GPUs, driven from Python through torch.distributed.

SINCE ROUND 3 THE PRODUCT'S MULTI-GPU PATH IS mvhdp_group_* INSIDE libmvhdp.so (csrc/mvhdp_group.hip; Python view:
mvtopicmodel_amd.native.NativeGroup): the same sequence -- sweep with NO_APPLY, sum of the deltas, apply by row ranges --
with RCCL called by the library itself.  This module stays for what torch.distributed can do and the library cannot:
the world_size-2 `gloo` tests on CPU (tests/test_dist_gloo.py, the shard backed by the test oracle), the host-staged
rehearsal of N ranks on one GPU, and bench.py's fallback when the native group cannot be formed.

One process per GPU, each holding a contiguous entity range and a full replica of n_wk / n_k.  The only
exchange step of the path is one integer all-reduce of the sweep's count deltas
(torch.distributed's "nccl" backend = RCCL over xGMI; "gloo" on CPU in the tests).

The reference has no counterpart (single JVM, shared arrays PTM:84-87); the
sharding follows its worker slices PTM:1051-1098: documents are the independent
unit, the model is shared.

Stream discipline: the library's kernels and the collective share ONE HIP stream
(a torch.cuda.Stream handed to mvhdp_set_stream), so sweep -> all-reduce ->
apply are ordered on the device and the host never waits between them (the
sweep's own statistics read-back is the only synchronisation per sweep).
"""
import time

import torch
import torch.distributed as dist

from .native import BUF_COUNTS, BUF_DELTA, SWEEP_NO_APPLY

KEY_NONE = (1 << 63) - 1   # MVHDP_ACT_KEY_NONE: "no activation"
# include/mvhdp.h MVHDP_ACT_*: doc << 34 | view << 31 | position << 11 | topic (pinned by tests/test_abi.py)
ACT_DOC_SHIFT, ACT_VIEW_SHIFT, ACT_POS_SHIFT = 34, 31, 11
ACT_TOPIC_MASK, ACT_VIEW_MASK = 0x7FF, 0x7


class _DevArray:
    """Minimal __cuda_array_interface__ exporter for a raw device pointer."""

    def __init__(self, ptr, n_int32):
        self.__cuda_array_interface__ = {
            "shape": (int(n_int32),), "typestr": "<i4", "data": (int(ptr), False), "version": 3, "strides": None,
        }


def device_int32_tensor(ptr, nbytes, device):
    """torch.int32 tensor aliasing library-owned HBM (no copy)."""
    t = torch.as_tensor(_DevArray(ptr, nbytes // 4), device=device)
    assert t.data_ptr() == ptr, "torch copied the buffer instead of aliasing it"
    return t


class GpuShard:
    """One NativeSampler (one GPU) as a shard of the global model."""

    def __init__(self, sampler, device, host_staged=False, share_stream=True):
        self.s = sampler
        self.device = torch.device(device)
        p, n = sampler.device_buffer(BUF_COUNTS)
        self._counts_dev = device_int32_tensor(p, n, self.device)
        p, n = sampler.device_buffer(BUF_DELTA)
        self._delta_dev = device_int32_tensor(p, n, self.device)
        # host_staged: the collective runs on CPU copies (gloo rehearsal of the N>1 path on one GPU)
        self.host_staged = host_staged
        self._counts_host = self._delta_host = None
        # one stream for the library's kernels and the collectives
        self.stream = None
        if share_stream and not host_staged:
            self.stream = torch.cuda.Stream(device=self.device)
            sampler.set_stream(self.stream.cuda_stream)

    def close(self):
        if self.stream is not None and getattr(self.s, "h", None):
            self.s.set_stream(None)            # back to a stream the library owns before torch drops this one
        self.stream = None

    def on_stream(self):
        return torch.cuda.stream(self.stream) if self.stream is not None else _NullCtx()

    @property
    def counts(self):
        if not self.host_staged:
            return self._counts_dev
        self._counts_host = self._counts_dev.cpu()
        return self._counts_host

    @property
    def delta(self):
        if not self.host_staged:
            return self._delta_dev
        self._delta_host = self._delta_dev.cpu()
        return self._delta_host

    def has_inactive(self):
        """inActiveTopicIndex non-empty (PTM:95)?  Every replica holds the same hyper-parameters, so every rank
        answers alike."""
        return bool(self.s.get_alpha()[1].any())

    def build_counts_local(self):
        self.s.build_counts()

    def counts_written(self):
        self.s.counts_written()

    def sweep_local(self, sweep_idx, seed, flags=0):
        return self.s.sweep(sweep_idx, seed, flags=flags | SWEEP_NO_APPLY)

    def sweep_and_apply(self, sweep_idx, seed, flags=0):
        """A single shard needs no exchange step: the library applies its own deltas."""
        return self.s.sweep(sweep_idx, seed, flags=flags)

    def apply(self, topic, modality):
        self.s.apply_delta(topic, modality)

    # ---- pipelined exchange: chunked all-reduce overlapped with apply + F+tree rebuild of the chunk's rows ----
    def can_pipeline(self):
        return self.stream is not None or self.host_staged

    def row_chunks(self, n_chunks):
        """[(row_begin, row_end)] covering the n_wk rows of every view, and the element offset of the n_k part."""
        rows = sum(self.s.V)
        n_chunks = max(1, min(int(n_chunks), rows))
        cuts = [rows * i // n_chunks for i in range(n_chunks + 1)]
        return [(cuts[i], cuts[i + 1]) for i in range(n_chunks) if cuts[i + 1] > cuts[i]], rows * self.s.K

    def trees_current(self):
        return self.s.trees_current()

    def sync(self):
        if self.host_staged:                 # push all-reduced host copies back to the library's buffers
            if self._counts_host is not None:
                self._counts_dev.copy_(self._counts_host); self._counts_host = None
            if self._delta_host is not None:
                self._delta_dev.copy_(self._delta_host); self._delta_host = None
            torch.cuda.synchronize(self.device)
        elif self.stream is None:
            torch.cuda.synchronize(self.device)
        # shared stream: ordered on the device, nothing to wait for


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def decode_activation(key):
    """(topic, modality) of an activation key (include/mvhdp.h MVHDP_ACT_KEY)."""
    if key == KEY_NONE:
        return -1, -1
    return int(key & ACT_TOPIC_MASK), int((key >> ACT_VIEW_SHIFT) & ACT_VIEW_MASK)


def build_counts_all_reduce(shard, group=None):
    """buildInitialTypeTopicCounts PTM:600-652 over every shard: local counts, then a sum all-reduce."""
    shard.build_counts_local()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        shard.sync()
        with shard.on_stream():
            t = shard.counts
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        shard.sync()
        shard.counts_written()


PIPELINE_CHUNKS = 4      # all-reduce chunks per sweep: the apply + tree rebuild of chunk i runs while chunk i+1 is on the wire


def sweep_all_reduce(shard, sweep_idx, seed, group=None, flags=0, timings=None, pipeline=True, force_exchange=False):
    """One global Gibbs sweep: every shard samples its entities against the same snapshot of
    n_wk / n_k, the deltas are summed across shards, every replica applies the same sum
    (AD-LDA style).  Bit-identical to the single-shard sweep: entities are independent under
    the snapshot semantics and integer sums do not depend on the order.  (With MVHDP_SWEEP_LIVE
    in `flags` a shard is live for its own entities and one sweep stale for the others'.)

    When inActiveTopicIndex is non-empty the first activating delta in (entity, view, position)
    order must win on every replica alike (UPD:263-270): the 8-byte activation key is
    MIN-all-reduced -- decided from the replicated hyper-parameters, never by the caller.

    timings: optional dict, accumulates milliseconds per phase: "sweep_call" (host wall time of
    mvhdp_sweep: view weights + trees + kernels + statistics read-back), "sweep_kernel" (device),
    "allreduce" (host wall time until the collective is enqueued and, with a shared stream, device
    time by events), "apply" (host wall time of mvhdp_apply_delta, which waits for the collective)."""
    t0 = time.perf_counter()
    multi = dist.is_initialized() and (dist.get_world_size(group) > 1 or force_exchange)
    if multi and pipeline and getattr(shard, "can_pipeline", lambda: False)():
        return _sweep_pipelined(shard, sweep_idx, seed, group, flags, timings, t0)
    if not multi and hasattr(shard, "sweep_and_apply"):
        st = shard.sweep_and_apply(sweep_idx, seed, flags)
        if timings is not None:
            timings["sweep_call"] = timings.get("sweep_call", 0.0) + (time.perf_counter() - t0) * 1e3
            timings["sweep_kernel"] = timings.get("sweep_kernel", 0.0) + st.sweep_kernel_ms
            timings["sweep_device_total"] = timings.get("sweep_device_total", 0.0) + st.total_ms
            timings["n"] = timings.get("n", 0) + 1
        return st
    st = shard.sweep_local(sweep_idx, seed, flags)
    t1 = time.perf_counter()
    topic, modality = st.activated_topic, st.activated_modality
    ev = None
    if multi:
        need_key = shard.has_inactive()
        shard.sync()
        with shard.on_stream():
            if timings is not None and shard.stream is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            t = shard.delta
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            if need_key:
                key = torch.tensor([st.activation_key], dtype=torch.int64, device=t.device)
                dist.all_reduce(key, op=dist.ReduceOp.MIN, group=group)
            if ev is not None:
                ev[1].record()
            if need_key:
                topic, modality = decode_activation(int(key.item()))
        shard.sync()
    t2 = time.perf_counter()
    shard.apply(topic, modality)
    t3 = time.perf_counter()
    if timings is not None:
        timings["sweep_call"] = timings.get("sweep_call", 0.0) + (t1 - t0) * 1e3
        timings["sweep_kernel"] = timings.get("sweep_kernel", 0.0) + st.sweep_kernel_ms
        timings["sweep_device_total"] = timings.get("sweep_device_total", 0.0) + st.total_ms
        timings["allreduce_host"] = timings.get("allreduce_host", 0.0) + (t2 - t1) * 1e3
        timings["apply"] = timings.get("apply", 0.0) + (t3 - t2) * 1e3
        if ev is not None:
            timings["allreduce_device"] = timings.get("allreduce_device", 0.0) + ev[0].elapsed_time(ev[1])
        timings["n"] = timings.get("n", 0) + 1
    return st


def _sweep_pipelined(shard, sweep_idx, seed, group, flags, timings, t0):
    """The exchange step as a pipeline on the shard's stream (GPU shards only): the tokensPerTopic part and then
    PIPELINE_CHUNKS row ranges of the delta buffer are all-reduced one after the other on RCCL's stream; as each chunk
    arrives its rows are applied and their F+trees rebuilt (mvhdp_apply_delta_rows) while the next chunk is still on the
    wire.  The next sweep then reuses the trees (MVHDP_SWEEP_REUSE_TREES) instead of rebuilding them.  Same integers as
    the plain sequence: the trees are built from the same counts by the same kernel."""
    from .native import SWEEP_LIVE, SWEEP_REUSE_TREES
    # (a live sweep rebuilds its trees per segment by itself; REUSE_TREES would mean "never" there)
    reuse = SWEEP_REUSE_TREES if (shard.trees_current() and not (flags & SWEEP_LIVE)) else 0
    st = shard.sweep_local(sweep_idx, seed, flags | reuse)
    t1 = time.perf_counter()
    need_key = shard.has_inactive()
    chunks, nk_off = shard.row_chunks(PIPELINE_CHUNKS)
    K = shard.s.K
    ev = None
    if shard.host_staged:
        # gloo rehearsal on one GPU: the same sequence of collectives and row-range updates, each chunk staged through
        # the host (no overlap to measure; what it exercises is that every rank issues the same chunks in the same order)
        torch.cuda.synchronize(shard.device)
        t = shard._delta_dev
        pieces = [(nk_off, t.numel())] + [(r0 * K, r1 * K) for r0, r1 in chunks]
        for i, (a, b) in enumerate(pieces):
            h = t[a:b].cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            t[a:b].copy_(h)
            torch.cuda.synchronize(shard.device)
            if i == 0:
                shard.s.apply_delta_begin()
            else:
                shard.s.apply_delta_rows(*chunks[i - 1])
        topic, modality = st.activated_topic, st.activated_modality
        if need_key:
            key = torch.tensor([st.activation_key], dtype=torch.int64)
            dist.all_reduce(key, op=dist.ReduceOp.MIN, group=group)
            topic, modality = decode_activation(int(key.item()))
        shard.s.apply_delta_end(topic, modality)
        if timings is not None:
            timings["sweep_call"] = timings.get("sweep_call", 0.0) + (t1 - t0) * 1e3
            timings["sweep_kernel"] = timings.get("sweep_kernel", 0.0) + st.sweep_kernel_ms
            timings["n"] = timings.get("n", 0) + 1
        return st
    with shard.on_stream():
        if timings is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        t = shard.delta
        works = [dist.all_reduce(t[nk_off:], op=dist.ReduceOp.SUM, group=group, async_op=True)]
        for r0, r1 in chunks:
            works.append(dist.all_reduce(t[r0 * K:r1 * K], op=dist.ReduceOp.SUM, group=group, async_op=True))
        key = None
        if need_key:
            key = torch.tensor([st.activation_key], dtype=torch.int64, device=t.device)
            works.append(dist.all_reduce(key, op=dist.ReduceOp.MIN, group=group, async_op=True))
        works[0].wait()                                   # the stream waits, the host does not
        shard.s.apply_delta_begin()
        for (r0, r1), w in zip(chunks, works[1:]):
            w.wait()
            shard.s.apply_delta_rows(r0, r1)
        topic, modality = st.activated_topic, st.activated_modality
        if need_key:
            works[-1].wait()
            topic, modality = decode_activation(int(key.item()))
        if ev is not None:
            ev[1].record()
    t2 = time.perf_counter()
    shard.s.apply_delta_end(topic, modality)
    t3 = time.perf_counter()
    if timings is not None:
        timings["sweep_call"] = timings.get("sweep_call", 0.0) + (t1 - t0) * 1e3
        timings["sweep_kernel"] = timings.get("sweep_kernel", 0.0) + st.sweep_kernel_ms
        timings["sweep_device_total"] = timings.get("sweep_device_total", 0.0) + st.total_ms
        timings["exchange_enqueue_host"] = timings.get("exchange_enqueue_host", 0.0) + (t2 - t1) * 1e3
        timings["exchange_wait_host"] = timings.get("exchange_wait_host", 0.0) + (t3 - t2) * 1e3
        timings["exchange_device"] = timings.get("exchange_device", 0.0) + ev[0].elapsed_time(ev[1])
        timings["n"] = timings.get("n", 0) + 1
    return st
