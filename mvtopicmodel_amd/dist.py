"""Document shards across GPUs: one process per GPU, each
holding a contiguous entity range and a full replica of n_wk / n_k.  The only
exchange step of the path is one integer all-reduce of the sweep's count deltas
(RCCL over xGMI through torch.distributed's "nccl" backend; "gloo" on CPU in
the tests, where the shard is backed by the test oracle instead of the GPU).

The reference has no counterpart (single JVM, shared arrays PTM:84-87); the
sharding follows its worker slices PTM:1051-1098: documents are the independent
unit, the model is shared.
"""
import numpy as np
import torch
import torch.distributed as dist

from .native import BUF_COUNTS, BUF_DELTA, SWEEP_NO_APPLY

KEY_NONE = (1 << 63) - 1   # LLONG_MAX: "no activation"


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

    def __init__(self, sampler, device, host_staged=False):
        self.s = sampler
        self.device = torch.device(device)
        p, n = sampler.device_buffer(BUF_COUNTS)
        self._counts_dev = device_int32_tensor(p, n, self.device)
        p, n = sampler.device_buffer(BUF_DELTA)
        self._delta_dev = device_int32_tensor(p, n, self.device)
        # host_staged: the collective runs on CPU copies (gloo rehearsal of the N>1 path on one GPU)
        self.host_staged = host_staged
        self._counts_host = self._delta_host = None

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

    def build_counts_local(self):
        self.s.build_counts()

    def sweep_local(self, sweep_idx, seed, flags=0):
        return self.s.sweep(sweep_idx, seed, flags=flags | SWEEP_NO_APPLY)

    def apply(self, topic, modality):
        self.s.apply_delta(topic, modality)

    def sync(self):
        if self.host_staged:                 # push all-reduced host copies back to the library's buffers
            if self._counts_host is not None:
                self._counts_dev.copy_(self._counts_host); self._counts_host = None
            if self._delta_host is not None:
                self._delta_dev.copy_(self._delta_host); self._delta_host = None
        torch.cuda.synchronize(self.device)


def decode_activation(key):
    """(topic, modality) of an activation key (mvhdp.h: doc<<34 | view<<31 | pos<<11 | topic)."""
    if key == KEY_NONE:
        return -1, -1
    return int(key & 0x7FF), int((key >> 31) & 0x7)


def build_counts_all_reduce(shard, group=None):
    """buildInitialTypeTopicCounts PTM:600-652 over every shard: local counts, then a sum all-reduce."""
    shard.build_counts_local()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        shard.sync()
        t = shard.counts
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        shard.sync()


def sweep_all_reduce(shard, sweep_idx, seed, group=None, flags=0, has_inactive=False):
    """One global Gibbs sweep: every shard samples its entities against the same snapshot of
    n_wk / n_k, the deltas are summed across shards, every replica applies the same sum
    (AD-LDA style).  Bit-identical to the single-shard sweep: entities are independent under
    the snapshot semantics and integer sums do not depend on the order."""
    st = shard.sweep_local(sweep_idx, seed, flags)
    topic, modality = st.activated_topic, st.activated_modality
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        shard.sync()
        t = shard.delta
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        if has_inactive:
            # UPD:263-270: the first delta in (entity, view, position) order wins, on every replica alike
            key = torch.tensor([st.activation_key], dtype=torch.int64, device=t.device)
            dist.all_reduce(key, op=dist.ReduceOp.MIN, group=group)
            topic, modality = decode_activation(int(key.item()))
        shard.sync()
    shard.apply(topic, modality)
    return st
