"""Data-parallel gradient reduction: one process per GPU, RCCL all-reduce over xGMI.

Replaces the single-process ``torch.nn.DataParallel`` of the reference
(/root/reference/models/unetbaseline_model.py:52-56; implicit broadcast / scatter / gather /
reduce_add per step) with replicated weights and ONE exchange per step:

  * the flat f32 gradient buffer of the engine is cut into contiguous buckets from its END towards
    its start -- the order in which backward finalises gradients (outermost up layer first) -- and
    each bucket is all-reduced (SUM) as soon as it is final, so the collectives overlap the rest of
    backward (torch.distributed backend "nccl" == RCCL on ROCm, its own stream);
  * a 4-double all-reduce of the loss statistics (N, sum|e|, sum d, sum d^2) between forward and
    backward reproduces DataParallel's single global-batch loss exactly (the SIlog term is not
    decomposable over shards); with it every rank back-propagates d(global loss)/d(local pred),
    hence the gradient reduction is a SUM, not a mean;
  * BatchNorm statistics stay per replica (DataParallel semantics, no SyncBN): no collective.
Bucket size: xGMI is point-to-point (7 links x ~153 GB/s per GPU) so collectives are per-link
bound; 217.6 MB of U-Net gradients go out as a few large (default 32 MiB) buckets rather than 43
per-tensor messages.  Works with the gloo backend on CPU tensors too (tests, world_size 2).
Optional ``payload='bf16'``: every bucket is cast to bf16 before the exchange and back afterwards (108.8 MB instead of
217.6 MB on the links for the U-Net; SURVEY section 5: the exchange time decides the 8-GPU scaling target).  The SUM then
runs in bf16 -- about 3 significant digits per element, the order of the bf16 compute path's own gradient error -- and
finish() OVERWRITES the f32 gradients with that bf16 sum; the default stays f32, which is what the parity tests
(bit-identical replicas vs a single-process emulation) pin.
Clip norm: ``enable_bucket_norm()`` makes finish() take the sum of squares of each bucket as its collective lands
(adn_grad_sqsum_partials), so the norm pass over the reduced gradients overlaps the buckets still in flight.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradientAllReducer:
    def __init__(self, process_group=None, bucket_bytes: int = 32 << 20, payload: str = 'f32'):
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError('GradientAllReducer needs an initialised torch.distributed process group')
        if payload not in ('f32', 'bf16'):
            raise ValueError(f"payload must be 'f32' or 'bf16', got {payload!r}")
        self.pg = process_group
        self.payload = payload
        self.g16 = None            # bf16 staging buffer of the exchange (payload='bf16')
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.buckets = []          # (lo, hi) element ranges, ordered from the end of the buffer
        self.flat_g = None
        self._next = 0
        self._works = []
        self.norm_slots = None     # f64 partial sums of squares per bucket (enable_bucket_norm)

    @property
    def world_size(self):
        return dist.get_world_size(self.pg)

    def attach(self, engine):
        """Bind to an engine (anything exposing ``flat_g`` and an ``on_grad_ready`` hook slot)."""
        self.flat_g = engine.flat_g
        n = self.flat_g.numel()
        self.buckets = []
        hi = n
        while hi > 0:
            lo = max(0, hi - self.bucket_elems)
            self.buckets.append((lo, hi))
            hi = lo
        engine.on_grad_ready = self.on_grad_ready
        self._next = 0
        self._works = []
        self.g16 = torch.empty(n, dtype=torch.bfloat16, device=self.flat_g.device) if self.payload == 'bf16' else None

    def enable_bucket_norm(self):
        """Take the clip norm bucket by bucket: finish() leaves the sums of squares of every reduced bucket in
        ``norm_slots`` (f64), computed as each collective lands while the later ones are still on the links, instead
        of one pass over the whole gradient buffer after the exchange (device tensors only).  Returns the slots."""
        from . import kernels as K
        counts = [K.grad_sqsum_count(hi - lo) for lo, hi in self.buckets]
        self._slot_ranges, at = [], 0
        for c in counts:
            self._slot_ranges.append((at, at + c))
            at += c
        self.norm_slots = torch.zeros(at, dtype=torch.float64, device=self.flat_g.device)
        return self.norm_slots

    def broadcast_parameters(self, flat_p, src=0):
        """Replicate rank ``src``'s weights (what DataParallel.replicate does every forward; once here)."""
        dist.broadcast(flat_p, src=src, group=self.pg)

    def all_reduce_loss_stats(self, stats):
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.pg)

    def begin_backward(self):
        self._next = 0
        self._works = []

    def on_grad_ready(self, offset_lo: int):
        """flat_g[offset_lo:] is final: launch every bucket that lies entirely above the watermark."""
        while self._next < len(self.buckets) and self.buckets[self._next][0] >= offset_lo:
            lo, hi = self.buckets[self._next]
            if self.g16 is not None:
                # cast on the compute stream (ordered behind the kernels that produced the bucket), exchange bf16
                self.g16[lo:hi].copy_(self.flat_g[lo:hi])
                buf = self.g16[lo:hi]
            else:
                buf = self.flat_g[lo:hi]
            self._works.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), lo, hi))
            self._next += 1

    def finish(self):
        """Flush the remaining buckets and make the compute stream wait for all collectives."""
        self.on_grad_ready(0)
        for i, (w, lo, hi) in enumerate(self._works):
            w.wait()                              # the compute stream now waits for the collective's stream
            if self.g16 is not None:
                self.flat_g[lo:hi].copy_(self.g16[lo:hi])
            if self.norm_slots is not None:
                from . import kernels as K
                a, b = self._slot_ranges[i]
                K.grad_sqsum_partials(self.flat_g[lo:hi], self.norm_slots[a:b])
        self._works = []


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun); returns (rank, world, local_rank)."""
    import os
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = os.environ.get('ADN_DIST_BACKEND', backend)      # e.g. gloo to rehearse N>1 on one GPU
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local
