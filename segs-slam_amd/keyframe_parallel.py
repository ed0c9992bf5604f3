"""The one exchange of a keyframe-parallel training step, over torch.distributed (backend "nccl" = RCCL over xGMI).

Every rank holds a full replica of the flat parameter bucket and renders its own keyframe; per step the ranks exchange

  * the FLAG word: the resident rasterizer's overflow word (status[3], segs_raster.h) of each rank, summed.  It is known as
    soon as the forward's binning has run, so its (tiny) all-reduce is issued asynchronously right after the forward and
    hides behind the loss and the whole backward.  The summed word is what guards statistics and optimizer ON THE DEVICE
    (segs_training_statis_guarded, segs_adam_step_device): every replica drops the same steps and no rank synchronises
    with its device to find out.
  * the GRADIENT bucket, either
      dense:    all_reduce(sum) of the whole bucket, every rank runs the fused Adam over all of it; or
      sharded:  reduce_scatter(sum) -> rank r owns flat range [r*L/N, (r+1)*L/N) of the bucket, runs the fused Adam on that
                range only (1/N of the 28 B/parameter optimizer traffic, 1/N of the moments kept current) and the updated
                parameters travel back with one all_gather.  Same bytes per link as the ring all-reduce (reduce-scatter +
                all-gather IS a ring all-reduce), N times less HBM traffic in the optimizer.
Replicas stay bit-identical in both modes: every rank ends the step with the same all-reduced (or all-gathered) words.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring moves 2 (N-1)/N * bytes over ONE link pair per hop, so the
exchange of B bytes costs about 2 (N-1)/N * B / 153 GB/s when RCCL rings over single links, and up to 7x less when it
stripes rings over all links of the fully connected node (DESIGN.md section 6 prices both).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def _group_info(process_group) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(process_group), dist.get_rank(process_group)
    return 1, 0


class BucketExchange:
    """Gradient / parameter exchange over one flat fp32 bucket of `n` elements (see module docstring)."""

    ALIGN = 4      # shard boundaries fall on 16-byte boundaries (float4 accesses of the fused Adam)
    # sharded="auto": the sharded exchange pays two more collective launches and a pass over the gradient bucket per step
    # (measured with one rank: +0.075 ... 0.105 ms against +0.03 ... 0.05 ms for the dense one, profiles/
    # r03_single_rank_exchange_cost.txt) and saves (N-1)/N of the optimizer's 28 B per parameter: worth it from about here
    AUTO_SHARD_BYTES = 32 << 20

    def __init__(self, n: int, device, process_group=None, sharded=True, single_rank_collectives: bool = False,
                 grads: Optional[torch.Tensor] = None, offset: int = 0):
        """`offset`: the exchange covers elements [offset, offset + n) of the buckets handed to reduce_gradients() / gather()
        (which are then offset + n long); what lies in front -- a frozen segment nobody updates, e.g. the anchor positions
        under Optimization.position_lr_* = 0 -- never crosses a link.  Ranges (shard_range, clip_segments) stay in
        whole-bucket coordinates."""
        self.pg = process_group
        self.world, self.rank = _group_info(process_group)
        self.n = int(n)
        self.offset = int(offset)
        assert self.offset % self.ALIGN == 0, "the exchanged range must start on a 16-byte boundary"
        self.device = torch.device(device)
        # measurement support (bench.py): when set to a list, every collective of reduce_gradients() / gather() is bracketed
        # by two events on the current stream, appended as (label, start, end, bytes)
        self.timing = None
        if sharded == "auto":
            sharded = self.n * 4 >= self.AUTO_SHARD_BYTES
        # single_rank_collectives: issue every collective even with ONE rank in an initialised group (bench.py --force-dist,
        # tests/test_rccl_single_rank_gpu.py): the reduce_scatter_tensor / all_gather_into_tensor calls of the N > 1 path then
        # run under the real RCCL backend on a one-GPU box, where they must leave the dense path's parameters
        self.active = self.world > 1 or (bool(single_rank_collectives) and dist.is_available() and dist.is_initialized())
        self.sharded = bool(sharded) and self.active
        per = -(-self.n // self.world)
        self.shard_len = -(-per // self.ALIGN) * self.ALIGN
        self.flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._flag_work = None
        # gloo (only used to rehearse the N > 1 path, on the CPU or with every rank on one GPU) has no tensor-shaped
        # reduce-scatter / all-gather for device tensors: same result from an all-reduce + slice and a list all-gather
        self._emulate = (self.active and self.device.type == "cuda" and dist.get_backend(process_group) != "nccl")
        # DENSE exchange with a gradient bucket that has one spare element behind it (raster_engine / ScaffoldModel allocate
        # four): the overflow word rides there through the gradient all-reduce itself -- no collective of its own, no second
        # stream -- which is most of the dense exchange's fixed cost on a 0.3 ms step.  `_ext` = the bucket + that element.
        self._ext, self._piggy = None, False
        if grads is not None and self.active and not self.sharded and grads.dtype == torch.float32:
            st, off = grads.untyped_storage(), grads.storage_offset() + self.offset
            if grads.numel() == self.offset + self.n and grads.is_contiguous() and st.nbytes() >= (off + self.n + 1) * 4:
                self._ext = torch.empty(0, dtype=torch.float32, device=grads.device).set_(st, off, (self.n + 1,))
        if self.sharded:
            f = dict(dtype=torch.float32, device=self.device)
            # staging in equal-sized shards: the collectives need world * shard_len elements, the bucket has n
            self._send = torch.zeros(self.shard_len * self.world, **f) if self.shard_len * self.world != self.n else None
            self._shard = torch.zeros(self.shard_len, **f)
            self._full = torch.zeros(self.shard_len * self.world, **f) if self._send is not None else None

    # ---- ranges
    def shard_range(self, rank: Optional[int] = None) -> Tuple[int, int]:
        """[lo, hi) of the bucket this rank updates, in whole-bucket coordinates.  Not sharded: the whole bucket with one rank
        (the part in front of the exchanged range included: the optimizer then keeps that group's moments exactly as the
        reference's does); with an ACTIVE exchange only the exchanged range -- the frozen segment in front of it holds this rank's
        own, un-summed gradient, and an optimizer pass over it would let its moments differ from rank to rank (its parameters
        never move either way: the segment is frozen because its learning rate is 0)."""
        if not self.sharded:
            return (self.offset if self.active else 0), self.offset + self.n
        r = self.rank if rank is None else rank
        lo = min(r * self.shard_len, self.n)
        return self.offset + lo, self.offset + min(lo + self.shard_len, self.n)

    def clip_segments(self, segments: Sequence[Tuple[int, int, float]]) -> List[Tuple[int, int, float]]:
        """(offset, count, lr) Adam segments intersected with this rank's shard."""
        lo, hi = self.shard_range()
        out = []
        for off, cnt, lr in segments:
            a, b = max(off, lo), min(off + cnt, hi)
            if b > a:
                out.append((a, b - a, lr))
        return out

    # ---- flag
    def reduce_flag_async(self, local_flag: Optional[torch.Tensor], allow_piggyback: bool = True):
        """Start the all-reduce of this step's overflow word (a 1-element int32 device tensor, or None for 0).  With one rank
        the word itself is the guard: no copy, no launch.  Dense exchange over a bucket with a spare tail element: the word is
        only PLACED there (as a float) and summed by reduce_gradients(); `allow_piggyback=False` for the steps that must read
        the summed word on the host before the gradients are exchanged (adjust_anchor iterations)."""
        self._piggy = False
        self._piggy_summed = False
        if not self.active:
            self._local = local_flag
            return
        if self._ext is not None and allow_piggyback:
            tail = self._ext[self.n:]
            if local_flag is None:
                tail.zero_()
            else:
                tail.copy_(local_flag.reshape(1))       # int32 0 / 1 -> float 0.0 / 1.0
            self._piggy = True
            return
        if local_flag is None:
            self.flag.zero_()
        else:
            self.flag.copy_(local_flag.reshape(1))
        self._flag_work = dist.all_reduce(self.flag, group=self.pg, async_op=True)

    def wait_flag(self) -> torch.Tensor:
        """Make the current stream wait for the flag; returns the device word (non-zero: some rank's pass is invalid).

        CONTRACT when the word rides with the gradients (dense exchange, allow_piggyback): the returned view is a FLOAT32
        element that holds only this rank's own word until reduce_gradients() has run; it is the summed word -- any non-zero
        bit pattern means drop -- only for work enqueued after that call.  A caller that needs the sum earlier (a host read
        before the gradient exchange) must pass allow_piggyback=False to reduce_flag_async(); flag_on_host() enforces this."""
        if self._piggy:
            return self._ext[self.n:]
        if not self.active:
            if getattr(self, "_local", None) is not None:
                return self._local
            if not getattr(self, "_zeroed", False):     # (the buffer is never written with one rank)
                self.flag.zero_()
                self._zeroed = True
            return self.flag
        if self._flag_work is not None:
            self._flag_work.wait()
            self._flag_work = None
        return self.flag

    def flag_on_host(self) -> int:
        """The summed overflow word read on the host (synchronises).  Refuses to hand out an un-summed word."""
        if self._piggy and not self._piggy_summed:
            raise RuntimeError("the overflow word rides with the gradients and has not been summed yet: call reduce_gradients() "
                               "first, or start the step with reduce_flag_async(..., allow_piggyback=False)")
        w = self.wait_flag()
        return int(w.view(torch.int32).item()) if w.dtype == torch.float32 else int(w.item())

    # ---- the summed word on the host, one step late (so that NO iteration is lost with N > 1 ranks either)
    def mirror_flag(self):
        """Queue a copy of the SUMMED overflow word into pinned host memory and an event behind it.  Call when the word is final
        for work enqueued now: after reduce_gradients() (the word may ride in the gradient all-reduce).  Every rank mirrors the
        same word, so every rank takes the same decision in step_dropped() -- no further collective."""
        w = self.wait_flag()
        if self.device.type != "cuda":
            self._mirror = ("value", int(w.view(torch.int32).reshape(-1)[0]) if w.dtype == torch.float32 else int(w.reshape(-1)[0]))
            return
        if getattr(self, "_mirror_host", None) is None:
            self._mirror_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._mirror_event = torch.cuda.Event()
        self._mirror_host.copy_(w.view(torch.int32).reshape(1), non_blocking=True)   # (a float 1.0, 2.0 ... is a non-zero bit pattern)
        self._mirror_event.record()
        self._mirror = ("event", None)

    def step_dropped(self):
        """True / False: the device dropped / took the last mirrored step (waits for that step's exchange, not for its optimizer);
        None when nothing was mirrored since the last call."""
        m = getattr(self, "_mirror", None)
        self._mirror = None
        if m is None:
            return None
        if m[0] == "value":
            return m[1] != 0
        self._mirror_event.synchronize()
        return int(self._mirror_host[0]) != 0

    def _timed(self, label: str, nbytes: int, fn):
        if self.timing is None or self.device.type != "cuda":
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        self.timing.append((label, a, b, nbytes))
        return out

    # ---- gradients
    def reduce_gradients(self, grads: torch.Tensor, dense: bool = False):
        """dense (or an unsharded exchange): grads <- sum over ranks (in place).  sharded: grads[lo:hi] <- sum over ranks; the
        rest of the bucket still holds this rank's own contribution and must be cleared by the caller before the next
        backward accumulates.  `dense=True` is for the steps whose shard partition changes between this call and the
        optimizer (a densification that re-sizes the bucket): every element must then hold the sum, whoever updates it."""
        assert grads.numel() == self.offset + self.n
        if not self.active:
            return
        if self.offset:
            grads[:self.offset].zero_()      # nobody reads the frozen segment's gradient (shard_range): it must not pile up
            grads = grads[self.offset:]
        if self._piggy:
            assert grads.data_ptr() == self._ext.data_ptr(), "the exchange was built for another gradient bucket"
            self._timed("all_reduce", (self.n + 1) * 4, lambda: dist.all_reduce(self._ext, group=self.pg))      # gradients + the overflow word behind them
            self._piggy_summed = True
            return
        if dense or not self.sharded:
            self._timed("all_reduce", self.n * 4, lambda: dist.all_reduce(grads, group=self.pg))
            return
        if self._emulate:
            # same RESULT as the reduce-scatter below, outside the shard included: there the bucket keeps this rank's own
            # contribution (an emulation that left the full sum everywhere would hide a step reading outside its shard)
            lo, hi = (x - self.offset for x in self.shard_range())
            own = grads.clone()
            dist.all_reduce(grads, group=self.pg)
            own[lo:hi] = grads[lo:hi]
            grads.copy_(own)
            return
        src = grads
        if self._send is not None:
            self._send[:self.n].copy_(grads)
            src = self._send
        lo, hi = (x - self.offset for x in self.shard_range())
        self._timed("reduce_scatter", self.shard_len * self.world * 4, lambda: dist.reduce_scatter_tensor(self._shard, src, group=self.pg))
        grads[lo:hi].copy_(self._shard[:hi - lo])

    # ---- parameters
    def gather(self, bucket: torch.Tensor):
        """bucket <- concatenation of every rank's shard (parameters after the sharded Adam; moments before a densification)."""
        if not self.sharded:
            return
        assert bucket.numel() == self.offset + self.n
        if self.offset:
            bucket = bucket[self.offset:]
        lo, hi = (x - self.offset for x in self.shard_range())
        if hi - lo < self.shard_len:
            self._shard[hi - lo:].zero_()          # (only the last rank's shard has a padded tail)
        self._shard[:hi - lo].copy_(bucket[lo:hi])
        if self._emulate:
            parts = [torch.empty_like(self._shard) for _ in range(self.world)]
            dist.all_gather(parts, self._shard, group=self.pg)
            for r, part in enumerate(parts):
                a, b = (x - self.offset for x in self.shard_range(r))
                bucket[a:b].copy_(part[:b - a])
        elif self._full is None:
            self._timed("all_gather", self.shard_len * self.world * 4, lambda: dist.all_gather_into_tensor(bucket, self._shard, group=self.pg))
        else:
            self._timed("all_gather", self.shard_len * self.world * 4, lambda: dist.all_gather_into_tensor(self._full, self._shard, group=self.pg))
            bucket.copy_(self._full[:self.n])
