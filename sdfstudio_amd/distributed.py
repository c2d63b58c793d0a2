"""Data-parallel gradient exchange: one process per GPU, one flat fp32 buffer, bucketed RCCL all-reduces overlapped with backward.

The reference wraps the model in torch DDP with find_unused_parameters=True (pipelines/base_pipeline.py:241-243):
bucketed NCCL all-reduce of ~58 MB of gradients plus a per-step graph walk.  Every rank here owns a full replica and
its own rays (scripts/train.py:86); parameter gradients are VIEWS into a single contiguous buffer, so a bucket's
exchange is an in-place all-reduce (mean) over xGMI of one slice with no packing copies.  A bucket (one per parameter
group by default: "fields", "proposal_networks") is reduced as soon as autograd has accumulated the last of its
gradients (post-accumulate-grad hooks), i.e. the field's 58 MB travel while the proposal networks' backward still
runs; ``finish()`` waits for the outstanding collectives before the optimiser step.

Collective order.  Every rank issues the SAME sequence of all-reduces over the SAME slices: buckets are launched strictly in
index order (bucket k leaves from a hook only once buckets 0..k-1 have left; otherwise it waits for ``finish()``, which launches
what is left, again in index order), and a bucket larger than ``chunk_numel`` elements travels as fixed-size chunks in address
order.  A rank whose autograd graph misses a whole parameter group in some step (the zero-sample branch of NeuS-acc, a background
field no ray hits) therefore still matches its peers, as DDP's fixed bucket order does; its untouched gradients are the zeros
``zero()`` left.

``set_active_numel(param, n)`` restricts a parameter's exchange to its first ``n`` elements: with the progressive
level mask of neus-facto-angelo (sdf_field.py:376-378) the hash-table rows of the masked levels have exactly zero
gradient on every rank, so BASELINE config 5's 1.8 GB table moves only its active prefix.

Protocol per step: ``zero()`` -> one backward -> ``finish()``.  Anything else raises: a second backward before ``finish()`` would
accumulate into slices that are being reduced, a backward without ``zero()`` would mix last step's means with new local sums.
"""
import time
from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist

from sdfstudio_amd.grad_slots import CLAIM_ATTR, SLOT_ATTR  # noqa: F401


def _dist_on(group=None) -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


class FlatGradients:
    def __init__(self, params: Iterable[torch.nn.Parameter], buckets: Optional[Sequence[Sequence[torch.nn.Parameter]]] = None,
                 group=None, overlap: bool = True, chunk_numel: Optional[int] = 32 * 1024 * 1024):
        """params: every parameter whose gradient lives in the flat buffer, in buffer order.  buckets: a partition of them
        into exchange units (default: one bucket); each bucket's parameters must be contiguous in `params`.  chunk_numel: a
        bucket's ranges travel in pieces of at most this many elements (128 MB by default: config 2's 50 MB field bucket is one
        collective, config 5's 1.8 GB table is 14, so the ring starts delivering finished pieces while later ones are in
        flight and no single collective monopolises the links); None: one collective per contiguous range."""
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self._offset: Dict[int, int] = {}
        off = 0
        for p in self.params:
            self._offset[id(p)] = off
            off += p.numel()
        self._active: Dict[int, int] = {}
        # high-water mark of set_active_numel per parameter: elements beyond it have NEVER carried a gradient, so they are still the
        # zeros the buffer was created with (zero() need not rewrite them, Adam need not visit them: live_ranges())
        self._hwm: Dict[int, int] = {}
        self._finished_steps = 0
        self._attach()
        if buckets is None:
            buckets = [self.params]
        self._buckets: List[List[torch.nn.Parameter]] = [[p for p in b if p.requires_grad] for b in buckets]
        self._buckets = [b for b in self._buckets if b]
        seen = [id(p) for b in self._buckets for p in b]
        assert sorted(seen) == sorted(self._offset), "buckets must partition the parameters"
        for b in self._buckets:
            offs = [self._offset[id(p)] for p in b]
            assert offs == sorted(offs) and all(offs[i] + b[i].numel() == offs[i + 1] for i in range(len(b) - 1)), \
                "a bucket's parameters must be contiguous in the flat buffer"
        self._bucket_of = {id(p): bi for bi, b in enumerate(self._buckets) for p in b}
        self._pending = [0] * len(self._buckets)
        self._launched = [False] * len(self._buckets)
        self._next = 0          # first bucket that has not been launched: launches happen in index order only
        self._armed = False
        self._work = []
        self._overlap = overlap
        self._chunk = None if chunk_numel is None else max(int(chunk_numel), 1)
        # diagnostics of the last step (bench.py prints them per rank): seconds finish() spent blocked in work.wait(), collectives
        # issued, buckets that left from the autograd hooks (i.e. overlapped with the rest of backward)
        self.last_wait_s = 0.0
        self.last_collectives = 0
        self.last_overlapped_buckets = 0
        # with time_waits = True on a CUDA buffer, every finish() brackets its waits with a pair of events on the current stream:
        # the GPU time between them is the part of the exchange that was NOT hidden behind backward (wait() itself only makes
        # the stream wait, the host returns at once); exposed_ms() reads them
        self.time_waits = False
        self._wait_events = []
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._arm()

    # ---- buffer ownership
    def _view(self, p):
        off = self._offset[id(p)]
        return self.flat[off:off + p.numel()].view_as(p)

    def _attach(self):
        for p in self.params:
            p.grad = self._view(p)
            # native backward kernels write this parameter's gradient straight into its slice (grad_slots.py)
            setattr(p, SLOT_ATTR, (lambda q=p: self._view(q)))
            setattr(p, CLAIM_ATTR, False)

    def _is_view(self, p) -> bool:
        return p.grad is not None and p.grad.data_ptr() == self.flat.data_ptr() + 4 * self._offset[id(p)]

    def zero(self):
        """Clear every gradient (the ONLY way gradients of these parameters should be cleared) and re-arm the buckets.
        optimizer.zero_grad() / module.zero_grad() default to set_to_none=True, which drops the views (autograd would then
        allocate fresh gradients and the flat buffer would go stale): the views are re-attached here, and whatever stray
        tensor sat in .grad is DISCARDED - this call means "all gradients are zero now"."""
        if self._work:
            raise RuntimeError("FlatGradients.zero() while all-reduces of the previous backward are in flight: call finish() first")
        for a, b in self.live_ranges():  # one range (the whole buffer) unless a parameter has a never-active suffix
            self.flat[a:b].zero_()
        for p in self.params:
            # None, not a view: AccumulateGrad then ADOPTS the first incoming gradient instead of adding it into .grad - and the native
            # backward kernels hand it a view of this buffer they have already written (grad_slots.py): no launch per parameter
            p.grad = None
            setattr(p, CLAIM_ATTR, False)
        self._arm()

    zero_grad = zero

    def set_active_numel(self, param: torch.nn.Parameter, numel: Optional[int]):
        """Exchange only param.view(-1)[:numel] (None: all of it); the rest must be identically zero on every rank."""
        if numel is None or numel >= param.numel():
            self._active.pop(id(param), None)
            self._hwm.pop(id(param), None)
        else:
            n = max(int(numel), 0)
            self._active[id(param)] = n
            # a restriction that arrives after unrestricted steps finds gradients (and Adam moments) beyond it: nothing to skip then
            first = param.numel() if (id(param) not in self._hwm and self._finished_steps > 0) else 0
            self._hwm[id(param)] = max(self._hwm.get(id(param), first), n)

    def mark_all_live(self):
        """Forget the never-active suffixes (e.g. after loading optimizer moments from a checkpoint)."""
        for k in list(self._hwm):
            self._hwm[k] = 1 << 62

    def live_ranges(self):
        """Contiguous [start, end) ranges of the flat buffer that may hold a non-zero gradient: everything except the suffix of a
        parameter that set_active_numel has kept inactive since the first step (progressive hash levels: the table rows of levels
        that have not been switched on yet - BASELINE config 5 starts with 8 of 16 levels of a 2.1 GB table).  zero() rewrites and
        the fused Adam step visits only these: beyond them gradient, moments and update are exactly zero."""
        out = []
        for p in self.params:
            off = self._offset[id(p)]
            n = min(self._hwm.get(id(p), p.numel()), p.numel())
            if n > 0:
                if out and out[-1][1] == off:
                    out[-1][1] = off + n
                else:
                    out.append([off, off + n])
            if n < p.numel():
                out.append([off + p.numel(), off + p.numel()])  # break the run: the next parameter starts a new range
        return [(a, b) for a, b in out if b > a]

    # ---- exchange
    def _arm(self):
        for bi, b in enumerate(self._buckets):
            self._pending[bi] = len(b)
            self._launched[bi] = False
        self._next = 0
        self._work = []
        self._armed = True
        self.last_overlapped_buckets = 0
        self.last_collectives = 0

    def _ranges(self, bi):
        """Contiguous [start, end) element ranges of bucket bi that have to travel, cut into chunks of at most chunk_numel."""
        out = []
        for p in self._buckets[bi]:
            off = self._offset[id(p)]
            n = self._active.get(id(p), p.numel())
            if n <= 0:
                continue
            if out and out[-1][1] == off:
                out[-1][1] = off + n
            else:
                out.append([off, off + n])
            if n < p.numel():  # a gap follows: the next parameter starts a new range
                out.append([off + p.numel(), off + p.numel()])
        cut = []
        for a, b in out:
            if b <= a:
                continue
            step = (b - a) if self._chunk is None else self._chunk
            for s in range(a, b, step):
                cut.append((s, min(s + step, b)))
        return cut

    def _launch(self, bi):
        assert bi == self._next and not self._launched[bi], "buckets leave in index order"
        self._launched[bi] = True
        self._next = bi + 1
        if not _dist_on(self.group):
            return
        for a, b in self._ranges(bi):
            self._work.append((dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True), a, b))
            self.last_collectives += 1

    def _launch_ready(self, from_hook: bool):
        while self._next < len(self._buckets) and self._pending[self._next] == 0:
            self._launch(self._next)
            if from_hook:
                self.last_overlapped_buckets += 1

    def _on_grad(self, p):
        if not self._armed:
            raise RuntimeError("FlatGradients: backward without zero() since the last finish() - the flat buffer still holds the "
                               "reduced gradients of the previous step")
        bi = self._bucket_of[id(p)]
        self._pending[bi] -= 1
        if self._pending[bi] < 0 or self._launched[bi]:
            raise RuntimeError("FlatGradients: a second backward reached a parameter before finish() - its bucket may already be "
                               "in flight (accumulate micro-batches into one loss, or call finish() / zero() between backwards)")
        if not self._is_view(p):
            # AccumulateGrad runs once per leaf and backward with the SUM of everything that reached the parameter.  It adopted an
            # ordinary tensor: a torch op produced the gradient, or several producers did and the engine summed them out of place (a
            # native kernel's slice is then one of the summands, already inside this total), or it cloned the view it was handed.
            # In every case p.grad is the whole gradient of this pass: it REPLACES the slice.
            v = self._view(p)
            v.copy_(p.grad)
            p.grad = v
        if self._overlap:
            self._launch_ready(from_hook=True)

    def finish(self, average: bool = True) -> float:
        """Wait for the outstanding bucket all-reduces (launching, in index order, every bucket that has not left yet: e.g. one
        behind a bucket with parameters unused in this step).  average=True turns the sums into means in place; average=False
        leaves the SUMS and returns the scale (1 / world_size) for the consumer to apply - the fused Adam step multiplies the
        gradient by it as it reads it (engine/optimizers.py), which saves the pass over the buffer."""
        if not self._armed:
            raise RuntimeError("FlatGradients.finish() twice without zero() + backward in between")
        for p in self.params:
            if not self._is_view(p):
                if p.grad is None:
                    p.grad = self._view(p)  # unused in this step: its gradient is the zeros zero() left
                    continue
                if self._launched[self._bucket_of[id(p)]]:
                    raise RuntimeError("FlatGradients: a gradient outside the flat buffer appeared after its bucket was launched")
                v = self._view(p)  # stray gradient of a parameter whose hook never fired (grad set by hand): fold it in
                v.add_(p.grad)
                p.grad = v
        for bi in range(self._next, len(self._buckets)):
            self._launch(bi)
        scale = 1.0
        self.last_wait_s = 0.0
        if _dist_on(self.group):
            w = dist.get_world_size(self.group)
            scale = 1.0 / w
            ev = None
            if self.time_waits and self.flat.is_cuda:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            t0 = time.perf_counter()
            for work, a, b in self._work:
                work.wait()
                if average:
                    self.flat[a:b].div_(w)
            self.last_wait_s = time.perf_counter() - t0
            if ev is not None:
                ev[1].record()
                self._wait_events.append(ev)
        self._work = []
        self._armed = False
        self._finished_steps += 1
        return 1.0 if average else scale

    all_reduce_mean = finish

    def exposed_ms(self, reset: bool = True) -> List[float]:
        """GPU milliseconds the compute stream stalled on the exchange in every finish() since the last reset (time_waits)."""
        if self._wait_events:
            torch.cuda.synchronize(self.flat.device)
        out = [a.elapsed_time(b) for a, b in self._wait_events]
        if reset:
            self._wait_events = []
        return out

    def exchanged_numel(self) -> int:
        return sum(b - a for bi in range(len(self._buckets)) for a, b in self._ranges(bi))


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    """DDP's initial parameter broadcast from rank 0 (base_pipeline.py:242)."""
    if _dist_on():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)
