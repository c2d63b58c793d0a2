"""Data-parallel gradient exchange: one process per GPU, one flat fp32 buffer, bucketed RCCL all-reduces overlapped with backward.

The reference wraps the model in torch DDP with find_unused_parameters=True (pipelines/base_pipeline.py:241-243):
bucketed NCCL all-reduce of ~58 MB of gradients plus a per-step graph walk.  Every rank here owns a full replica and
its own rays (scripts/train.py:86); parameter gradients are VIEWS into a single contiguous buffer, so a bucket's
exchange is an in-place all-reduce (mean) over xGMI of one slice with no packing copies.  A bucket (one per parameter
group by default: "fields", "proposal_networks") is reduced as soon as autograd has accumulated the last of its
gradients (post-accumulate-grad hooks), i.e. the field's 58 MB travel while the proposal networks' backward still
runs; ``finish()`` waits for the outstanding collectives before the optimiser step.

``set_active_numel(param, n)`` restricts a parameter's exchange to its first ``n`` elements: with the progressive
level mask of neus-facto-angelo (sdf_field.py:376-378) the hash-table rows of the masked levels have exactly zero
gradient on every rank, so BASELINE config 5's 1.8 GB table moves only its active prefix.
"""
from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def _dist_on(group=None) -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


class FlatGradients:
    def __init__(self, params: Iterable[torch.nn.Parameter], buckets: Optional[Sequence[Sequence[torch.nn.Parameter]]] = None,
                 group=None, overlap: bool = True):
        """params: every parameter whose gradient lives in the flat buffer, in buffer order.  buckets: a partition of them
        into exchange units (default: one bucket); each bucket's parameters must be contiguous in `params`."""
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
        self._work = []
        self._overlap = overlap
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._arm()

    # ---- buffer ownership
    def _view(self, p):
        off = self._offset[id(p)]
        return self.flat[off:off + p.numel()].view_as(p)

    def _attach(self):
        for p in self.params:
            p.grad = self._view(p)

    def _check_attached(self):
        """optimizer.zero_grad() / module.zero_grad() default to set_to_none=True, which drops the views; autograd then
        allocates fresh gradients and the flat buffer goes stale.  Detect that and fold the stray gradients back in."""
        for p in self.params:
            want = self.flat.data_ptr() + 4 * self._offset[id(p)]
            if p.grad is None:
                p.grad = self._view(p)
            elif p.grad.data_ptr() != want:
                v = self._view(p)
                v.copy_(p.grad)
                p.grad = v

    def zero(self):
        """Clear every gradient (the ONLY way gradients of these parameters should be cleared) and re-arm the buckets."""
        self.flat.zero_()
        self._check_attached()
        self._arm()

    zero_grad = zero

    def set_active_numel(self, param: torch.nn.Parameter, numel: Optional[int]):
        """Exchange only param.view(-1)[:numel] (None: all of it); the rest must be identically zero on every rank."""
        if numel is None or numel >= param.numel():
            self._active.pop(id(param), None)
        else:
            self._active[id(param)] = max(int(numel), 0)

    # ---- exchange
    def _arm(self):
        for bi, b in enumerate(self._buckets):
            self._pending[bi] = len(b)
            self._launched[bi] = False
        self._work = []

    def _ranges(self, bi):
        """Contiguous [start, end) element ranges of bucket bi that have to travel."""
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
        return [(a, b) for a, b in out if b > a]

    def _launch(self, bi):
        if self._launched[bi]:
            return
        self._launched[bi] = True
        if not _dist_on(self.group):
            return
        for a, b in self._ranges(bi):
            self._work.append((dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True), a, b))

    def _on_grad(self, p):
        bi = self._bucket_of[id(p)]
        self._pending[bi] -= 1
        if self._pending[bi] == 0 and self._overlap:
            self._launch(bi)

    def finish(self, average: bool = True) -> float:
        """Wait for the outstanding bucket all-reduces (launching any bucket whose hooks did not all fire, e.g. parameters
        unused in this step).  average=True turns the sums into means in place; average=False leaves the SUMS and returns the
        scale (1 / world_size) for the consumer to apply - the fused Adam step multiplies the gradient by it as it reads it
        (engine/optimizers.py), which saves the pass over the buffer."""
        self._check_attached()
        for bi in range(len(self._buckets)):
            self._launch(bi)
        scale = 1.0
        if _dist_on(self.group):
            w = dist.get_world_size(self.group)
            scale = 1.0 / w
            for work, a, b in self._work:
                work.wait()
                if average:
                    self.flat[a:b].div_(w)
        self._work = []
        return 1.0 if average else scale

    all_reduce_mean = finish

    def exchanged_numel(self) -> int:
        return sum(b - a for bi in range(len(self._buckets)) for a, b in self._ranges(bi))


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    """DDP's initial parameter broadcast from rank 0 (base_pipeline.py:242)."""
    if _dist_on():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)
