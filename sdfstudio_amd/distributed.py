"""Data-parallel gradient exchange: one process per GPU, one flat fp32 buffer, one RCCL all-reduce per step.

The reference wraps the model in torch DDP with find_unused_parameters=True (pipelines/base_pipeline.py:241-243):
bucketed NCCL all-reduce of ~58 MB of gradients plus a per-step graph walk.  Every rank here owns a full replica and
its own rays (scripts/train.py:86); parameter gradients are VIEWS into a single contiguous buffer so the exchange is a
single in-place all-reduce (mean) over xGMI with no packing copies, issued once backward has finished.
"""
from typing import Iterable, List

import torch
import torch.distributed as dist


class FlatGradients:
    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def zero(self):
        self.flat.zero_()

    def all_reduce_mean(self, group=None):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    """DDP's initial parameter broadcast from rank 0 (base_pipeline.py:242)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)
