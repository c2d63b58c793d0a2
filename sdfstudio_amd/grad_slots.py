"""Gradient slots: where a native backward may write a parameter's gradient so that autograd needs no kernel to take it.

``distributed.FlatGradients`` owns one flat fp32 buffer for the gradients of all parameters.  Autograd's AccumulateGrad node normally
ADDS an incoming gradient into ``p.grad`` - one elementwise launch per parameter and backward (about 50 per step for the SDF field's
42 weight-normalised tensors, the hash table and the proposal networks) plus the producer's own zero-fill.  When ``p.grad`` is None,
AccumulateGrad instead TAKES the incoming tensor as the gradient (no kernel) provided nobody else references it.  So:

* ``FlatGradients.zero()`` zeroes the flat buffer, sets ``p.grad = None`` and un-claims every slot;
* a native backward asks ``grad_target(p)`` where to write: the first producer of a parameter in a backward pass gets a FRESH VIEW of
  the parameter's slice of the flat buffer (already zero: kernels may assign or accumulate into it) and returns it as the gradient -
  autograd adopts it and ``p.grad`` aliases the flat buffer; any later producer gets an ordinary fresh tensor, which autograd adds;
* without a FlatGradients (tests, the reference's trainer) there are no slots and everything is ordinary autograd.
AccumulateGrad runs once per parameter and backward with the engine's SUM of all producers; when that total is not the slot view
(several producers summed out of place, or the view was cloned) the post-accumulate hook of FlatGradients copies it over the slice.
"""
from typing import Tuple

import torch

SLOT_ATTR = "_sdfhip_grad_slot"        # callable() -> fresh view of the flat gradient buffer, set by FlatGradients
CLAIM_ATTR = "_sdfhip_slot_claimed"    # True once a producer has written the slot in this backward pass


def grad_target(p: torch.Tensor, zero_init: bool = False) -> Tuple[torch.Tensor, bool]:
    """(tensor to write the gradient of parameter ``p`` into, is_slot).  zero_init: the producer ACCUMULATES (atomics), so an
    ordinary tensor must start at zero (a slot already is)."""
    slot = getattr(p, SLOT_ATTR, None)
    if slot is not None and p.grad is None and not getattr(p, CLAIM_ATTR, False):
        setattr(p, CLAIM_ATTR, True)
        return slot(), True
    return (torch.zeros_like(p) if zero_init else torch.empty_like(p)), False
