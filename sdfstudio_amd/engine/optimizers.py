"""Optimisers and LR schedulers, mirroring nerfstudio/engine/optimizers.py:93-160 and nerfstudio/engine/schedulers.py.

The reference builds one ``torch.optim.Adam`` per parameter group (eps 1e-15; lr 5e-4 for "fields", 1e-2 for
"proposal_networks", method_configs.py:483-500) and a LambdaLR scheduler per group.  Here the parameters, their gradients
and the two Adam moments of ALL groups live in four flat fp32 buffers (parameters become views, like the gradients of
``distributed.FlatGradients``), a group is a contiguous slice, and one step of a group is ONE launch of
``sdfhip_adam_step`` - with the data-parallel mean folded into the gradient read, so the all-reduce can be a plain SUM.
There is no PyTorch fallback (``FusedAdam.step`` raises without the library or on CPU tensors); the formula as plain torch ops
is ``oracle.sdf_path.adam_reference``, the checker the tests compare with.
"""
import math
from typing import Callable, Dict, Iterable, List, Optional

import numpy as np
import torch

from sdfstudio_amd import _lib
from sdfstudio_amd.distributed import FlatGradients


# ---------------------------------------------------------------------------------------------- schedulers (multiplicative factors)
def neus_scheduler(warm_up_end: int = 5000, learning_rate_alpha: float = 0.05, max_steps: int = 300000) -> Callable[[int], float]:
    """schedulers.py:170-189 NeuSScheduler: linear warm-up, then cosine decay to alpha."""

    def func(step: int) -> float:
        if step < warm_up_end:
            return step / warm_up_end
        progress = (step - warm_up_end) / (max_steps - warm_up_end)
        return float((np.cos(np.pi * progress) + 1.0) * 0.5 * (1 - learning_rate_alpha) + learning_rate_alpha)

    return func


def multi_step_warmup_scheduler(warm_up_end: int = 5000, milestones=(300000, 400000, 500000), gamma: float = 0.33) -> Callable[[int], float]:
    """schedulers.py:191-222 MultiStepWarmupScheduler."""

    def func(step: int) -> float:
        if step < warm_up_end:
            return step / warm_up_end
        return float(gamma ** int(np.searchsorted(list(milestones), step, side="left")))

    return func


def multi_step_scheduler(max_steps: int = 1000000, gamma: float = 0.33) -> Callable[[int], float]:
    """schedulers.py:120-132 MultiStepSchedulerConfig -> MultiStepLR(milestones = max_steps / 2, 3/4, 9/10, gamma = 0.33)."""
    milestones = [max_steps // 2, max_steps * 3 // 4, max_steps * 9 // 10]
    return lambda step: float(gamma ** sum(1 for m_ in milestones if step >= m_))


def exponential_decay_scheduler(decay_rate: float = 0.1, max_steps: int = 1000000) -> Callable[[int], float]:
    """schedulers.py:137-152 ExponentialSchedulerConfig -> lr_scheduler.ExponentialLR(gamma = decay_rate ** (1 / max_steps))."""
    gamma = decay_rate ** (1.0 / max_steps)
    return lambda step: float(gamma ** step)


# ---------------------------------------------------------------------------------------------- flat parameter storage
class FlatParameters:
    """Moves the given parameters into one contiguous fp32 buffer (``p.data`` becomes a view) in the given order - the same
    order as ``FlatGradients`` - so that a parameter group is one slice of parameter, gradient and moment buffers alike."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.offset: Dict[int, int] = {}
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                self.flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat[off:off + n].view_as(p)
                self.offset[id(p)] = off
                off += n


class FusedAdam:
    """Adam over flat buffers: ``groups`` maps a name to (parameters, lr); one native launch per group and step.

    ``flat_grads``: the ``FlatGradients`` of the same parameters in the same order (its buffer is read directly; with
    ``grad_scale`` the all-reduce may stay a SUM).  The moments start at zero like torch.optim.Adam's."""

    def __init__(self, groups: Dict[str, Dict], flat_grads: FlatGradients, betas=(0.9, 0.999), eps: float = 1e-15,
                 weight_decay: float = 0.0):
        self.groups = {}
        order = [p for g in groups.values() for p in g["params"] if p.requires_grad]
        assert [id(p) for p in order] == [id(p) for p in flat_grads.params], "groups must list the parameters in FlatGradients order"
        self.flat_params = FlatParameters(order)
        self.flat_grads = flat_grads
        self.exp_avg = torch.zeros_like(self.flat_params.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat_params.flat)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.step_count = 0
        off = 0
        for name, g in groups.items():
            n = sum(p.numel() for p in g["params"] if p.requires_grad)
            sched = g.get("scheduler")
            # LambdaLR semantics: the factor of step 0 applies from construction (a warm-up schedule starts at lr = 0)
            self.groups[name] = {"start": off, "numel": n, "lr_init": float(g["lr"]), "scheduler": sched,
                                 "lr": float(g["lr"]) * (sched(0) if sched is not None else 1.0)}
            off += n

    def scheduler_step(self):
        """Optimizers.scheduler_step_all (optimizers.py:146-156): LambdaLR semantics, lr = lr_init * factor(number of scheduler steps)."""
        for g in self.groups.values():
            if g["scheduler"] is not None:
                g["lr"] = g["lr_init"] * g["scheduler"](self.step_count)

    def step(self, grad_scale: float = 1.0):
        lib = _lib.load()
        self.step_count += 1
        P, G = self.flat_params.flat, self.flat_grads.flat
        if not P.is_cuda:
            raise _lib.SdfHipError("FusedAdam.step needs HIP device tensors; there is no CPU fallback")
        # Elements that have never carried a gradient (FlatGradients.live_ranges: table rows of hash levels that are still switched
        # off) have zero gradient and zero moments: their update is exactly 0 unless weight decay moves them, so they are skipped.
        live = self.flat_grads.live_ranges() if self.weight_decay == 0.0 else [(0, P.numel())]
        for g in self.groups.values():
            for la, lb in live:
                a, b = max(g["start"], la), min(g["start"] + g["numel"], lb)
                if b <= a:
                    continue
                _lib.check(lib.sdfhip_adam_step(_lib.ptr(P[a:b]), _lib.ptr(G[a:b]), _lib.ptr(self.exp_avg[a:b]),
                                                _lib.ptr(self.exp_avg_sq[a:b]), b - a, g["lr"], self.betas[0], self.betas[1], self.eps,
                                                self.weight_decay, self.step_count, float(grad_scale), _lib.stream()), "adam_step")

    def zero_grad(self):
        self.flat_grads.zero()

    # ---- checkpointing (the trainer's checkpoint holds {"optimizers": {group: optimizer.state_dict()}}, engine/trainer.py:351-360,
    # and Optimizers.load_optimizers (optimizers.py:157-160) restores it: without this a resumed run would restart Adam's bias
    # correction and the warm-up)
    def state_dict(self) -> Dict:
        out = {"step_count": self.step_count, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "groups": {}}
        for name, g in self.groups.items():
            a, n = g["start"], g["numel"]
            out["groups"][name] = {"lr": g["lr"], "lr_init": g["lr_init"], "numel": n,
                                   "exp_avg": self.exp_avg[a:a + n].detach().clone(), "exp_avg_sq": self.exp_avg_sq[a:a + n].detach().clone()}
        return out

    def load_state_dict(self, state: Dict) -> None:
        if set(state["groups"]) != set(self.groups):
            raise KeyError(f"optimizer groups differ: checkpoint {sorted(state['groups'])}, model {sorted(self.groups)}")
        for name, g in self.groups.items():
            sg = state["groups"][name]
            if int(sg["numel"]) != g["numel"]:
                raise ValueError(f"group {name!r}: {sg['numel']} parameters in the checkpoint, {g['numel']} in the model")
            a, n = g["start"], g["numel"]
            self.exp_avg[a:a + n].copy_(sg["exp_avg"])
            self.exp_avg_sq[a:a + n].copy_(sg["exp_avg_sq"])
            g["lr"], g["lr_init"] = float(sg["lr"]), float(sg["lr_init"])
        self.flat_grads.mark_all_live()  # loaded moments may be non-zero anywhere
        self.step_count = int(state["step_count"])
        self.betas, self.eps, self.weight_decay = tuple(state["betas"]), float(state["eps"]), float(state["weight_decay"])


class Optimizers:
    """engine/optimizers.py:93-160: the trainer-facing wrapper (zero_grad_all / optimizer_step_all / scheduler_step_all)."""

    def __init__(self, config: Dict[str, Dict], param_groups: Dict[str, List[torch.nn.Parameter]],
                 flat_grads: Optional[FlatGradients] = None):
        """config[name] = {"lr": float, "scheduler": callable | None, ...}; groups without parameters are skipped."""
        groups = {k: {"params": v, **config[k]} for k, v in param_groups.items() if len(v) > 0}
        if flat_grads is None:
            flat_grads = FlatGradients([p for g in groups.values() for p in g["params"]], buckets=[g["params"] for g in groups.values()])
        self.flat_grads = flat_grads
        self._group_params = {k: list(g["params"]) for k, g in groups.items()}  # full lists (incl. requires_grad=False), as the reference indexes them
        eps = {config[k].get("eps", 1e-15) for k in groups}
        assert len(eps) == 1, "one eps for all groups (the reference uses 1e-15 throughout)"
        self.adam = FusedAdam(groups, flat_grads, eps=eps.pop())

    def zero_grad_all(self):
        self.adam.zero_grad()

    def optimizer_step_all(self, grad_scale: float = 1.0):
        self.adam.step(grad_scale)

    def scheduler_step_all(self, step: int = 0):
        self.adam.scheduler_step()

    def state_dict(self) -> Dict:
        """What the trainer stores under "optimizers" (engine/trainer.py:351-360).  One difference from per-group torch.optim.Adam
        is documented here rather than hidden: the fused step updates EVERY element of a group, also parameters whose gradient
        is None in torch terms (their flat gradient is the zero zero() left, so moments decay and the bias-corrected update is
        0 while they were never used; once a parameter has been used, torch would freeze its moments in steps that skip it,
        this keeps decaying them)."""
        return self.adam.state_dict()

    def load_optimizers(self, loaded_state: Dict) -> None:
        """optimizers.py:157-160.  Accepts this class's own state_dict() and the REFERENCE's checkpoint layout
        {group: torch.optim.Adam.state_dict()} (engine/trainer.py:351-360): per-parameter exp_avg / exp_avg_sq / step, indexed by the
        position of the parameter in the group's parameter list."""
        if "groups" in loaded_state and "step_count" in loaded_state:
            self.adam.load_state_dict(loaded_state)
            return
        if not all(isinstance(v, dict) and "state" in v and "param_groups" in v for v in loaded_state.values()):
            raise ValueError("optimizer state is neither Optimizers.state_dict() of this repo nor {group: torch.optim.Adam.state_dict()} of the "
                             f"reference (top-level keys: {sorted(loaded_state)})")
        if set(loaded_state) != set(self._group_params):
            raise KeyError(f"optimizer groups differ: checkpoint {sorted(loaded_state)}, model {sorted(self._group_params)}")
        adam = self.adam
        steps = set()
        for name, ref in loaded_state.items():
            plist = self._group_params[name]
            for idx, st in ref["state"].items():
                idx = int(idx)
                if idx >= len(plist):
                    raise ValueError(f"group {name!r}: the checkpoint has state for parameter #{idx}, the model's group has {len(plist)} parameters")
                p = plist[idx]
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"group {name!r}, parameter #{idx}: moment of shape {tuple(st['exp_avg'].shape)} for a parameter of shape "
                                     f"{tuple(p.shape)} - the reference keeps its tiny-cuda-nn modules (proposal networks, 'grid' background) as "
                                     "one flat `params` vector each; their optimizer state does not map onto this repo's tensors")
                if id(p) not in adam.flat_params.offset:
                    continue  # requires_grad = False here
                a, n = adam.flat_params.offset[id(p)], p.numel()
                adam.exp_avg[a:a + n].copy_(st["exp_avg"].reshape(-1))
                adam.exp_avg_sq[a:a + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(st["step"]))
            pg = ref["param_groups"][0]
            adam.groups[name]["lr"] = float(pg["lr"])
            adam.groups[name]["lr_init"] = float(pg.get("initial_lr", adam.groups[name]["lr_init"]))
        if len(steps) > 1:
            # torch counts steps per parameter (a parameter unused in some iterations lags); the fused step has ONE counter: take the
            # latest, which is exact for every parameter that was used in every iteration
            pass
        adam.step_count = max(steps) if steps else 0
        adam.flat_grads.mark_all_live()

    load_state_dict = load_optimizers

    def load_schedulers(self, loaded_state: Dict) -> None:
        """optimizers.py (trainer resume path): the schedules here are pure functions of the step count, which load_optimizers restores
        together with every group's current lr - there is no separate scheduler state to load."""
        return None
