"""Optimisers and LR schedulers, mirroring nerfstudio/engine/optimizers.py:93-160 and nerfstudio/engine/schedulers.py.

The reference builds one ``torch.optim.Adam`` per parameter group (eps 1e-15; lr 5e-4 for "fields", 1e-2 for
"proposal_networks", method_configs.py:483-500) and a LambdaLR scheduler per group.  Here the parameters, their gradients
and the two Adam moments of ALL groups live in four flat fp32 buffers (parameters become views, like the gradients of
``distributed.FlatGradients``), a group is a contiguous slice, and one step of a group is ONE launch of
``sdfhip_adam_step`` - with the data-parallel mean folded into the gradient read, so the all-reduce can be a plain SUM.
There is no PyTorch fallback (``FusedAdam.step`` raises without the library or on CPU tensors); the formula as plain torch ops
is ``oracle.sdf_path.adam_reference``, the checker the tests compare with.
"""
import dataclasses
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
    order AND offsets as ``FlatGradients`` (``layout``: its offsets and total size; the sharded exchange pads every bucket) - so that
    a parameter group is one slice of parameter, gradient and moment buffers alike."""

    def __init__(self, params: Iterable[torch.nn.Parameter], layout=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        self.offset: Dict[int, int] = {}
        if layout is None:
            off = 0
            for p in self.params:
                self.offset[id(p)] = off
                off += p.numel()
            total = off
        else:
            offsets, total = layout
            self.offset = {id(p): int(offsets[id(p)]) for p in self.params}
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)  # zeros: the padding of the sharded layout holds no parameter
        with torch.no_grad():
            for p in self.params:
                off, n = self.offset[id(p)], p.numel()
                self.flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat[off:off + n].view_as(p)


class FusedAdam:
    """Adam over flat buffers: ``groups`` maps a name to (parameters, lr); one native launch per group and step.

    ``flat_grads``: the ``FlatGradients`` of the same parameters in the same order (its buffer is read directly; with
    ``grad_scale`` the all-reduce may stay a SUM).  The moments start at zero like torch.optim.Adam's."""

    def __init__(self, groups: Dict[str, Dict], flat_grads: FlatGradients, betas=(0.9, 0.999), eps: float = 1e-15,
                 weight_decay: float = 0.0):
        self.groups = {}
        # a group's parameters in BUFFER order (distributed.plan_buckets moves the big tables to the front of their group); the groups
        # themselves must follow each other in the buffer as they are listed
        off_of = flat_grads._offset
        order = [p for g in groups.values() for p in sorted((q for q in g["params"] if q.requires_grad), key=lambda q: off_of[id(q)])]
        assert [id(p) for p in order] == [id(p) for p in flat_grads.params], "groups must list the parameters of FlatGradients, group after group"
        self.flat_params = FlatParameters(order, layout=(flat_grads._offset, flat_grads.flat.numel()))
        self.flat_grads = flat_grads
        # moments: one element per parameter element this rank OWNS - everything, or 1 / W of the buffer under the sharded exchange
        # (distributed.py: owned_slices; a slice's offset in these local buffers is fixed for the run)
        self.sharded = bool(getattr(flat_grads, "shard", False)) and (flat_grads.world > 1 or bool(getattr(flat_grads, "exchanging", False)))
        self.exp_avg = torch.zeros(flat_grads.local_numel(), dtype=torch.float32, device=self.flat_params.flat.device)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.step_count = 0
        self._gathered = None  # sharded mode: (exp_avg, exp_avg_sq) over the whole layout, assembled by gather_moments()
        self.last_elements_visited = 0  # parameter elements the last step() of THIS rank updated
        for name, g in groups.items():
            plist = sorted((p for p in g["params"] if p.requires_grad), key=lambda q: off_of[id(q)])
            n = sum(p.numel() for p in plist)
            start = flat_grads._offset[id(plist[0])]
            end = flat_grads._offset[id(plist[-1])] + plist[-1].numel()  # incl. the padding BETWEEN this group's buckets (zeros: a no-op update)
            sched = g.get("scheduler")
            # LambdaLR semantics: the factor of step 0 applies from construction (a warm-up schedule starts at lr = 0)
            # weight decay per group: torch.optim.Adam's L2 form, or (decoupled) torch.optim.AdamW's - AdamWOptimizerConfig of the reference
            self.groups[name] = {"start": start, "end": end, "numel": n, "lr_init": float(g["lr"]), "scheduler": sched,
                                 "lr": float(g["lr"]) * (sched(0) if sched is not None else 1.0),
                                 "weight_decay": float(g.get("weight_decay", weight_decay)), "decoupled": bool(g.get("decoupled", False))}
        self._cur_decay = (float(weight_decay), False)

    def scheduler_step(self):
        """Optimizers.scheduler_step_all (optimizers.py:146-156): LambdaLR semantics, lr = lr_init * factor(number of scheduler steps)."""
        for g in self.groups.values():
            if g["scheduler"] is not None:
                g["lr"] = g["lr_init"] * g["scheduler"](self.step_count)

    def _step_slice(self, a: int, b: int, loc: int, lr: float, grad_scale: float):
        """One native launch: Adam on flat elements [a, b), whose moments sit at [loc, loc + b - a) of the local moment buffers.
        (The CPU tests of the sharded protocol replace THIS method by the oracle's formula; the product has no CPU path.)"""
        P, G = self.flat_params.flat, self.flat_grads.flat
        if not P.is_cuda:
            raise _lib.SdfHipError("FusedAdam.step needs HIP device tensors; there is no CPU fallback")
        lib = _lib.load()
        n = b - a
        wd, decoupled = self._cur_decay  # of the group being visited (_visit)
        fn = lib.sdfhip_adamw_step if decoupled else lib.sdfhip_adam_step
        _lib.check(fn(_lib.ptr(P[a:b]), _lib.ptr(G[a:b]), _lib.ptr(self.exp_avg[loc:loc + n]), _lib.ptr(self.exp_avg_sq[loc:loc + n]), n, lr,
                      self.betas[0], self.betas[1], self.eps, wd, self.step_count, float(grad_scale), _lib.stream()), "adam_step")

    def _visit(self, lo: int, hi: int, slices, grad_scale: float):
        """Adam on the part of `slices` ([(a, b, local offset)]) that lies inside [lo, hi), group by group (a group = one lr)."""
        for g in self.groups.values():
            self._cur_decay = (g["weight_decay"], g["decoupled"])
            for a, b, loc in slices:
                x, y = max(a, g["start"], lo), min(b, g["end"], hi)
                if y > x:
                    self._step_slice(x, y, loc + (x - a), g["lr"], grad_scale)
                    self.last_elements_visited += y - x

    def step(self, grad_scale: float = 1.0, works=None):
        """All-reduce mode: one launch per group (and live range).  Sharded mode (`works` = FlatGradients.pop_work() after
        finish(wait=False)): chunk by chunk - wait for the chunk's reduce-scatter, update the owned slice, send it back (all-gather,
        asynchronous: FlatGradients.wait_parameters) - so that the first chunks' parameters travel while the later chunks' gradients
        are still arriving."""
        self.step_count += 1
        self.last_elements_visited = 0
        self._gathered = None  # moments assembled for a checkpoint (gather_moments) are stale from here on
        fg = self.flat_grads
        # Elements that have never carried a gradient (FlatGradients.live_ranges: table rows of hash levels that are still switched
        # off) have zero gradient and zero moments: their update is exactly 0 unless weight decay moves them, so they are skipped.
        if all(g["weight_decay"] == 0.0 for g in self.groups.values()):
            slices = fg.owned_live()
        else:
            slices = fg.owned_slices()
        if not self.sharded:
            for w, _, _ in (works or []):
                fg.wait_chunk(w)
            self._visit(0, self.flat_params.flat.numel(), slices, grad_scale)
            return
        travelled = {(a, b): w for w, a, b in (works or [])}
        for a, b in fg.gather_chunks():  # buffer order = launch order of the reduce-scatters: chunk k's Adam runs beside chunk k + 1's arrival
            w = travelled.pop((a, b), None)
            if w is not None:
                fg.wait_chunk(w)
            self._visit(a, b, slices, grad_scale)
        for w in travelled.values():  # chunks outside the live ranges cannot have travelled; be safe
            fg.wait_chunk(w)
        # the updated slices go back in the order the next step reads them (one communicator = one queue: a gather issued between two
        # reduce-scatters would only wait behind them)
        fg.gather_parameters(self.flat_params.flat)

    def zero_grad(self):
        self.flat_grads.zero()

    # ---- checkpointing (the trainer's checkpoint holds {"optimizers": {group: optimizer.state_dict()}}, engine/trainer.py:351-360,
    # and Optimizers.load_optimizers (optimizers.py:157-160) restores it: without this a resumed run would restart Adam's bias
    # correction and the warm-up)
    def gather_moments(self) -> None:
        """Sharded mode: assemble (exp_avg, exp_avg_sq) over the WHOLE flat layout on every rank - every rank contributes its owned slices,
        one all-gather per grid chunk (2 x the buffer in memory until the next step() drops it).  A COLLECTIVE: EVERY rank must call it, at
        the same point of the step.  The reference saves checkpoints on the main process only (engine/trainer.py:277 @check_main_thread),
        so the call belongs OUTSIDE that guard; state_dict() itself is local and raises when the moments have not been gathered.
        All-reduce mode: nothing to do."""
        if self.sharded:
            self._gathered = self._full_moments()

    def _full_moments(self):
        fg = self.flat_grads
        if not self.sharded:
            return self.exp_avg, self.exp_avg_sq
        import torch.distributed as dist

        full = [torch.zeros_like(self.flat_params.flat) for _ in range(2)]
        for src, dst in zip((self.exp_avg, self.exp_avg_sq), full):
            for (a, b, loc) in fg.owned_slices():
                n = b - a
                ca = a - fg.rank * n  # the chunk this slice belongs to starts `rank` slices earlier
                dist.all_gather([dst[ca + r * n:ca + (r + 1) * n] for r in range(fg.world)], src[loc:loc + n].clone(), group=fg.group)
        return full[0], full[1]

    def state_dict(self) -> Dict:
        out = {"step_count": self.step_count, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "groups": {}}
        if self.sharded:
            if self._gathered is None:
                raise RuntimeError("FusedAdam.state_dict() under the sharded exchange: this rank holds 1 / W of the moments.  Call "
                                   "gather_moments() on EVERY rank first (a collective - outside the reference's main-process-only checkpoint "
                                   "guard, engine/trainer.py:277), then state_dict() on the rank that writes")
            m, v = self._gathered
        else:
            m, v = self.exp_avg, self.exp_avg_sq
        for name, g in self.groups.items():
            a, b = g["start"], g["end"]
            out["groups"][name] = {"lr": g["lr"], "lr_init": g["lr_init"], "numel": g["numel"], "span": b - a,
                                   "exp_avg": m[a:b].detach().clone(), "exp_avg_sq": v[a:b].detach().clone()}
        return out

    def _store_moments(self, a: int, b: int, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor) -> None:
        """Write moments for flat elements [a, b) (given as tensors of b - a elements): the part of them this rank owns."""
        for oa, ob, loc in self.flat_grads.owned_slices():
            x, y = max(a, oa), min(b, ob)
            if y > x:
                self.exp_avg[loc + (x - oa):loc + (y - oa)].copy_(exp_avg.reshape(-1)[x - a:y - a])
                self.exp_avg_sq[loc + (x - oa):loc + (y - oa)].copy_(exp_avg_sq.reshape(-1)[x - a:y - a])

    def load_state_dict(self, state: Dict) -> None:
        if set(state["groups"]) != set(self.groups):
            raise KeyError(f"optimizer groups differ: checkpoint {sorted(state['groups'])}, model {sorted(self.groups)}")
        for name, g in self.groups.items():
            sg = state["groups"][name]
            if int(sg["numel"]) != g["numel"]:
                raise ValueError(f"group {name!r}: {sg['numel']} parameters in the checkpoint, {g['numel']} in the model")
            if int(sg.get("span", sg["numel"])) != g["end"] - g["start"]:
                raise ValueError(f"group {name!r}: the checkpoint's moments cover {sg.get('span', sg['numel'])} buffer elements, this layout "
                                 f"{g['end'] - g['start']} (the sharded exchange pads every bucket to 64 x world size: save and load with the same "
                                 "world size and shard setting, or go through the reference's per-parameter layout)")
        self.flat_grads.mark_all_live()  # loaded moments may be non-zero anywhere
        for name, g in self.groups.items():
            sg = state["groups"][name]
            self._store_moments(g["start"], g["end"], sg["exp_avg"], sg["exp_avg_sq"])
            g["lr"], g["lr_init"] = float(sg["lr"]), float(sg["lr_init"])
        self.step_count = int(state["step_count"])
        self.betas, self.eps, self.weight_decay = tuple(state["betas"]), float(state["eps"]), float(state["weight_decay"])


# ---- the reference's optimiser / scheduler CONFIG objects (engine/optimizers.py:30-71, engine/schedulers.py:118-222) as plain data: what the
# entries of a method config's `optimizers` dictionary are made of (configs/method_configs.py).  Optimizers() converts them
# (group_config_from_reference below); the reference's own objects are accepted just the same.
@dataclasses.dataclass
class AdamOptimizerConfig:
    lr: float = 0.0005
    eps: float = 1e-08
    weight_decay: float = 0


@dataclasses.dataclass
class AdamWOptimizerConfig:
    lr: float = 0.0005
    eps: float = 1e-08
    weight_decay: float = 0


@dataclasses.dataclass
class MultiStepSchedulerConfig:
    max_steps: int = 1000000


@dataclasses.dataclass
class ExponentialSchedulerConfig:
    decay_rate: float = 0.1
    max_steps: int = 1000000


@dataclasses.dataclass
class NeuSSchedulerConfig:
    warm_up_end: int = 5000
    learning_rate_alpha: float = 0.05
    max_steps: int = 300000


@dataclasses.dataclass
class MultiStepWarmupSchedulerConfig:
    warm_up_end: int = 5000
    milestones: List[int] = dataclasses.field(default_factory=lambda: [300000, 400000, 500000])
    gamma: float = 0.33


def group_config_from_reference(entry: Dict) -> Dict:
    """One parameter group of the reference's optimizer dictionary (configs/method_configs.py: {"optimizer": AdamOptimizerConfig(lr, eps,
    weight_decay), "scheduler": NeuSSchedulerConfig(...) | MultiStepSchedulerConfig(...) | ... | None}; engine/optimizers.py:30-71,
    engine/schedulers.py:118-222) -> this module's {"lr", "eps", "weight_decay", "scheduler": step -> factor}.  The config objects are read by
    attribute (duck-typed: the reference's classes or anything with the same fields); entries already in this module's form pass through."""
    if "optimizer" not in entry:
        return entry
    opt = entry["optimizer"]
    kind = getattr(getattr(opt, "_target", None), "__name__", type(opt).__name__)
    wd = float(getattr(opt, "weight_decay", 0.0) or 0.0)
    if "RAdam" in kind:
        raise NotImplementedError("RAdamOptimizerConfig: the fused optimiser is Adam (no surface preset of the reference uses RAdam)")
    out = {"lr": float(opt.lr), "eps": float(getattr(opt, "eps", 1e-8)), "weight_decay": wd, "decoupled": "AdamW" in kind}  # AdamW: sdfhip_adamw_step
    sch = entry.get("scheduler")
    if sch is None or callable(sch):
        out["scheduler"] = sch
        return out
    name = type(sch).__name__
    if name.startswith("NeuSScheduler"):
        out["scheduler"] = neus_scheduler(sch.warm_up_end, sch.learning_rate_alpha, sch.max_steps)
    elif name.startswith("MultiStepWarmupScheduler"):
        out["scheduler"] = multi_step_warmup_scheduler(sch.warm_up_end, tuple(sch.milestones), sch.gamma)
    elif name.startswith("MultiStepScheduler"):
        out["scheduler"] = multi_step_scheduler(sch.max_steps)
    elif name.startswith("ExponentialScheduler"):
        out["scheduler"] = exponential_decay_scheduler(sch.decay_rate, sch.max_steps)
    else:
        raise NotImplementedError(f"scheduler config {name}: built are NeuS, MultiStep, MultiStepWarmup and Exponential (engine/schedulers.py)")
    return out


class Optimizers:
    """engine/optimizers.py:93-160: the trainer-facing wrapper (zero_grad_all / optimizer_step_all / scheduler_step_all)."""

    def __init__(self, config: Dict[str, Dict], param_groups: Dict[str, List[torch.nn.Parameter]],
                 flat_grads: Optional[FlatGradients] = None, shard: bool = False):
        """config[name] = {"lr": float, "scheduler": callable | None, ...} or the reference's {"optimizer": AdamOptimizerConfig, "scheduler":
        SchedulerConfig | None} (group_config_from_reference); groups without parameters are skipped.
        shard (when no FlatGradients is handed in): the sharded exchange of distributed.py."""
        config = {k: group_config_from_reference(v) for k, v in config.items()}  # the reference's {"optimizer": ..., "scheduler": ...} entries
        groups = {k: {"params": v, **config[k]} for k, v in param_groups.items() if len(v) > 0}
        if flat_grads is None:
            flat_grads = FlatGradients([p for g in groups.values() for p in g["params"]], buckets=[g["params"] for g in groups.values()],
                                       shard=shard)
        self.flat_grads = flat_grads
        self._group_params = {k: list(g["params"]) for k, g in groups.items()}  # full lists (incl. requires_grad=False), as the reference indexes them
        eps = {config[k].get("eps", 1e-15) for k in groups}
        assert len(eps) == 1, "one eps for all groups (the reference uses 1e-15 throughout)"
        self.adam = FusedAdam(groups, flat_grads, eps=eps.pop())  # weight decay (L2 or decoupled) is a property of each group

    def zero_grad_all(self):
        self.adam.zero_grad()

    def optimizer_step_all(self, grad_scale: Optional[float] = 1.0):
        """grad_scale=None: this call ALSO closes the gradient exchange - `flat_grads.finish(average=False, wait=False)` and the chunk-wise
        waits of FusedAdam.step (the only form the sharded exchange takes; valid for the all-reduce exchange too)."""
        if grad_scale is None:
            grad_scale = self.flat_grads.finish(average=False, wait=False)
            self.adam.step(grad_scale, works=self.flat_grads.pop_work())
        else:
            if self.adam.sharded:
                raise RuntimeError("sharded exchange: call optimizer_step_all(grad_scale=None) - the step waits for the reduce-scatters chunk "
                                   "by chunk and sends the updated slices back")
            self.adam.step(grad_scale)

    def optimizer_scaler_step_all(self, grad_scaler) -> None:
        """engine/optimizers.py:136-143 - what the reference's trainer calls every iteration (engine/trainer.py:320-324), with a GradScaler
        that is DISABLED unless mixed precision is on (no surface preset turns it on: configs/method_configs.py `mixed_precision=False`).
        A disabled scaler's step(optimizer) is optimizer.step(); an enabled one would have to unscale and inf-check inside the fused
        step, which is not built (the path computes in fp32: there is nothing to scale)."""
        if grad_scaler is not None and getattr(grad_scaler, "is_enabled", lambda: False)():
            raise NotImplementedError("Optimizers.optimizer_scaler_step_all: an ENABLED GradScaler (mixed precision) is not built - the native "
                                      "path trains in fp32; run with mixed_precision=False as every surface preset of the reference does")
        self.optimizer_step_all(grad_scale=None if self.adam.sharded else 1.0)

    def optimizer_step(self, param_group_name: str) -> None:
        """engine/optimizers.py:112-118 (one group's optimiser).  The fused Adam steps every group in ONE launch over the flat buffers."""
        raise NotImplementedError("Optimizers.optimizer_step(param_group_name): the fused Adam steps all groups at once - optimizer_step_all()")

    def scheduler_step(self, param_group_name: str) -> None:
        """engine/optimizers.py:120-127 (one group's scheduler); here the groups' schedules advance together - scheduler_step_all()."""
        raise NotImplementedError("Optimizers.scheduler_step(param_group_name): the schedules advance together - scheduler_step_all(step)")

    def wait_parameters(self, late: bool = True):
        """Sharded exchange: the parameters updated by the last step are complete on this rank after this (stream wait).  Call it before
        the next forward reads them; late=False leaves the late buckets (the big table, FlatGradients(late_buckets=...)) in flight - a
        second call before the field's forward collects them, after the proposal sampling has been enqueued."""
        self.flat_grads.wait_parameters(late)

    def scheduler_step_all(self, step: int = 0):
        self.adam.scheduler_step()

    def gather_moments(self) -> None:
        """Sharded exchange only: the COLLECTIVE half of checkpointing (FusedAdam.gather_moments) - call on every rank before the rank that
        writes the checkpoint calls state_dict().  A no-op under the all-reduce exchange and on one GPU."""
        self.adam.gather_moments()

    def state_dict(self) -> Dict:
        """What the trainer stores under "optimizers" (engine/trainer.py:351-360).  LOCAL: under the sharded exchange it raises unless
        gather_moments() ran on every rank since the last step (a state_dict() that gathered by itself would deadlock the reference's
        rank-0-only save_checkpoint).  One difference from per-group torch.optim.Adam
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
        # The reference builds an optimizer for EVERY key of get_param_groups(), also for its placeholder groups: with
        # background_model = "none" "field_background" is [Parameter(ones(1))] (base_surface_model.py:241-244; it never receives a
        # gradient, so its `state` is empty), and Optimizers here drops empty groups.  A checkpoint group the model does not have is
        # therefore accepted when it carries no moments (or only 0-d / single-element dummies); a model group the checkpoint lacks is not.
        def _is_placeholder(ref: Dict) -> bool:
            return all(st["exp_avg"].numel() <= 1 for st in ref["state"].values())

        extra = [k for k in loaded_state if k not in self._group_params]
        unknown = [k for k in extra if not _is_placeholder(loaded_state[k])]
        missing = [k for k in self._group_params if k not in loaded_state]
        if unknown or missing:
            raise KeyError(f"optimizer groups differ: checkpoint {sorted(loaded_state)}, model {sorted(self._group_params)}"
                           + (f"; checkpoint groups with moments the model has no parameters for: {unknown}" if unknown else "")
                           + (f"; model groups missing from the checkpoint: {missing}" if missing else ""))
        adam = self.adam
        # pass 1: validate everything (indices, shapes, hyper-parameters) before a single element is written
        plan, steps = [], set()
        for name, ref in loaded_state.items():
            if name in extra:
                continue
            plist = self._group_params[name]
            for idx, st in ref["state"].items():
                idx = int(idx)
                if idx >= len(plist):
                    raise ValueError(f"group {name!r}: the checkpoint has state for parameter #{idx}, the model's group has {len(plist)} parameters")
                p = plist[idx]
                if tuple(st["exp_avg"].shape) != tuple(p.shape) or tuple(st["exp_avg_sq"].shape) != tuple(p.shape):
                    raise ValueError(f"group {name!r}, parameter #{idx}: moment of shape {tuple(st['exp_avg'].shape)} for a parameter of shape "
                                     f"{tuple(p.shape)} - the reference keeps its tiny-cuda-nn modules (proposal networks, 'grid' background) as "
                                     "one flat `params` vector each; their optimizer state does not map onto this repo's tensors")
                steps.add(int(st["step"]))
                if id(p) in adam.flat_params.offset:  # else: requires_grad = False here
                    plan.append((adam.flat_params.offset[id(p)], p.numel(), st))
            pg = ref["param_groups"][0]
            gwd = float(adam.groups[name]["weight_decay"])
            betas, eps, wd = tuple(pg.get("betas", adam.betas)), float(pg.get("eps", adam.eps)), float(pg.get("weight_decay", gwd))
            if tuple(float(b) for b in betas) != tuple(float(b) for b in adam.betas) or eps != float(adam.eps) or wd != gwd:
                raise ValueError(f"group {name!r}: the checkpoint was trained with betas {betas}, eps {eps}, weight_decay {wd}; this optimizer is "
                                 f"configured with betas {tuple(adam.betas)}, eps {adam.eps}, weight_decay {gwd} (betas and eps: one setting for "
                                 "all groups, FusedAdam)")
        # pass 2: copy.  Loaded moments may be non-zero anywhere: every row is live from here on
        adam.flat_grads.mark_all_live()
        for a0, n, st in plan:
            adam._store_moments(a0, a0 + n, st["exp_avg"], st["exp_avg_sq"])
        for name, ref in loaded_state.items():
            if name in extra:
                continue
            pg = ref["param_groups"][0]
            adam.groups[name]["lr"] = float(pg["lr"])
            adam.groups[name]["lr_init"] = float(pg.get("initial_lr", adam.groups[name]["lr_init"]))
        # torch counts steps per parameter (a parameter unused in some iterations lags); the fused step has ONE counter: the latest,
        # which is exact for every parameter that was used in every iteration
        adam.step_count = max(steps) if steps else 0

    load_state_dict = load_optimizers

    def load_schedulers(self, loaded_state: Dict) -> None:
        """optimizers.py (trainer resume path): the schedules here are pure functions of the step count, which load_optimizers restores
        together with every group's current lr - there is no separate scheduler state to load."""
        return None
