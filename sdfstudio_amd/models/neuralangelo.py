"""Neuralangelo model, mirroring nerfstudio/models/neuralangelo.py (NeuralangeloModelConfig :39-61, NeuralangeloModel :64-180): the NeuS
model (hierarchical up-sampling, models/neus.py) on a numerical-gradient hash-grid field with three step schedules - the tap distance of
the numerical gradients, the progressive hash levels, the curvature-loss weight - and the curvature loss.  The `neuralangelo` preset
(configs/method_configs.py:184-243) is this model on the field shape of BASELINE config 5 (16 x 8 x 2^22 linear grid, 1 x 256 + 4 x 256).
Host glue only; every stage is a native call (models/neus.py, fields/sdf_field.py numerical branch)."""
from dataclasses import dataclass, field
from typing import Dict, NamedTuple, Type

import torch

from sdfstudio_amd.models.neus import NeuSModel, NeuSModelConfig


@dataclass
class NeuralangeloModelConfig(NeuSModelConfig):
    """models/neuralangelo.py:39-61 (same names, same defaults)."""

    _target: Type = field(default_factory=lambda: NeuralangeloModel)
    enable_progressive_hash_encoding: bool = True
    enable_numerical_gradients_schedule: bool = True
    enable_curvature_loss_schedule: bool = True
    curvature_loss_multi: float = 5e-4
    curvature_loss_warmup_steps: int = 5000
    level_init: int = 4
    steps_per_level: int = 5000


class NeuralangeloSchedule(NamedTuple):
    delta: float             # argument of SDFField.set_numerical_gradients_delta (None: schedule off)
    level: int               # argument of SDFField.update_mask (None: schedule off)
    curvature_factor: float  # NeuralangeloModel.curvature_loss_multi_factor


def neuralangelo_schedule(step: int, config, base_res: float, max_res: float, growth_factor: float) -> NeuralangeloSchedule:
    """The three BEFORE_TRAIN_ITERATION callbacks of models/neuralangelo.py:75-150 as one pure function of the step (what they leave in
    the field and the model).  NOT the neus-facto-angelo formulas (models/neus_facto.py:222-282: x 4, floors 1 / (4 max_res) and
    1 / (10 max_res)): here the delta is doubled (:98) and both floors are 1 / max_res (:97, :138)."""
    spl = config.steps_per_level
    delta = None
    if config.enable_numerical_gradients_schedule:  # :94-106
        delta = max(1.0 / max_res, 1.0 / (base_res * growth_factor ** (step / spl))) * 2.0
    level = None
    if config.enable_progressive_hash_encoding:  # :109-122
        level = max(int(step / spl) + 1, config.level_init)
    factor = 1.0  # populate_modules (:71)
    if config.enable_curvature_loss_schedule:  # :126-145: linear warm-up, then decay with the delta
        if step < config.curvature_loss_warmup_steps:
            factor = step / config.curvature_loss_warmup_steps
        else:
            d = max(1.0 / max_res, 1.0 / (base_res * growth_factor ** ((step - config.curvature_loss_warmup_steps) / spl)))
            factor = d / (1.0 / base_res)
    return NeuralangeloSchedule(delta, level, factor)


class NeuralangeloModel(NeuSModel):
    """models/neuralangelo.py:64-180."""

    def populate_modules(self):
        super().populate_modules()
        self.curvature_loss_multi_factor = 1.0  # :71

    def before_train_iteration(self, step: int):
        """neus.py:80-84 (cos anneal) + neuralangelo.py:94-145, in the reference's callback order."""
        super().before_train_iteration(step)
        f = self.field
        s = neuralangelo_schedule(step, self.config, f.base_res, f.max_res, f.growth_factor)
        if s.delta is not None:
            f.set_numerical_gradients_delta(s.delta)
        if s.level is not None:
            f.update_mask(s.level)
        self.curvature_loss_multi_factor = s.curvature_factor

    def get_metrics_dict(self, outputs, batch) -> Dict[str, torch.Tensor]:
        m = super().get_metrics_dict(outputs, batch)
        if self.training:  # :152-161
            m["activated_encoding"] = self.field.hash_encoding_mask.mean().item()
            m["numerical_gradients_delta"] = self.field.numerical_gradients_delta
            m["curvature_loss_multi"] = self.curvature_loss_multi_factor * self.config.curvature_loss_multi
        return m
    # get_loss_dict: NeuSModel's, which adds the curvature loss (:163-178) inside the fused loss operator whenever curvature_loss_multi > 0
    # and the field evaluated its numerical-gradient taps (field_outputs["sampled_sdf"])
