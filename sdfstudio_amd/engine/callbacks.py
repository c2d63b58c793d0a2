"""Training callbacks, mirroring nerfstudio/engine/callbacks.py:47-116: the objects a model hands its trainer (Model.get_training_callbacks,
models/base_model.py:95-101) so that the trainer drives the model's step schedules.  Same attributes and the same
run_callback_at_location(step, location) contract as the reference's TrainingCallback, so the reference's trainer loop
(engine/trainer.py:185-206) runs them unchanged; the locations compare by NAME, so the reference's own enum members are accepted."""
from enum import Enum, auto
from inspect import signature
from typing import Callable, Dict, List, Optional, Tuple


class TrainingCallbackLocation(Enum):
    """callbacks.py:47-51."""

    BEFORE_TRAIN_ITERATION = auto()
    AFTER_TRAIN_ITERATION = auto()


class TrainingCallback:
    """callbacks.py:54-116."""

    def __init__(self, where_to_run: List[TrainingCallbackLocation], func: Callable, update_every_num_iters: Optional[int] = None,
                 iters: Optional[Tuple[int, ...]] = None, args: Optional[List] = None, kwargs: Optional[Dict] = None):
        assert "step" in signature(func).parameters.keys(), f"'step: int' must be an argument in the callback function 'func': {func.__name__}"
        self.where_to_run = where_to_run
        self.update_every_num_iters = update_every_num_iters
        self.iters = iters
        self.func = func
        self.args = args if args is not None else []
        self.kwargs = kwargs if kwargs is not None else {}

    def run_callback(self, step: int):
        if self.update_every_num_iters is not None:
            if step % self.update_every_num_iters == 0:
                self.func(*self.args, **self.kwargs, step=step)
        elif self.iters is not None:
            if step in self.iters:
                self.func(*self.args, **self.kwargs, step=step)

    def run_callback_at_location(self, step: int, location) -> None:
        if any(getattr(w, "name", w) == getattr(location, "name", location) for w in self.where_to_run):
            self.run_callback(step=step)
