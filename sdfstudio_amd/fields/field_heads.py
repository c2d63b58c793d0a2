"""Dictionary keys of the field boundary (nerfstudio/field_components/field_heads.py:28-44).

The reference's models index a field's output dictionary with ITS enum (``field_outputs[FieldHeadNames.RGB]``) and members of
two different Enum classes never compare equal by default.  For the drop-in deployment (our field behind the reference's
SurfaceModel) the mirror's members therefore hash like the reference's (Enum hashes its member NAME)
and compare equal to the same-named member of any enum called ``FieldHeadNames``: a dictionary built with either enum can
be read with the other, whichever module was imported first."""
from enum import Enum


class FieldHeadNames(Enum):
    RGB = "rgb"
    SH = "sh"
    DENSITY = "density"
    NORMALS = "normals"
    PRED_NORMALS = "pred_normals"
    UNCERTAINTY = "uncertainty"
    TRANSIENT_RGB = "transient_rgb"
    TRANSIENT_DENSITY = "transient_density"
    SEMANTICS = "semantics"
    NORMAL = "normal"
    SDF = "sdf"
    ALPHA = "alpha"
    GRADIENT = "gradient"
    OCCUPANCY = "occupancy"

    def __eq__(self, other):
        if self is other:
            return True
        return isinstance(other, Enum) and type(other).__name__ == "FieldHeadNames" and other.name == self.name and other.value == self.value

    __hash__ = Enum.__hash__
