"""Surfaces the reference declares without an implementation (reference: geom/complex.py:6-11)."""
from .primitives import Surface


class Aspheric(Surface):
    """Declared and refused in the reference as well (geom/complex.py:7-11)."""

    def __init__(self, Coefficients, transform=None):
        super().__init__(transform=transform)
        raise NotImplementedError("Aspheric Not Implemented")
