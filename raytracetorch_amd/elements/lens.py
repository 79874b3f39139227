"""Refractive lens elements (reference: elements/lens.py:13-429).

Each constructor builds the lens geometry and wires *shared* index-of-refraction Parameters into
the per-surface Snell functors, reproducing the reference's wiring exactly (including which index
is called `ior_in` on which face): see `_snell`.
"""
from typing import Optional

import torch
import torch.nn as nn

from ..geom import CylSinglet, Doublet, RayTransform, Singlet, Triplet
from ..phys import Block, RefractFresnel, RefractSnell
from .parent import Element


def _snell(ior_in: nn.Parameter, ior_out: nn.Parameter, fresnel: bool = False) -> RefractSnell:
    """Functor whose index attributes are the given Parameter objects themselves; `fresnel` selects the
    stochastic RefractFresnel instead of RefractSnell (elements/lens.py:36-49)."""
    sf = (RefractFresnel if fresnel else RefractSnell)(ior_in=1.0, ior_out=1.0)
    sf.ior_in = ior_in
    sf.ior_out = ior_out
    return sf


def _ior(v, grad):
    return nn.Parameter(torch.as_tensor(v, dtype=torch.float32), requires_grad=grad)


def _vertex_chain(el, n, iors):
    from .ideal import ParaxialRefractMat
    zs = [el.shape.z + el.shape.surfaces[i].z for i in range(n)]
    return zs, [el._posed(ParaxialRefractMat(zs[i], zs[i], iors[i], iors[i + 1])) for i in range(n)]


class SingletLens(Element):
    """Faces: Snell(in=glass, out=media), Snell(in=media, out=glass); edge: Block when `inked`,
    else Snell(in=glass, out=media) (elements/lens.py:13-58)."""

    def __init__(self, c1: float, c2: float, d: float, t: float, ior_glass: float, ior_media: float = 1.0,
                 c1_grad: bool = False, c2_grad: bool = False, t_grad: bool = False, d_grad: bool = False,
                 ior_glass_grad: bool = False, ior_media_grad: bool = False,
                 fresnel: bool = False, inked: bool = False, transform: Optional[RayTransform] = None):
        super().__init__()
        self.ior_glass = _ior(ior_glass, ior_glass_grad)
        self.ior_media = _ior(ior_media, ior_media_grad)
        self.shape = Singlet(C1=c1, C2=c2, D=d, T=t, C1_grad=c1_grad, C2_grad=c2_grad, D_grad=d_grad,
                             T_grad=t_grad, transform=transform)
        self.surface_functions.append(_snell(self.ior_glass, self.ior_media, fresnel))
        self.surface_functions.append(_snell(self.ior_media, self.ior_glass, fresnel))
        self.surface_functions.append(Block() if inked else _snell(self.ior_glass, self.ior_media, fresnel))

    # paraxial conveniences (elements/lens.py:60-127)
    @property
    def power1(self):
        return self.shape.surfaces[0].c * (self.ior_glass - self.ior_media)

    @property
    def power2(self):
        return self.shape.surfaces[1].c * (self.ior_media - self.ior_glass)

    @property
    def Power(self):
        return self.power1 + self.power2 - self.power1 * self.power2 * (self.T / self.ior_glass)

    @property
    def f(self):
        return 1 / self.Power

    @property
    def f_bfl(self):
        return self.f * (1 - self.T * self.power1 / self.ior_glass)

    @property
    def f_ffl(self):
        return -self.f * (1 - self.T * self.power2 / self.ior_glass)

    @property
    def R1(self):
        return 1 / self.shape.surfaces[0].c

    @property
    def R2(self):
        return -1 / self.shape.surfaces[1].c

    @property
    def T(self):
        return self.shape.T

    @property
    def T_edge(self):
        return self.shape.T_edge

    @property
    def P1z(self):
        """z of the first principal plane in the element frame: front vertex + h1, h1 = -f (n - n0) T c2 / n
        (elements/lens.py:112-119)."""
        h1 = -self.f * (self.ior_glass - self.ior_media) * self.T * self.shape.surfaces[1].c / self.ior_glass
        return self.shape.surfaces[0].z + h1

    @property
    def P2z(self):
        """z of the second principal plane: back vertex + h2, h2 = -f (n - n0) T c1 / n (elements/lens.py:121-127)."""
        h2 = -self.f * (self.ior_glass - self.ior_media) * self.T * self.shape.surfaces[0].c / self.ior_glass
        return self.shape.surfaces[1].z + h2

    def Bend(self, delta_c):
        """Declared in the reference to trade c1 against c2 at constant power (elements/lens.py:150-183); there it
        stops at its first access of `self.shape.C1`, an attribute no lens shape has, so there is no behaviour beyond
        that AttributeError to reproduce."""
        raise AttributeError(f"'{type(self.shape).__name__}' object has no attribute 'C1'")


    def _face_matrices(self, cx_of):
        from .ideal import ParaxialRefractMat
        s0, s1 = self.shape.surfaces[0], self.shape.surfaces[1]
        zs = [self.shape.z + s0.z, self.shape.z + s1.z]
        m1 = ParaxialRefractMat(cx_of(s0.c), s0.c, self.ior_media, self.ior_glass)
        m2 = ParaxialRefractMat(cx_of(s1.c), s1.c, self.ior_glass, self.ior_media)
        return zs, [self._posed(m1), self._posed(m2)]

    def getParaxial(self):
        """Refraction matrices of the two faces at their vertex z (elements/lens.py:129-148)."""
        return self._face_matrices(lambda c: c)


class CylSingletLens(SingletLens):
    """Cylindrical singlet: the singlet's three functors with the edge functor repeated for the
    four side planes (elements/lens.py:185-208)."""

    def __init__(self, c1, c2, height, width, t, ior_glass, ior_media=1.0,
                 c1_grad=False, c2_grad=False, t_grad=False, height_grad=False, width_grad=False,
                 ior_glass_grad=False, ior_media_grad=False, fresnel=False, inked=False,
                 transform: Optional[RayTransform] = None):
        super().__init__(c1, c2, height, t, ior_glass, ior_media=ior_media,
                         ior_glass_grad=ior_glass_grad, ior_media_grad=ior_media_grad,
                         fresnel=fresnel, inked=inked, transform=None)
        self.shape = CylSinglet(C1=c1, C2=c2, height=height, width=width, T=t, C1_grad=c1_grad,
                                C2_grad=c2_grad, h_grad=height_grad, w_grad=width_grad, T_grad=t_grad,
                                transform=transform)
        for _ in range(3):
            self.surface_functions.append(self.surface_functions[-1])
        self.Nsurfaces = 6

    def getParaxial(self):
        """Power in y only (elements/lens.py:210-227)."""
        return self._face_matrices(lambda c: 0.0 * c)


class DoubletLens(Element):
    """Snell(media->glass1), Snell(glass1->glass2), Snell(glass2->media), two Block edges
    (elements/lens.py:231-280; `inked` is accepted and ignored there as well)."""

    def __init__(self, c1, c2, c3, d, t1, t2, ior_glass1, ior_glass2, ior_media=1.0,
                 c1_grad=False, c2_grad=False, c3_grad=False, t1_grad=False, t2_grad=False, d_grad=False,
                 ior_glass1_grad=False, ior_glass2_grad=False, ior_media_grad=False,
                 fresnel=False, inked=True, transform: Optional[RayTransform] = None):
        super().__init__()
        self.ior_glass1 = _ior(ior_glass1, ior_glass1_grad)
        self.ior_glass2 = _ior(ior_glass2, ior_glass2_grad)
        self.ior_media = _ior(ior_media, ior_media_grad)
        self.shape = Doublet(C1=c1, C2=c2, C3=c3, D=d, T1=t1, T2=t2, C1_grad=c1_grad, C2_grad=c2_grad,
                             C3_grad=c3_grad, D_grad=d_grad, T1_grad=t1_grad, T2_grad=t2_grad,
                             transform=transform)
        chain = [self.ior_media, self.ior_glass1, self.ior_glass2, self.ior_media]
        for a, b in zip(chain[:-1], chain[1:]):
            self.surface_functions.append(_snell(a, b, fresnel))
        for _ in range(2):
            self.surface_functions.append(Block())

    T1 = property(lambda self: self.shape.T1)
    T2 = property(lambda self: self.shape.T2)
    # radii as the reference reports them (elements/lens.py:283-294: the sign of the last one flipped)
    R1 = property(lambda self: 1.0 / self.shape.surfaces[0].c)
    R2 = property(lambda self: 1.0 / self.shape.surfaces[1].c)
    R3 = property(lambda self: -1.0 / self.shape.surfaces[2].c)

    def getParaxial(self):
        """One refraction matrix per optical surface at its vertex z.  As in the reference, the value handed to the
        matrix as curvature is the vertex z of the surface (elements/lens.py:303-322, :410-429)."""
        return _vertex_chain(self, 3, [self.ior_media, self.ior_glass1, self.ior_glass2, self.ior_media])


class TripletLens(Element):
    """Four Snell faces media->g1->g2->g3->media and three Block edges (elements/lens.py:325-384)."""

    def __init__(self, c1, c2, c3, c4, d, t1, t2, t3, ior_glass1, ior_glass2, ior_glass3, ior_media=1.0,
                 c1_grad=False, c2_grad=False, c3_grad=False, c4_grad=False,
                 t1_grad=False, t2_grad=False, t3_grad=False, d_grad=False,
                 ior_glass1_grad=False, ior_glass2_grad=False, ior_glass3_grad=False, ior_media_grad=False,
                 fresnel=False, inked=True, transform: Optional[RayTransform] = None):
        super().__init__()
        self.ior_glass1 = _ior(ior_glass1, ior_glass1_grad)
        self.ior_glass2 = _ior(ior_glass2, ior_glass2_grad)
        self.ior_glass3 = _ior(ior_glass3, ior_glass3_grad)
        self.ior_media = _ior(ior_media, ior_media_grad)
        self.shape = Triplet(C1=c1, C2=c2, C3=c3, C4=c4, D=d, T1=t1, T2=t2, T3=t3,
                             C1_grad=c1_grad, C2_grad=c2_grad, C3_grad=c3_grad, C4_grad=c4_grad,
                             D_grad=d_grad, T1_grad=t1_grad, T2_grad=t2_grad, T3_grad=t3_grad,
                             transform=transform)
        chain = [self.ior_media, self.ior_glass1, self.ior_glass2, self.ior_glass3, self.ior_media]
        for a, b in zip(chain[:-1], chain[1:]):
            self.surface_functions.append(_snell(a, b, fresnel))
        for _ in range(3):
            self.surface_functions.append(Block())

    T1 = property(lambda self: self.shape.T1)
    T2 = property(lambda self: self.shape.T2)
    T3 = property(lambda self: self.shape.T3)
    # elements/lens.py:387-398: the last two with the sign flipped
    R1 = property(lambda self: 1.0 / self.shape.surfaces[0].c)
    R2 = property(lambda self: 1.0 / self.shape.surfaces[1].c)
    R3 = property(lambda self: -1.0 / self.shape.surfaces[2].c)
    R4 = property(lambda self: -1.0 / self.shape.surfaces[3].c)

    def getParaxial(self):
        """One refraction matrix per optical surface at its vertex z.  As in the reference, the value handed to the
        matrix as curvature is the vertex z of the surface (elements/lens.py:303-322, :410-429)."""
        return _vertex_chain(self, 4, [self.ior_media, self.ior_glass1, self.ior_glass2, self.ior_glass3, self.ior_media])

