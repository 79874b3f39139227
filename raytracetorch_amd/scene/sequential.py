"""Fixed-order scene: rays visit every surface of every element in list order
(reference: scene/sequential.py:6-105).

``simulate(rays)`` is ONE kernel launch: each thread carries its ray through the whole surface
program in registers, so HBM sees one read and one write of the ray state per trace instead of a
gather / scatter per surface visit.
"""
import torch.nn as nn

from .. import kernels as K
from ..rays.ray import Rays, SampledRays
from .base import Scene


class SequentialScene(Scene):

    def __init__(self, elements):
        super().__init__()
        self.elements = nn.ModuleList(elements)

    def simulate(self, rays, out=None):
        """Propagate `rays` through all surfaces; mutates and returns `rays` like the reference
        (scene/sequential.py:12-36).  With `out` (a second Rays of the same size) the result goes
        there and `rays` is left untouched."""
        prog = self._program()
        if type(rays) is SampledRays:
            # a fresh Bundle.sample whose planes nobody has read: generate and trace in one launch (33 B written per
            # ray, nothing read) when the result goes back into `rays` and nothing is to be differentiated
            pend = rays._pending()
            if out is None and K.can_generate_in_trace(pend, prog) and not self._needs_grad():
                rec = K.sample_trace_sequential(rays, prog)
                if rec is not None:
                    self._deliver_sensor_visits(prog, rays, rec)
                return rays
            rays._force()   # plain Rays from here on
        if type(rays) is Rays and not self._needs_grad(rays):
            # the common call: plain rays, nothing to differentiate
            rec = K.trace_sequential_record(rays, prog, out=out)
            if rec is not None:
                self._deliver_sensor_visits(prog, rays, rec, dst=out)
            return rays if out is None else out
        from ..rays.ray import Paths
        if isinstance(rays, Paths):
            # visualisation path: in place on the wrapped rays, one snapshot per surface that was hit
            raw = rays.unwrap()
            sens_hit, sens_w, sens_mask, hist, hit_any = K.trace_sequential_history(raw, prog)
            self._deliver_sensor_records(prog, raw, sens_hit, sens_w, sens_mask)
            for s in hit_any.nonzero().flatten().tolist():
                rays._snapshot(hist[s])
            return rays
        from .. import autograd as AG
        if self._needs_grad(rays) and self._has_fuzzy():
            raise NotImplementedError("a Fuzzy filter in a differentiable trace: its function runs on the host and "
                                      "carries no gradient; trace under torch.no_grad()")
        if self._needs_grad(rays):
            # differentiable w.r.t. the scene parameters and the input rays: the backward re-traces the input
            # bundle, whose planes stay untouched because the result goes into new tensors
            pos, d, inten, sens_hit, sens_w, sens_mask = AG.trace_sequential_with_grad(
                list(self.elements), rays, prog, group=getattr(self, "grad_group", None))
            dst = rays if out is None else out
            dst._pos, dst._dir, dst.intensity = pos, d, inten
            if sens_hit.numel() == 0:
                sens_hit = sens_w = sens_mask = None
            self._deliver_sensor_records(prog, rays, sens_hit, sens_w, sens_mask)
            return dst
        sens_hit, sens_w, sens_mask = K.trace_sequential(rays, prog, out=out)
        self._deliver_sensor_records(prog, rays, sens_hit, sens_w, sens_mask, dst=out)
        return rays if out is None else out

    def getParaxial(self):
        """5x5 paraxial matrix of the whole system: the surface matrices of the elements in order, with free-space
        matrices over the z gaps between consecutive paraxial surfaces (scene/sequential.py:38-62).  Host-side
        algebra on the parameters; differentiable through torch as it stands."""
        import torch
        from ..elements.ideal import ParaxialDistMat
        zs, mats = [], []
        for el in self.elements:
            z, m = el.getParaxial()
            zs.extend(z)
            mats.extend(m)
        system = mats[0].clone()
        zs = torch.as_tensor(zs)   # detaches the positions, like the reference
        gaps = zs[1:] - zs[:-1]
        for i in range(len(mats) - 1):
            system = mats[i + 1] @ (ParaxialDistMat(gaps[i]) @ system)
        return system

    def checkDimensions(self):
        """Declared in the reference as a check of thicknesses and clearances (scene/sequential.py:64-74); there it
        stops at its first line, a call of `_getCenterThickness`, which no class defines."""
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '_getCenterThickness'")

    def to_base(self):
        """scene/sequential.py:80-105."""
        base = Scene()
        base.Nbounces = self.Nbounces
        for el in self.elements:
            base.add_element(el)
        for b, n in zip(self.bundles, self._bundle_N_rays):
            base.add_bundle(b, n)
        base.rays = self.rays
        base._build_index_maps()
        return base
