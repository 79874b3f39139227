"""Non-sequential ("naive") scene: every bounce, each ray goes to the nearest surface it hits
(reference: scene/base.py:8-289).

The reference materialises an [N, S] distance matrix, takes a row-min, then dispatches S masked
sub-batches from Python.  Here one kernel per bounce keeps the running (t_min, winner) in
registers while it loops over the packed surface program, then applies the winner's interaction.
"""

import numpy as np
import torch
import torch.nn as nn

from .. import kernels as K
from .. import program as P
from ..elements import Element
from ..rays import Bundle, Rays




class Scene(nn.Module):

    def __init__(self):
        super().__init__()
        self.elements = nn.ModuleList()
        self.bundles = nn.ModuleList()
        self._bundle_N_rays = []
        self.rays = None
        self.Nbounces = 100
        self._build_index_maps()

    # ---- population -----------------------------------------------------------------------------
    def add_element(self, element: Element):
        self.elements.append(element)

    def add_bundle(self, bundle: Bundle, N_rays: int = 200):
        self.bundles.append(bundle)
        self._bundle_N_rays.append(N_rays)

    def clear_elements(self):
        self.elements = nn.ModuleList()
        self._build_index_maps()

    def clear_bundles(self):
        self.bundles = nn.ModuleList()
        self._bundle_N_rays = []
        self.rays = None

    def clear_rays(self):
        self.rays = None

    # ---- rays -----------------------------------------------------------------------------------
    def _build_rays(self):
        """Sample every bundle and concatenate (scene/base.py:57-90)."""
        if len(self.bundles) == 0:
            self.rays = None
            return
        parts = [b.sample(n) for b, n in zip(self.bundles, self._bundle_N_rays)]
        if len(parts) == 1:
            self.rays = parts[0]
            return
        self.rays = Rays.from_planes(
            torch.cat([r._pos for r in parts], dim=1).contiguous(),
            torch.cat([r._dir for r in parts], dim=1).contiguous(),
            torch.cat([r.intensity for r in parts]), torch.cat([r.id for r in parts]),
            torch.cat([r.wavelength for r in parts]))

    # ---- surface bookkeeping ----------------------------------------------------------------------
    def _build_index_maps(self):
        """Global surface index -> (element index, surface index) (scene/base.py:96-123)."""
        old = getattr(self, "map_to_element", None)
        device = old.device if isinstance(old, torch.Tensor) else torch.device("cpu")
        el, sf = [], []
        for k, element in enumerate(self.elements):
            n = len(element.shape)
            el += [k] * n
            sf += list(range(n))
        # simulate() asks for the maps on every call; they change only with the element list
        if isinstance(old, torch.Tensor) and self.__dict__.get("_index_map_lists") == (el, sf):
            return
        self.__dict__["_index_map_lists"] = (el, sf)
        self.register_buffer("map_to_element", torch.tensor(el, dtype=torch.long, device=device))
        self.register_buffer("map_to_surface", torch.tensor(sf, dtype=torch.long, device=device))
        self.total_surfaces = len(el)

    def _element_params(self):
        """(ids of the elements, parameters of the scene), re-collected only when the element list changes:
        walking the module tree costs more than launching a small trace."""
        els = tuple(map(id, self.elements._modules.values()))
        plist = self.__dict__.get("_param_list")
        if plist is None or plist[0] != els:
            # the entry keeps the elements themselves: while it exists none of them can be freed and hand its id to a
            # new element, which would make a changed element list look unchanged
            plist = self.__dict__["_param_list"] = (els, list(self.elements.parameters()), list(self.elements))
        return plist

    def _needs_grad(self, rays=None) -> bool:
        """Whether a trace must be differentiable: a trainable element parameter, or (when `rays` is given) input
        rays that carry a graph themselves (a trainable bundle pose, an earlier differentiable trace)."""
        if not torch.is_grad_enabled():
            return False
        if rays is not None and (rays._pos.requires_grad or rays._dir.requires_grad):
            return True
        params = self._element_params()[1]
        snap = P.snapshot(params)
        # `_program()` is what a caller runs next (no user code in between): it takes this pass instead of making its own
        self.__dict__["_snapshot_ahead"] = (params, snap)
        return snap[1]

    def _program(self):
        """Packed surface program of the current elements; re-packed only when an element list entry or a
        parameter changed since the last call.  Host-resident parameters are compared by value (so a write through
        `.data`, which leaves the version counter alone, is seen as well); device-resident ones by storage address and
        version counter: an in-place write through `p.data` on the device is not seen before `invalidate_program()` —
        write through `p` under `torch.no_grad()` instead.  All of it is one pass over the parameters (`P.snapshot`)."""
        els, params, _ = self._element_params()
        ahead = self.__dict__.pop("_snapshot_ahead", None)
        # one pass over the parameters in C++ (just made by `_needs_grad`, or made now)
        storage, _, versions, values = ahead[1] if ahead is not None and ahead[0] is params else P.snapshot(params)
        key = (els, storage, values, versions)
        cached = getattr(self, "_prog_cache", None)
        if cached is None or cached[0] != key:
            P.prefetch_host(params)
            with torch.random.fork_rng(devices=[]):   # keys drawn while packing are replaced below
                cached = (key, P.compile_elements(self.elements))
            self._prog_cache = cached
        prog = cached[1]
        if P.has_stochastic_functor(prog):
            # RefractFresnel draws new random numbers in every trace (phys/std.py:200): one key per call from
            # torch's default generator; the kernels add the surface's position to it
            stochastic = prog["phys_kind"] == P.PHYS_FRESNEL
            key = int(torch.randint(0, 2 ** 62, (1,)).item())
            prog = prog.copy()
            prog["seed"][stochastic] = (key & 0xFFFFFFFF, key >> 32)
        return prog

    def invalidate_program(self):
        """Forget the packed program and every cached host copy of a device-resident parameter of this scene: the next
        trace reads all parameters again (needed only after in-place writes through `p.data` on the device)."""
        self.__dict__.pop("_prog_cache", None)
        self.__dict__.pop("_snapshot_ahead", None)
        for q in self._element_params()[1]:
            P.forget_host_copy(q)

    def _slot_plan(self, prog):
        """[(sensor slot, element, intensity function of a Fuzzy filter or None)] in program order, kept while the
        element list is the same (the per-record walk costs as much as a small launch)."""
        plan = self.__dict__.get("_slot_plan_cache")
        if plan is not None and plan[2] is prog:   # the cached program object itself: nothing to compare
            return plan[1]
        key = (self._element_params()[0], prog.shape[0])   # slots depend on the element list, not on parameter values
        if plan is not None and plan[0] == key:
            plan = self.__dict__["_slot_plan_cache"] = (key, plan[1], prog)
        else:
            rows = []
            for rec in prog:
                k = int(rec["sensor_slot"])
                if k >= 0:
                    el = self.elements[int(rec["element_id"])]
                    fn = el.surface_functions[int(rec["surface_id"])]
                    rows.append((k, el, getattr(fn, "block_function", None)))
            plan = self.__dict__["_slot_plan_cache"] = (key, rows, prog)
        return plan[1]

    def _deliver_sensor_visits(self, prog, rays, record, dst=None):
        """The same for a launch whose record is still one allocation (`kernels.SensorRecord`): every sensor gets a
        `SensorVisit` that makes its views when it is read.  A Fuzzy filter needs the tensors at once."""
        plan = self._slot_plan(prog)
        for row in plan:
            if row[2] is not None:
                return self._deliver_sensor_records(prog, rays, *record.views(), dst=dst)
        ids = rays.id
        for k, el, _ in plan:
            el._record_visit(K.SensorVisit(record, k, ids))

    def _deliver_sensor_records(self, prog, rays, sens_hit, sens_w, sens_mask, dst=None):
        """Hand the dense records of a trace to the sensors.  Slots are numbered in program order; a Fuzzy
        filter's slot is not a sensor's: its function of the recorded local hit points multiplies the intensity of
        the rays that hit it, i.e. the final intensities and what every later slot recorded as incoming intensity
        (phys/filter.py:36-50 inside scene/sequential.py:12-36)."""
        if sens_hit is None:
            return
        factor = None
        for k, el, fuzzy in self._slot_plan(prog):
            if factor is not None:
                sens_w[k] *= factor
            if fuzzy is not None:
                with torch.no_grad():
                    f = fuzzy(sens_hit[k].t()).to(sens_w.dtype).reshape(-1)
                    f = torch.where(sens_mask[k].bool(), f, torch.ones_like(f))
                factor = f if factor is None else factor * f
            else:
                el._record_dense(sens_hit[k], sens_w[k], sens_mask[k], rays.id)
        if factor is not None:
            dst = rays if dst is None else dst   # the rays that hold the result of the trace
            dst.intensity = dst.intensity * factor

    # ---- simulation -------------------------------------------------------------------------------
    def simulate(self):
        """Up to `Nbounces` bounces of `self.rays` (scene/base.py:129-142)."""
        if self.rays is None:
            self._build_rays()
        if self.rays is None:
            return
        self._build_index_maps()
        if not hasattr(self.rays, "_snapshot") and len(self.elements) and self._needs_grad(self.rays):
            return self._bounce_with_grad(self.rays, self.Nbounces)
        if (hasattr(self.rays, "_snapshot") or len(self.elements) == 0 or self._has_fuzzy()
                or self.total_surfaces > P.MAX_SURFACES):   # (the fused bounce loop is one launch: at most 64 surfaces)
            # rays wrapped in Paths want a snapshot per bounce, and a Fuzzy filter changes intensities (which decide
            # whether a ray bounces on) between bounces: one launch per bounce, like the reference
            for _ in range(self.Nbounces):
                if not (self.rays.intensity > 0).any():
                    break
                self.step()
            return
        # the whole bounce loop in one launch: a ray that misses or dies is not touched by later bounces, so
        # letting each ray stop on its own gives what the reference's lock-step loop gives
        prog = self._program()
        _, hits = K.simulate_nearest(self.rays, prog, self.Nbounces, first_capacity=getattr(self, "_hit_capacity", None),
                                     defer=True)
        if hits is None:
            return
        if not isinstance(hits, K.PendingHits):
            return self._deliver_sensor_hits(prog, self.rays, *hits)
        # The call returns without waiting for the device.  The launch's crossings become the sensors' list entries
        # when one of them is read: count, sort and split happen then, once for all sensors of the scene.
        plan, ids, split = self._slot_plan(prog), self.rays.id, {}

        def entries_of(sensor):
            if not split:
                split["entries"] = self._split_sensor_hits(plan, ids, *hits.resolve())
            return [e[1:] for e in split["entries"] if e[0] is sensor]
        held = hits.nbytes()   # the hit list and the input planes stay allocated until the entry is read or dropped
        for el in {id(el): el for _, el, fuzzy in plan if fuzzy is None}.values():
            el._record_pending(entries_of, held)

    def _has_fuzzy(self) -> bool:
        return any(hasattr(fn, "block_function") for el in self.elements for fn in el.surface_functions)

    def _bounce_with_grad(self, rays, max_bounces):
        """Bounces of `rays` as one differentiable function of the element parameters (the reference records
        every tensor op of scene/base.py:180-235 on the autograd tape; here the trace carries the tangents)."""
        from .. import autograd as AG
        if self._has_fuzzy():
            raise NotImplementedError("a Fuzzy filter in a differentiable trace: its function runs on the host and "
                                      "carries no gradient; trace under torch.no_grad()")
        prog = self._program()
        if prog.shape[0] > P.MAX_SURFACES:
            raise NotImplementedError(f"the differentiable nearest-hit bounce takes at most {P.MAX_SURFACES} surfaces "
                                      f"(this scene has {prog.shape[0]}); a SequentialScene of any length is differentiable")
        pos, dir_, inten, hit, keys = AG.simulate_nearest_with_grad(self.elements, rays, prog, max_bounces,
                                                                    getattr(self, "_hit_capacity", None),
                                                                    group=getattr(self, "grad_group", None))
        rays._pos, rays._dir, rays.intensity = pos, dir_, inten
        if P.n_sensor_slots(prog) > 0:
            self._deliver_sensor_hits(prog, rays, hit, keys)

    def _deliver_sensor_hits(self, prog, rays, hit, keys):
        """Hand an unsorted hit list to the sensors as the reference's per-bounce list entries: one entry per
        bounce in which the sensor was hit, rows in ray order (elements/sensor.py:35-37 under scene/base.py:213-226)."""
        for el, locs, w, ids in self._split_sensor_hits(self._slot_plan(prog), rays.id, hit, keys):
            el._record_hits(locs, w, ids)

    @staticmethod
    def _split_sensor_hits(plan, ray_ids, hit, keys):
        """[(sensor, locs [m,3], w [m], ids [m])] in the order the reference appends them: by bounce, then by sensor
        surface, rows in ray order."""
        if hit.shape[0] == 0:
            return []
        # one 64-bit key per row, (bounce, slot) in the high word and the ray in the low one: a single sort gives the
        # order, and the sorted keys give the groups and the rays without a second gather
        key, order = torch.sort((keys[:, 0].to(torch.int64) << 32) | keys[:, 1].to(torch.int64))
        hit = hit[order]
        ids = ray_ids[key & 0xFFFFFFFF]
        heads, counts = torch.unique_consecutive(key >> 32, return_counts=True)
        heads, counts = torch.stack([heads, counts]).tolist()
        slot_of = {k: el for k, el, _ in plan}
        a, out = 0, []
        for g, c in zip(heads, counts):
            out.append((slot_of[g % P.MAX_SURFACES], hit[a:a + c, :3], hit[a:a + c, 3], ids[a:a + c]))
            a += c
        return out

    def ray_cast(self, rays):
        """(hit_mask [N], winner_element_ids [N], winner_surf_ids [N]) or None when nothing is hit
        (scene/base.py:144-178)."""
        if len(self.elements) == 0:
            return None
        raw = rays.unwrap() if hasattr(rays, "unwrap") else rays
        winner, _ = K.ray_cast(raw, self._program())
        hit = winner >= 0
        if not hit.any():
            return None
        idx = winner.clamp_min(0).long()
        if self.map_to_element.device != idx.device:
            self._build_index_maps()
            self.map_to_element = self.map_to_element.to(idx.device)
            self.map_to_surface = self.map_to_surface.to(idx.device)
        return hit, self.map_to_element[idx], self.map_to_surface[idx]

    def step(self):
        """One bounce for every ray that hits something and still carries intensity
        (scene/base.py:180-235)."""
        if self.rays is None or len(self.elements) == 0:
            return
        raw = self.rays.unwrap() if hasattr(self.rays, "unwrap") else self.rays
        recording = raw is not self.rays and hasattr(self.rays, "_snapshot")   # rays wrapped in Paths
        if self._needs_grad(raw):
            before = raw._pos.detach()
            self._bounce_with_grad(raw, 1)
            if recording and not torch.equal(before, raw._pos.detach()):
                self.rays._snapshot()
            return
        prog = self._program()
        if not recording and type(raw) is Rays and prog.shape[0] <= P.MAX_SURFACES:
            # the common call: plain rays in place, nobody asks for the winners
            rec = K.step_nearest_record(raw, prog)
            if rec is not None:
                self._deliver_sensor_visits(prog, raw, rec)
            return
        lit = raw.intensity > 0 if recording else None
        winner, sens_hit, sens_w, sens_mask = K.step_nearest(raw, prog, want_winner=recording)
        self._deliver_sensor_records(prog, raw, sens_hit, sens_w, sens_mask)
        if recording and bool(((winner >= 0) & lit).any()):
            self.rays._snapshot()   # one snapshot per bounce that moved a ray (scene/base.py:201-235)

    def compile_elements(self):
        """The reference wraps elements in torch.compile here (scene/base.py:237-255); the HIP
        path is already a single fused kernel per bounce, so this only refreshes the index maps."""
        self._build_index_maps()

    def to_sequential(self):
        """Elements sorted by the z of their transform (scene/base.py:261-289)."""
        from .sequential import SequentialScene
        seq = SequentialScene(sorted(self.elements, key=lambda el: float(el.shape.transform.trans[2])))
        seq.Nbounces = self.Nbounces
        for b, n in zip(self.bundles, self._bundle_N_rays):
            seq.add_bundle(b, n)
        seq.rays = self.rays
        return seq
