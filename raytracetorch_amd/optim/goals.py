"""Optimisation goals on sensor hits (reference: optim/goals.py:99-187).

``SpotSizeLoss`` = mean over bundles of  sum_i sqrt(((x_i-cx)^2 + (y_i-cy)^2) * w_i / W)  over hits
with w_i > 0, W = sum w_i (clamped at 1e-12) and (cx, cy) the weighted centroid unless a target is
given (optim/goals.py:165-183; the square root is inside the sum).  The sums are accumulated on
the device in float64 by ``rtt_spot_moments_f32`` directly from the dense sensor records, so no
per-hit tensors are materialised.
"""
from typing import List, Optional

import torch
import torch.nn as nn

from .. import kernels as K
from ..elements import Sensor
from ..rays.bundle import Bundle


class Goal(nn.Module):
    pass


def spot_size_from_records(records, target_xy=None, group=None, device=None):
    """records: iterable of (x [M], y [M], w [M], mask [M] uint8 or None) device tensors.
    Returns the loss tensor [] (differentiable w.r.t. x and y), or None when nothing lit was recorded
    (optim/goals.py:165-167).  With a rank `group` the sums run over all ranks, "nothing lit" is read off the
    summed weight (the same decision on every rank), and a rank without any entry still takes part in the
    exchanges with an empty record on `device`."""
    from ..autograd import SpotLossFn
    records = [r for r in records if r[0].numel() > 0]
    if not records:
        if group is None:
            return None
        e = torch.zeros(0, device=device)
        records = [(e, e.clone(), e.clone(), None)]
    if len(records) == 1:
        x, y, w, m = records[0]
    else:
        ones = lambda r: torch.ones(r[0].numel(), dtype=torch.uint8, device=r[0].device) if r[3] is None else r[3]
        x, y, w = (torch.cat([r[i] for r in records]) for i in range(3))
        m = torch.cat([ones(r) for r in records])
    loss, wsum = SpotLossFn.apply(x, y, w, m, target_xy, group)
    if group is not None and not float(wsum) > 0.0:
        return None
    return loss


class FocalLengthLoss(Goal):
    """(P_system - 1/f_target)^2 with the system power read off the paraxial matrix, P = -M[1, 0]
    (optim/goals.py:16-39).  No rays: host-side algebra on the parameters."""

    def __init__(self, f_target: float):
        super().__init__()
        self.P_target = nn.Parameter(torch.as_tensor(1.0 / f_target), requires_grad=False)

    def forward(self, scene) -> torch.Tensor:
        M = scene.getParaxial()
        return (-M[1, 0] - self.P_target) ** 2


class SpotTargetLoss(Goal):
    """Mean over bundles of the squared distance between the intensity centroid of a bundle's sensor hits and
    its target (optim/goals.py:42-96).  The scene is simulated with its no-argument `simulate()`, i.e. the
    nearest-hit bounce loop of `Scene`; gradients come from the differentiable form of that loop."""

    def __init__(self, sensor: Sensor, target_xy: torch.Tensor):
        super().__init__()
        self.sensor = sensor
        target_xy = torch.as_tensor(target_xy, dtype=torch.float32)
        if target_xy.ndim == 1:
            target_xy = target_xy.unsqueeze(0)
        self.register_buffer("target_xy", target_xy)

    def forward(self, scene, bundles: List[Bundle], N_rays: int = 128) -> torch.Tensor:
        losses = []
        for i, bundle in enumerate(bundles):
            self.sensor.reset()
            rays = bundle.sample(N_rays)
            dev = next(iter(scene.parameters()), torch.zeros(1)).device
            if dev.type != "cuda" and torch.cuda.is_available():
                dev = torch.device("cuda", torch.cuda.current_device())
            scene.rays = rays.to(dev)
            scene.simulate()
            if self.sensor.hitLocs:
                locs, w, _ = self.sensor.getHitsTensors()
                xy = locs[:, :2]
            else:
                xy, w = scene.rays.pos[:, :2], scene.rays.intensity
            group = getattr(scene, "grad_group", None)
            if group is None and len(w) == 0:
                continue
            sums = torch.stack([w.sum(), (xy[:, 0] * w).sum(), (xy[:, 1] * w).sum(), w.new_tensor(float(len(w)))])
            if group is not None:
                # rays sharded over ranks: the centroid is that of all hits, so the sums (and the number of hits, on
                # which the skip is decided) run over the group; every rank enters the exchange, also one without a hit
                from ..autograd import GroupSumFn
                sums = GroupSumFn.apply(sums, group)
                if not float(sums[3]) > 0.0:
                    continue
            w_sum = sums[0].clamp(min=1e-12)
            cx, cy = sums[1] / w_sum, sums[2] / w_sum
            t = self.target_xy[min(i, self.target_xy.shape[0] - 1)].to(cx.device)
            losses.append((cx - t[0]) ** 2 + (cy - t[1]) ** 2)
        if not losses:
            return torch.tensor(0.0)
        return torch.stack(losses).mean()


class SpotSizeLoss(Goal):

    def __init__(self, sensor: Sensor, bundles: List[Bundle], N_rays: int = 128,
                 target_xy: Optional[torch.Tensor] = None):
        super().__init__()
        self.sensor = sensor
        self.bundles = bundles
        self.N_rays = N_rays
        if target_xy is not None:
            target_xy = torch.as_tensor(target_xy, dtype=torch.float32)
            if target_xy.ndim == 1:
                target_xy = target_xy.unsqueeze(0).expand(len(bundles), 2)
        self._target_xy = target_xy

    def _fused(self, scene, rays, tgt):
        """Trainable sequential scene whose sensor is visited once: trace without a graph and attach the
        single fused backward.  Returns (loss, lit) — `lit` a 0/1 device tensor, 0 when nothing lit reached the
        sensor — or None when the fused form does not apply."""
        from .. import autograd as AG
        from .. import kernels as K
        from .. import program as P
        from ..rays.ray import Rays
        if not scene._needs_grad() or rays._pos.requires_grad or rays._dir.requires_grad:
            return None   # rays with a graph of their own (a trainable bundle pose) take the general path
        prog = scene._program()
        slots = [int(r["sensor_slot"]) for r in prog if int(r["sensor_slot"]) >= 0
                 and scene.elements[int(r["element_id"])] is self.sensor]
        if len(slots) != 1 or P.n_sensor_slots(prog) != 1 or not AG.one_sweep(prog, len(scene.elements)):
            return None   # (a long program is differentiated in stretches, by the general path)
        from ..rays.ray import PendingTrace, SampledRays
        src = Rays.from_planes(rays._pos.detach(), rays._dir.detach(), rays.intensity.detach(), rays.id, rays.wavelength)
        k = slots[0]
        # the trace leaves the first moments of the sensor's hits (W, sum w x, sum w y) beside the record
        mom = torch.zeros(6, device=src.device, dtype=torch.float64)
        if len(src) == 0:
            out = Rays.from_planes(torch.empty_like(src._pos), torch.empty_like(src._dir), torch.empty_like(src.intensity),
                                   rays.id, rays.wavelength)
            hit, w, mask = K.trace_sequential(src, prog, out=out, moments=(k, mom))
            scene._deliver_sensor_records(prog, src, hit, w, mask, dst=out)
            rays._pos, rays._dir, rays.intensity = out._pos, out._dir, out.intensity
        else:
            # The loss reads the sensor, not the traced bundle: the trace leaves the rays unwritten (28 B per ray) and
            # `scene.rays` gets them from a repeat of the trace should anybody look at them.
            kept = None
            if AG.keeps_tape(prog):   # the trace also keeps what the reverse sweep needs (no second trace in the backward)
                hit, w, mask, kept = K.trace_sequential(src, prog, moments=(k, mom), store=False, tape=True)
            else:
                hit, w, mask = K.trace_sequential(src, prog, moments=(k, mom), store=False)
            scene._deliver_sensor_records(prog, src, hit, w, mask)
            SampledRays.defer(rays, PendingTrace(src, prog, src.id, src.wavelength))
            if kept is not None:
                return AG.fused_spot_loss(list(scene.elements), src, prog, k, (hit[k], w[k], mask[k], kept), target=tgt,
                                          group=getattr(scene, "grad_group", None), params=scene._element_params()[1],
                                          table_cache=(scene.__dict__, scene._prog_cache[0]), mom=mom)
        # the host work of the backward's set-up (parameter list, slot Jacobians) runs while the device traces;
        # "nothing lit reached the sensor" (optim/goals.py:165-167) is read off the first moment, W = 0
        return AG.fused_spot_loss(list(scene.elements), src, prog, k, (hit[k], w[k], mask[k]), target=tgt,
                                  group=getattr(scene, "grad_group", None), params=scene._element_params()[1],
                                  table_cache=(scene.__dict__, scene._prog_cache[0]), mom=mom)

    def _sensor_records(self):
        s = self.sensor
        if s._pending:
            s._resolve_pending()
        recs = [(hit[0], hit[1], w, mask) for hit, w, mask, _ in s._dense]
        for locs, w in zip(s._locs, s._inten):
            recs.append((locs[:, 0].contiguous(), locs[:, 1].contiguous(), w.contiguous(), None))
        return recs

    def forward(self, scene) -> torch.Tensor:
        from ..scene import SequentialScene
        losses, counted = [], []   # per bundle: the loss, and None (counts) or a 0/1 device tensor (counts when lit)
        for i, bundle in enumerate(self.bundles):
            self.sensor.reset()
            rays = bundle.sample(self.N_rays)
            dev = next(iter(scene.parameters()), torch.zeros(1)).device
            if dev.type != "cuda" and torch.cuda.is_available():
                dev = torch.device("cuda", torch.cuda.current_device())
            rays = rays.to(dev)
            scene.rays = rays
            tgt = None
            if self._target_xy is not None:
                tgt = self._target_xy[min(i, self._target_xy.shape[0] - 1)]
            if isinstance(scene, SequentialScene):
                fused = self._fused(scene, rays, tgt)
                if fused is not None:
                    # whether anything lit reached the sensor is known on the device only; the bundle is weighed by
                    # that flag below instead of being skipped here, which would cost a host round trip per bundle
                    losses.append(fused[0])
                    counted.append(fused[1])
                    continue
                scene.simulate(rays)   # the reference's no-argument call only works on Scene
            else:
                scene.simulate()
            recs = self._sensor_records()
            if not recs:
                r = scene.rays
                recs = [(r._pos[0], r._pos[1], r.intensity, None)]
            group = getattr(scene, "grad_group", None)
            if group is None and not bool(torch.stack([((r[2] > 0) if r[3] is None else ((r[2] > 0) & r[3].bool())).any()
                                                       for r in recs]).any()):
                continue  # nothing lit reached the sensor (optim/goals.py:165-167)
            # with a rank group that decision is taken inside, on the weight summed over all ranks: a rank whose
            # shard stayed dark must still enter the exchanges of the forward and of the backward
            loss = spot_size_from_records(recs, tgt, group=group, device=dev)
            if loss is not None:
                losses.append(loss)
                counted.append(None)
        if not losses:
            return torch.tensor(0.0)
        if len(losses) == 1 and counted[0] is not None:
            return losses[0]   # an unlit bundle's loss is 0 with zero gradients: the mean over "the lit ones" is the same number
        if all(c is None for c in counted):
            return torch.stack(losses).mean()
        # mean over the bundles that were lit (optim/goals.py:165-167, :185-187); 0 when none was
        flags = torch.stack([torch.ones((), device=l.device) if c is None else c.to(l.device) for l, c in zip(losses, counted)])
        return (torch.stack(losses) * flags).sum() / flags.sum().clamp_min(1.0)
