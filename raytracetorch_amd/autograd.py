"""Autograd for the device trace: gradients of anything computed from the traced rays / sensor hits
with respect to the scene's trainable parameters.

The reference gets these by letting torch autograd record every elementwise op of the trace
(hundreds of saved [N] tensors per surface visit).  Here the trace is one kernel, so the backward is
one kernel as well.  The sequential trace uses reverse mode (``rtt_trace_seq_adjoint_f32``): every ray is
re-traced once, the adjoint of its state is swept back through the surfaces it hit, and the kernel
returns the adjoints of the record slots summed over the rays — its cost does not depend on the number
of trainable scalars, and the adjoint that arrives at the start of a ray is the gradient with respect to
the input rays.  The forward-mode kernels (``rtt_trace_seq_tangent_f32``: dual numbers over up to 8
parameter directions per launch, contracted with the upstream adjoints on the fly) remain for programs
whose per-surface states do not fit the LDS of a compute unit, for the nearest-hit bounce loop, and as an
independent check (``RTT_BACKWARD=tangent``).  Nothing per-ray is saved between forward and backward
except the input bundle itself.

Host side, the derivative of each packed record slot with respect to each trainable scalar
(d(c(1+k))/dc, dR/d(rot_vec), d(n_out/n_in)/d(n_in), shared Parameters, ...) is obtained by building
the differentiable slots as tiny torch expressions (`_diff_slots`) and asking torch for their Jacobian.
"""
import ctypes
import os
from typing import List, Tuple

import numpy as np
import torch

from . import _native as N
from . import kernels as K
from . import program as P

from .sharding import allreduce_sum_

# the float64 moment sums of SpotLossFn; a module attribute so that the multi-process CPU test can
# substitute a host implementation (the product path has none)
_spot_moments = K.spot_moments

TAN_SLOTS = 32  # Rs[9] ts[3] Rj[9] tj[3] geom[4] phys[4]
_SLOT_ORDER = (("Rs", 9), ("ts", 3), ("Rj", 9), ("tj", 3), ("geom", 4), ("phys", 4))


_mask_cache = {}


def _mask_list(mask: torch.Tensor):
    """Host copy of a gradient mask; masks are constant buffers, so a device-resident one is fetched once.  The
    entry keeps a weak reference to the tensor it was made from: an id can be reused after that tensor died."""
    import weakref
    key = (id(mask), mask.data_ptr(), mask._version)
    hit = _mask_cache.get(key)
    if hit is None or hit[0]() is not mask:
        if len(_mask_cache) > 1024:
            _mask_cache.clear()
        hit = _mask_cache[key] = (weakref.ref(mask), mask.reshape(-1).tolist())
    return hit[1]


def trainable_scalars(elements) -> List[Tuple[torch.nn.Parameter, int]]:
    """(parameter, flat index) for every trainable scalar of the elements, shared Parameters once.
    Components that a gradient mask (trans_mask / rot_mask) zeroes anyway are skipped."""
    seen, out = set(), []
    for el in elements:
        masks = {}
        for mod in el.modules():
            for name in ("trans", "rot_vec"):
                p = getattr(mod, name, None)
                m = getattr(mod, name.split("_")[0] + "_mask", None)
                if isinstance(p, torch.nn.Parameter) and m is not None:
                    masks[id(p)] = m
        for p in el.parameters():
            if not p.requires_grad or id(p) in seen:
                continue
            seen.add(id(p))
            mask = masks.get(id(p))
            keep = None if mask is None else _mask_list(mask)
            for i in range(p.numel()):
                if keep is None or keep[i] != 0.0:
                    out.append((p, i))
    return out


def tangent_table(elements, thetas) -> np.ndarray:
    """[n_surf, 32, n_theta] float32: d(record slot)/d(theta_k), by torch autograd on the slot expressions."""
    rows = []  # (surface index, slot offset, tensor)
    s = 0
    for el in elements:
        for j in range(len(el.shape)):
            slots = el._diff_slots(j)
            off = 0
            for name, width in _SLOT_ORDER:
                t = slots.get(name)
                if t is not None and t.requires_grad:
                    rows.append((s, off, t.reshape(-1).to(torch.float32)))
                off += width
            s += 1
    table = np.zeros((s, TAN_SLOTS, len(thetas)), dtype=np.float32)
    if not rows or not thetas:
        return table
    params, index_of = [], {}
    for p, _ in thetas:
        if id(p) not in index_of:
            index_of[id(p)] = len(params)
            params.append(p)
    # the slot expressions live where the parameters live; everything comes to the host in one transfer
    home = rows[0][2].device
    y = torch.cat([r[2].to(home) for r in rows])
    eye = torch.eye(y.numel(), dtype=y.dtype, device=home)
    jac = torch.autograd.grad(y, params, grad_outputs=eye, is_grads_batched=True, allow_unused=True)
    widths = [p.numel() for p in params]
    flat = torch.cat([torch.zeros((y.numel(), w), dtype=y.dtype, device=home) if g is None else g.reshape(y.numel(), -1).to(home)
                      for g, w in zip(jac, widths)], dim=1).cpu().numpy()
    used = [g is not None for g in jac]
    jac, at = [], 0
    for w, u in zip(widths, used):
        jac.append(flat[:, at:at + w] if u else None)
        at += w
    pos = 0
    for surf, off, t in rows:
        w = t.numel()
        for k, (p, i) in enumerate(thetas):
            g = jac[index_of[id(p)]]
            if g is not None:
                table[surf, off:off + w, k] = g[pos:pos + w, i]
        pos += w
    return table


def _tangent_program(prog, table, dev):
    """The records the tangent kernels read.  A rotation that is exactly the identity lets the kernels skip the
    matrix product, which is right for the value but drops d(R)/d(theta): where the table carries a non-zero
    rotation tangent the identity flag is cleared (x*1 + y*0 + z*0 == x, so the primal values do not change)."""
    flags = prog["flags"].copy()
    flags[np.any(table[:, 0:9, :] != 0.0, axis=(1, 2))] &= ~np.int32(P.F_SHAPE_ROT_ID)
    flags[np.any(table[:, 12:21, :] != 0.0, axis=(1, 2))] &= ~np.int32(P.F_SURF_ROT_ID)
    if np.array_equal(flags, prog["flags"]):
        return N.DeviceProgram.upload(prog, dev)
    prog = prog.copy()
    prog["flags"] = flags
    return N.DeviceProgram.upload(prog, dev, tag="tangent")


def _tangent_contract(rays_in, prog, table, adj_pos, adj_dir, adj_sens):
    """sum_i <adjoint_i, d out_i / d theta_k> for all theta, at most 8 directions per launch."""
    n = len(rays_in)
    dev = rays_in.device
    n_theta = table.shape[2]
    out = torch.zeros(n_theta, dtype=torch.float64, device=dev)
    if n == 0 or n_theta == 0:
        return out
    dp = _tangent_program(prog, table, dev)
    fix = lambda a: None if a is None else a.contiguous().to(torch.float32)
    adj_pos, adj_dir, adj_sens = fix(adj_pos), fix(adj_dir), fix(adj_sens)
    for k0, used, padded in _chunks(table):  # kernels exist for 2, 4, 6 and 8 directions; 8 is never the cheapest
        width = padded.shape[2]
        tan_dev = torch.from_numpy(padded).to(dev)
        acc = torch.zeros(width, dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            code = N.lib().rtt_trace_seq_tangent_f32(
                N.ptr(rays_in._pos), N.ptr(rays_in._dir), N.ptr(rays_in.intensity), n, n,
                N.ptr(dp.dev), dp.host_ptr, dp.n, N.ptr(tan_dev), width,
                N.ptr(adj_pos), N.ptr(adj_dir), n, N.ptr(adj_sens), n, N.ptr(acc), N.stream_handle(dev))
        N.check(code, "rtt_trace_seq_tangent_f32")
        out[k0:k0 + used] = acc[:used]
    return out


def backward_mode() -> str:
    """How the sequential trace is differentiated: "adjoint" (reverse mode with a second trace inside the sweep, the
    default), "tape" (reverse mode swept from a tape that the forward trace keeps, for programs short enough; measured
    slower on the whole step: the tape costs the forward trace more HBM writes than the sweep saves arithmetic) or
    "tangent" (forward mode)."""
    mode = os.environ.get("RTT_BACKWARD", "adjoint")
    if mode not in ("adjoint", "tape", "tangent"):
        raise ValueError("RTT_BACKWARD must be 'adjoint', 'tape' or 'tangent'")
    return mode


def keeps_tape(prog) -> bool:
    """Whether a differentiable trace of this program keeps a tape for its backward (16 B per ray and surface hit
    instead of a second trace): only when asked for, and for a program short enough for the sweep that reads tapes."""
    return backward_mode() == "tape" and K.tape_slots(prog.shape[0]) > 0


def _slot_pairs(table):
    """(surface, slot) of every record slot that moves with some trainable scalar, in the kernel's accumulator
    order (by surface, then by slot)."""
    return np.argwhere(np.any(table != 0.0, axis=2))


def _adjoint_plan(table, budget, dev, derived=None):
    """Launches of the reverse sweep for a tangent table: [(slot masks [n_surf] uint32, number of slot sums, device
    matrix [slot sums, n_theta] float64 that turns the slot adjoints into d/d(theta))], at most `budget` slot sums
    per launch.  `derived` (a dict that lives as long as the table) keeps the plan, so that a table which is reused
    costs no host-to-device transfer: such a transfer from pageable memory waits for the stream."""
    key = ("adjoint plan", str(dev), budget)
    if derived is not None and key in derived:
        return derived[key]
    pairs = _slot_pairs(table)
    plan = []
    for k in range(0, max(len(pairs), 1), max(budget, 1)):
        chunk = pairs[k:k + budget]
        masks = np.zeros(table.shape[0], dtype=np.uint32)
        for sfc, slt in chunk:
            masks[sfc] |= np.uint32(1) << np.uint32(slt)
        rows = torch.from_numpy(table[chunk[:, 0], chunk[:, 1], :].astype(np.float64)).to(dev) if len(chunk) else None
        plan.append((masks, len(chunk), rows))
    if derived is not None:
        derived[key] = plan
    return plan


def _adjoint_sweep(rays_in, prog, table, adj_pos=None, adj_dir=None, adj_sens=None, want_ray_adj=False, spot=None,
                   derived=None, tape=None, spot_w=None):
    """Reverse-mode backward of the sequential trace: (d objective / d theta [n_theta] float64 on the device,
    [6, N] adjoint of the input rays or None).  The objective is given by the upstream adjoints of the output planes
    and of the sensor records, or, with `spot` = (sensor slot, centroid?, target x, target y, device tensor of the six
    moment sums), by the spot-size loss whose adjoint the kernel forms itself from those sums.  Returns None when the
    program is too long for the kernel's LDS budget (the caller then uses the forward-mode kernels).  `tape` (a
    `kernels.Tape` of the same rays and program; in spot mode with `spot_w`, the intensities that sensor recorded)
    replaces the sweep's own forward trace."""
    n, dev = len(rays_in), rays_in.device
    n_surf, n_theta = table.shape[0], table.shape[2]
    lib = N.lib()
    budget = K.tape_slots(n_surf) if tape is not None else lib.rtt_trace_seq_adjoint_max_slots(n_surf)
    if budget == 0:
        return None
    dtheta = torch.zeros(n_theta, dtype=torch.float64, device=dev)
    ray_adj = torch.zeros((6, n), dtype=torch.float32, device=dev) if want_ray_adj else None
    if n == 0:
        return dtheta, ray_adj
    dp = N.DeviceProgram.upload(prog, dev)
    fix = lambda a: None if a is None else a.contiguous().to(torch.float32)
    adj_pos, adj_dir, adj_sens = fix(adj_pos), fix(adj_dir), fix(adj_sens)
    slot, centroid, cx, cy, mom_dev = (-1, False, 0.0, 0.0, None) if spot is None else spot
    plan = _adjoint_plan(table, budget, dev, derived)
    for c, (masks, n_acc, rows) in enumerate(plan):
        slot_adj = torch.empty(max(n_acc, 1), dtype=torch.float64, device=dev)
        ws_bytes = int(lib.rtt_trace_seq_adjoint_workspace_bytes(n_acc))
        ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=dev)
        first = c == 0
        with torch.cuda.device(dev):
            if tape is not None:
                code = lib.rtt_trace_seq_adjoint_tape_f32(
                    N.ptr(rays_in._pos), n, n, N.ptr(dp.dev), dp.host_ptr, dp.n, masks.ctypes.data,
                    N.ptr(adj_pos), N.ptr(adj_dir), n, N.ptr(adj_sens), n,
                    int(slot), int(bool(centroid)), float(cx), float(cy), 1.0, N.ptr(mom_dev), N.ptr(spot_w),
                    N.ptr(ray_adj) if first else 0, n, N.ptr(slot_adj), N.ptr(ws), ws_bytes, N.ptr(tape.buf), tape.buf.numel(),
                    N.stream_handle(dev))
            else:
                code = lib.rtt_trace_seq_adjoint_f32(
                    N.ptr(rays_in._pos), N.ptr(rays_in._dir), N.ptr(rays_in.intensity), n, n,
                    N.ptr(dp.dev), dp.host_ptr, dp.n, masks.ctypes.data,
                    N.ptr(adj_pos), N.ptr(adj_dir), n, N.ptr(adj_sens), n,
                    int(slot), int(bool(centroid)), float(cx), float(cy), 1.0, N.ptr(mom_dev),
                    N.ptr(ray_adj) if first else 0, n, N.ptr(slot_adj), N.ptr(ws), ws_bytes, N.stream_handle(dev))
        N.check(code, "rtt_trace_seq_adjoint_tape_f32" if tape is not None else "rtt_trace_seq_adjoint_f32")
        if n_acc:
            contribution = slot_adj[:n_acc] @ rows
            dtheta = contribution if len(plan) == 1 else dtheta + contribution
    return dtheta, ray_adj


def _ray_adjoint(rays_in, prog, adj_pos, adj_dir, adj_sens):
    """[6, N]: d(objective)/d(pos x, y, z, dir x, y, z of every input ray) of a sequential trace."""
    n, dev = len(rays_in), rays_in.device
    out = torch.zeros((6, n), dtype=torch.float32, device=dev)
    if n == 0:
        return out
    dp = N.DeviceProgram.upload(prog, dev)
    fix = lambda a: None if a is None else a.contiguous().to(torch.float32)
    adj_pos, adj_dir, adj_sens = fix(adj_pos), fix(adj_dir), fix(adj_sens)
    zero_tan = torch.zeros((dp.n, TAN_SLOTS, 6), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        code = N.lib().rtt_trace_seq_ray_adjoint_f32(
            N.ptr(rays_in._pos), N.ptr(rays_in._dir), N.ptr(rays_in.intensity), n, n,
            N.ptr(dp.dev), dp.host_ptr, dp.n, N.ptr(zero_tan), N.ptr(adj_pos), N.ptr(adj_dir), n, N.ptr(adj_sens), n,
            N.ptr(out), n, N.stream_handle(dev))
    N.check(code, "rtt_trace_seq_ray_adjoint_f32")
    return out


class TraceSeqFn(torch.autograd.Function):
    """SequentialScene.simulate as a differentiable function of the scene parameters and of the input rays.

    forward(rays_in, prog, table, thetas, group, pos_in, dir_in, *params) -> (pos planes, dir planes, intensity,
    sens_hit, sens_w, sens_mask); only pos, dir and sens_hit carry gradient.  `rays_in` holds detached planes (what
    the kernels read); `pos_in` / `dir_in` are the same planes as the caller has them, possibly with a graph."""

    @staticmethod
    def forward(ctx, rays_in, prog, table, thetas, group, pos_in, dir_in, *params):
        from .rays.ray import Rays
        # the kernel writes every ray of `out`, so fresh uninitialised planes are enough
        out = Rays.from_planes(torch.empty_like(rays_in._pos), torch.empty_like(rays_in._dir),
                               torch.empty_like(rays_in.intensity), rays_in.id, rays_in.wavelength)
        ctx.tape = None
        if keeps_tape(prog):
            sens_hit, sens_w, sens_mask, ctx.tape = K.trace_sequential(rays_in, prog, out=out, tape=True)
        else:
            sens_hit, sens_w, sens_mask = K.trace_sequential(rays_in, prog, out=out)
        ctx.rays_in, ctx.prog, ctx.table, ctx.thetas, ctx.group = rays_in, prog, table, thetas, group
        ctx.param_meta = [(p.shape, p.dtype, p.device) for p in params]
        ctx.param_ids = [id(p) for p in params]
        ctx.set_materialize_grads(False)
        empty = torch.empty(0, device=rays_in.device)
        outs = (out._pos, out._dir, out.intensity, sens_hit if sens_hit is not None else empty,
                sens_w if sens_w is not None else empty, sens_mask if sens_mask is not None else empty)
        ctx.mark_non_differentiable(outs[2], outs[4], outs[5])
        return outs

    @staticmethod
    def backward(ctx, g_pos, g_dir, _gi, g_sens, _gw, _gm):
        if g_sens is not None and g_sens.numel() == 0:
            g_sens = None
        want_rays = ctx.needs_input_grad[5] or ctx.needs_input_grad[6]
        swept = None
        if backward_mode() != "tangent":   # one reverse sweep yields both kinds of gradient
            swept = _adjoint_sweep(ctx.rays_in, ctx.prog, ctx.table, g_pos, g_dir, g_sens, want_ray_adj=want_rays, tape=ctx.tape)
        g_pos_in = g_dir_in = None
        if swept is not None:
            dtheta, adj = swept
        else:
            dtheta = _tangent_contract(ctx.rays_in, ctx.prog, ctx.table, g_pos, g_dir, g_sens)
            adj = _ray_adjoint(ctx.rays_in, ctx.prog, g_pos, g_dir, g_sens) if want_rays else None
        allreduce_sum_(dtheta, ctx.group)
        if adj is not None:
            g_pos_in, g_dir_in = adj[:3], adj[3:]
        grads = _assemble_grads(dtheta, ctx.thetas, ctx.param_meta, ctx.param_ids)
        return (None, None, None, None, None, g_pos_in, g_dir_in, *grads)


def needs_grad(elements) -> bool:
    return torch.is_grad_enabled() and any(p.requires_grad for el in elements for p in el.parameters())


def rays_carry_grad(rays) -> bool:
    return torch.is_grad_enabled() and (rays._pos.requires_grad or rays._dir.requires_grad)


def _detached(rays):
    from .rays.ray import Rays
    return Rays.from_planes(rays._pos.detach(), rays._dir.detach(), rays.intensity.detach(), rays.id, rays.wavelength)


SEGMENT_SURFACES = 12   # surfaces per differentiated stretch of a long program (the sweep's LDS then holds 76 slot sums)


def one_sweep(prog, n_elements) -> bool:
    """Whether the program is differentiated in one piece: it fits a launch and (in reverse mode) the sweep's LDS holds
    a state per surface, or it cannot be cut (a single element)."""
    n_surf = prog.shape[0]
    if n_surf > P.MAX_SURFACES:
        return False
    return n_elements < 2 or backward_mode() == "tangent" or N.lib().rtt_trace_seq_adjoint_max_slots(n_surf) > 0


def _stretches(elements, prog):
    """A program too long to differentiate in one piece, as stretches of whole elements: [(elements, records)]."""
    if one_sweep(prog, len(elements)):
        return None
    reverse = backward_mode() != "tangent"
    ids = np.asarray(prog)["element_id"]
    if int(ids.max()) + 1 != len(elements):
        raise ValueError("the program was not compiled from these elements")
    widest = int(np.bincount(ids).max())
    limit = max(SEGMENT_SURFACES, widest) if reverse else P.MAX_SURFACES
    if limit > P.MAX_SURFACES:
        raise ValueError(f"an element of the scene has more than {P.MAX_SURFACES} surfaces")
    out = []
    for piece, first, _ in P.split_program(prog, limit):
        e0, e1 = int(ids[first]), int(ids[first + piece.shape[0] - 1]) + 1
        out.append((elements[e0:e1], piece))
    return out


def _trace_stretches(stretches, rays, group):
    """The differentiable trace of a long program: one TraceSeqFn per stretch, each fed the rays the one before it
    left.  Autograd chains their backward sweeps (the ray adjoint of a stretch is the upstream adjoint of the one in
    front of it) and adds up what a parameter shared between stretches receives."""
    from .rays.ray import Rays
    cur, parts = rays, []
    for els, piece in stretches:
        outs = trace_sequential_with_grad(els, cur, piece, group, whole=True)
        cur = Rays.from_planes(outs[0], outs[1], outs[2], rays.id, rays.wavelength)
        parts.append(outs[3:])
    joined = []
    for k in range(3):   # the sensor planes of the stretches, in slot order
        have = [p[k] for p in parts if p[k].numel()]
        joined.append(torch.cat(have, dim=0) if len(have) > 1 else have[0] if have else parts[0][k])
    return (cur._pos, cur._dir, cur.intensity, *joined)


def trace_sequential_with_grad(elements, rays, prog, group=None, whole=False):
    """Run the differentiable trace of `rays` (whose planes may carry a graph); returns the six outputs of
    TraceSeqFn.  A program longer than one sweep takes is differentiated in stretches (unless `whole`)."""
    stretches = None if whole else _stretches(elements, prog)
    if stretches is not None:
        return _trace_stretches(stretches, rays, group)
    thetas = _cached_scalars(elements)
    table = tangent_table(elements, thetas)
    params, seen = [], set()
    for p, _ in thetas:
        if id(p) not in seen:
            seen.add(id(p))
            params.append(p)
    return TraceSeqFn.apply(_detached(rays), prog, table, thetas, group, rays._pos, rays._dir, *params)


class SimulateNearestFn(torch.autograd.Function):
    """Scene.simulate (nearest-hit bounce loop) as a differentiable function of the scene parameters.

    forward(rays_in, prog, table, thetas, max_bounces, capacity, group, pos_in, dir_in, *params) -> (pos planes, dir
    planes, intensity, hit [M,4], keys [M,2]); pos, dir and the first three columns of hit carry gradient.  The
    forward launch carries the tangents along (forward mode, up to 8 directions per launch) and keeps them: d(final
    rays) as [6, K, N] and d(hit_local) as [M, 3, K]; the backward is their contraction with the upstream adjoints.
    When the input planes `pos_in` / `dir_in` carry a graph, one more launch keeps the per-ray Jacobian."""

    @staticmethod
    def forward(ctx, rays_in, prog, table, thetas, max_bounces, capacity, group, pos_in, dir_in, *params):
        dev = rays_in.device
        ctx.group = group
        dp = _tangent_program(prog, table, dev)
        chunks = list(_chunks(table, widest=8))
        seeds = ctx.needs_input_grad[7] or ctx.needs_input_grad[8]
        if not chunks:   # no trainable scalar: one launch for the values
            chunks = [(0, 0, np.zeros((table.shape[0], TAN_SLOTS, 2), np.float32))]
        if len(chunks) > 1 or seeds:
            # every launch must take the same branches: one kernel instantiation for all of them
            chunks = [(k0, used, np.concatenate([pad, np.zeros(pad.shape[:2] + (8 - pad.shape[2],), np.float32)], axis=2))
                      for k0, used, pad in chunks]
        kept, first = [], None
        for k0, used, padded in chunks:
            out = K.simulate_nearest_tangent(rays_in, dp, max_bounces, padded, capacity)
            if first is None:
                first = out
            elif not torch.equal(out[4], first[4]):
                raise RuntimeError("sensor hit lists of two tangent launches differ")
            kept.append((k0, used, out[5], out[6]))
        pos, dir_, inten, hit, keys = first[:5]
        ctx.jac = None
        if seeds:
            out = K.simulate_nearest_tangent(rays_in, dp, max_bounces, np.zeros((table.shape[0], TAN_SLOTS, 8), np.float32),
                                             capacity, seed_rays=True)
            if not torch.equal(out[4], keys):
                raise RuntimeError("sensor hit lists of two tangent launches differ")
            ctx.jac = (out[5][:, :6], out[6][:, :, :6], keys[:, 1].to(torch.int64))
        ctx.kept, ctx.thetas, ctx.n_theta = kept, thetas, table.shape[2]
        ctx.param_meta = [(p.shape, p.dtype, p.device) for p in params]
        ctx.param_ids = [id(p) for p in params]
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(inten, keys)
        return pos, dir_, inten, hit, keys

    @staticmethod
    def backward(ctx, g_pos, g_dir, _gi, g_hit, _gk):
        dev = ctx.kept[0][2].device
        dtheta = torch.zeros(ctx.n_theta, dtype=torch.float64, device=dev)
        f32 = lambda g: None if g is None else g.to(torch.float32)
        g_pos, g_dir, g_hit = f32(g_pos), f32(g_dir), (None if g_hit is None else f32(g_hit[:, :3]))
        for k0, used, ray_tan, hit_tan in ctx.kept:
            if used == 0:
                continue
            acc = torch.zeros(ray_tan.shape[1], dtype=torch.float64, device=dev)
            if g_pos is not None:
                acc += torch.einsum("ci,cki->k", g_pos, ray_tan[:3]).double()
            if g_dir is not None:
                acc += torch.einsum("ci,cki->k", g_dir, ray_tan[3:]).double()
            if g_hit is not None and hit_tan.shape[0]:
                acc += torch.einsum("mc,mck->k", g_hit, hit_tan).double()
            dtheta[k0:k0 + used] = acc[:used]
        allreduce_sum_(dtheta, ctx.group)   # rays sharded over ranks: parameter gradients are sums over all rays
        g_pos_in = g_dir_in = None
        if ctx.jac is not None:
            jac, hit_jac, hit_ray = ctx.jac
            adj = torch.zeros((6, jac.shape[2]), dtype=torch.float32, device=dev)
            if g_pos is not None:
                adj += torch.einsum("ci,cji->ji", g_pos, jac[:3])
            if g_dir is not None:
                adj += torch.einsum("ci,cji->ji", g_dir, jac[3:])
            if g_hit is not None and hit_jac.shape[0]:
                adj.index_add_(1, hit_ray, torch.einsum("mc,mcj->jm", g_hit, hit_jac))
            g_pos_in, g_dir_in = adj[:3], adj[3:]
        grads = _assemble_grads(dtheta, ctx.thetas, ctx.param_meta, ctx.param_ids)
        return (None,) * 7 + (g_pos_in, g_dir_in) + tuple(grads)


def simulate_nearest_with_grad(elements, rays, prog, max_bounces, capacity=None, group=None):
    """Run the differentiable nearest-hit bounce loop; returns the five outputs of SimulateNearestFn."""
    thetas = _cached_scalars(elements)
    table = tangent_table(elements, thetas)
    params, seen = [], set()
    for p, _ in thetas:
        if id(p) not in seen:
            seen.add(id(p))
            params.append(p)
    return SimulateNearestFn.apply(_detached(rays), prog, table, thetas, max_bounces, capacity, group, rays._pos, rays._dir, *params)


class GroupSumFn(torch.autograd.Function):
    """Sum of a small tensor over the ranks of a group, as a differentiable function: every rank goes on to compute
    the same global objective from the sum and back-propagates it into its own shard, and the parameter gradients
    of the shards are added up by the backward of the trace, so the adjoint passes through unchanged."""

    @staticmethod
    def forward(ctx, t, group):
        return allreduce_sum_(t.detach().clone(), group)

    @staticmethod
    def backward(ctx, g):
        return g, None


class SpotLossFn(torch.autograd.Function):
    """sum_i sqrt(((x_i-cx)^2 + (y_i-cy)^2) w_i / W) over entries with mask and w > 0
    (optim/goals.py:165-183), W and the centroid summed over all ranks of `group` when given.

    The value comes from the float64 moment kernels; the backward is the closed form
      dL/dx_i = g_i - (w_i / W) sum_j g_j  (centroid mode),  g_i = (x_i - cx) (w_i / W) / u_i,
    and w carries no gradient (it is the 0/1 intensity, as in the reference).

    Returns (loss, W): W (not differentiable) is the lit weight summed over all ranks, i.e. the one quantity on
    which every rank can base the same "nothing lit reached the sensor" decision (optim/goals.py:165-167).  Every
    rank of the group must call this function, also one whose shard recorded nothing: the exchanges pair up."""

    @staticmethod
    def forward(ctx, x, y, w, mask, target, group):
        mom = torch.zeros(4, device=x.device, dtype=torch.float64)
        xc, yc, wc = x.detach().contiguous(), y.detach().contiguous(), w.contiguous()
        _spot_moments(xc, yc, wc, mask, 0, 0.0, 0.0, 1.0, mom)
        allreduce_sum_(mom, group)
        W, Sx, Sy, _ = mom.tolist()
        W = max(W, 1e-12)
        cx, cy = (Sx / W, Sy / W) if target is None else (float(target[0]), float(target[1]))
        mom[3] = 0.0
        _spot_moments(xc, yc, wc, mask, 1, cx, cy, W, mom)
        loss = allreduce_sum_(mom[3:4].clone(), group)
        ctx.save_for_backward(xc, yc, wc, mask)
        ctx.consts = (cx, cy, W, target is None, group)
        wsum = mom[0:1].clone().reshape(())
        ctx.mark_non_differentiable(wsum)
        return loss.reshape(()).to(torch.float32), wsum

    @staticmethod
    def backward(ctx, g, _gw=None):
        x, y, w, mask = ctx.saved_tensors
        cx, cy, W, centroid, group = ctx.consts
        act = w > 0
        if mask is not None:
            act = act & mask.bool()
        wn = torch.where(act, w, torch.zeros_like(w)) / W
        dx = torch.where(act, x - cx, torch.zeros_like(x))
        dy = torch.where(act, y - cy, torch.zeros_like(y))
        u = torch.sqrt((dx * dx + dy * dy) * wn)
        inv = torch.where(u > 0, wn / u, torch.zeros_like(u))
        gx, gy = dx * inv, dy * inv
        if centroid:
            G = allreduce_sum_(torch.stack([gx.sum(dtype=torch.float64), gy.sum(dtype=torch.float64)]), group)
            gx = gx - wn * G[0].to(gx.dtype)
            gy = gy - wn * G[1].to(gy.dtype)
        return gx * g, gy * g, None, None, None, None


def _widths(n_theta, widest):
    """Directions per launch.  `widest` = 8: as few launches as possible.  `widest` = 6 (the sequential tangent
    kernels): the 8-wide instantiation holds one wavefront per SIMD and costs more than two 4-wide launches
    (16 Mi rays: 0.8 / 0.95 / 1.3 / 2.15 ms for 2 / 4 / 6 / 8 directions), so launches are 6 wide, and a rest of 8
    goes as 4 + 4."""
    if widest == 8:
        return [min(8, n_theta - k) for k in range(0, n_theta, 8)]
    n, out = n_theta, []
    while n > 0:
        if n <= 6:
            take = n
        elif n <= 8:
            take = 4
        else:
            take = 6
        out.append(take)
        n -= take
    return out


def _chunks(table, widest=6):
    """Yield (k0, width_used, padded [n_surf, 32, width]) with width in {2, 4, 6, 8}."""
    k0 = 0
    for used in _widths(table.shape[2], widest):
        chunk = table[:, :, k0:k0 + used]
        width = max(2, used + (used & 1))
        padded = np.zeros((table.shape[0], TAN_SLOTS, width), dtype=np.float32)
        padded[:, :, :used] = chunk
        yield k0, used, padded
        k0 += used


def _assemble_grads(dtheta, thetas, param_meta, param_ids, derived=None):
    """Per-parameter gradient tensors from the flat vector d(loss)/d(theta_k).  Parameters that live on the device
    of `dtheta` get theirs by one scatter on that device, with no synchronisation (the host may run ahead into the
    next step; `derived`, a dict that lives as long as the parameter list, keeps the scatter index on the device);
    otherwise one device-to-host transfer and the scatter on the host."""
    sizes = [int(np.prod(shape, dtype=np.int64)) for shape, _, _ in param_meta]
    starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    where = {pid: k for k, pid in enumerate(param_ids)}
    dst = [int(starts[where[id(p)]]) + int(i) for p, i in thetas]
    if param_meta and all(dev == dtheta.device for _, _, dev in param_meta):
        flat = torch.zeros(int(starts[-1]), dtype=torch.float64, device=dtheta.device)
        if dst:
            key = ("scatter index", str(dtheta.device), tuple(dst))
            index = None if derived is None else derived.get(key)
            if index is None:
                index = torch.tensor(dst, dtype=torch.int64).to(dtheta.device)
                if derived is not None:
                    derived[key] = index
            flat.index_copy_(0, index, dtheta)
    else:
        flat = torch.zeros(int(starts[-1]), dtype=torch.float64)
        if dst:
            flat[torch.tensor(dst, dtype=torch.int64)] = dtheta.cpu()
    if param_meta and all(dt == param_meta[0][1] and dev == flat.device for _, dt, dev in param_meta):
        flat = flat.to(param_meta[0][1])   # one conversion; the gradients are views of it
        return [flat[starts[k]:starts[k + 1]].reshape(shape) for k, (shape, _, _) in enumerate(param_meta)]
    return [flat[starts[k]:starts[k + 1]].reshape(shape).to(device=dev, dtype=dt)
            for k, (shape, dt, dev) in enumerate(param_meta)]


class FusedSpotLossFn(torch.autograd.Function):
    """Spot-size loss of a sequential scene as ONE differentiable function of the scene parameters
    (optim/goals.py:144-187): trace + float64 moments + one tangent trace that forms the loss adjoint in the
    kernel (rtt_trace_seq_spot_grad_f32).  No per-ray adjoint or tangent ever reaches HBM.

    The tangent trace is launched by the forward, right behind the moment passes: it needs nothing from the
    backward but the scalar upstream gradient, so d(loss)/d(theta) is kept and the backward only scales it.  The
    device then runs trace, moments and tangent trace back to back while the host is already in the caller's
    code, instead of idling until the autograd engine gets to this function."""

    @staticmethod
    def forward(ctx, rays_in, prog, table, thetas, group, target, sensor_slot, sens, mom, *params):
        table, derived = table   # the slot Jacobians and the dict of device objects derived from them
        hit, w, mask = sens[:3]  # dense record of the sensor slot from the forward trace
        tape = sens[3] if len(sens) > 3 else None   # and what that trace kept for the reverse sweep, if it kept a tape
        x, y = hit[0], hit[1]
        n, dev = len(rays_in), rays_in.device
        ctx.thetas, ctx.derived = thetas, derived
        ctx.param_meta = [(p.shape, p.dtype, p.device) for p in params]
        ctx.param_ids = [id(p) for p in params]
        if backward_mode() != "tangent" and (tape is not None or N.lib().rtt_trace_seq_adjoint_max_slots(int(table.shape[0])) > 0):
            # Reverse mode, without a host round trip: the six moment sums stay on the device.  `mom` holds this rank's
            # first moments (pass 0, launched by the caller); the second pass reads weight and centroid from it and adds
            # the loss sum and the centroid term (Gx, Gy); the reverse sweep reads all of them and yields the slot
            # adjoints, which one small matrix product turns into d(loss)/d(theta).  Three small exchanges per step.
            allreduce_sum_(mom[0:3], group)
            centroid = target is None
            tx, ty = (0.0, 0.0) if centroid else (float(target[0]), float(target[1]))
            _spot_moments(x, y, w, mask, 3 if centroid else 4, tx, ty, 1.0, mom)
            allreduce_sum_(mom[3:6], group)
            # nothing lit on any rank: every sum is zero and no ray carries a seed, so loss and gradients come out as
            # zeros by themselves; `lit` only tells the caller not to count this bundle (optim/goals.py:165-167)
            dtheta, _ = _adjoint_sweep(rays_in, prog, table, spot=(sensor_slot, centroid, tx, ty, mom), derived=derived,
                                       tape=tape, spot_w=w if tape is not None else None)
            allreduce_sum_(dtheta, group)
            ctx.dtheta = dtheta
            out = mom.to(torch.float32)   # one conversion: the loss and the lit weight are views of it
            lit = out[0].sign()
            ctx.mark_non_differentiable(lit)
            return out[3], lit
        # everything the tangent launches need that does not depend on the centroid goes to the device now, while it
        # is still busy with the trace and the first moments: after the synchronisation below only launches remain
        dp = _tangent_program(prog, table, dev) if n else None
        staged = []
        for k0, used, padded in _chunks(table):
            width = padded.shape[2]
            staged.append((k0, used, width, torch.from_numpy(padded).to(dev) if n else None,
                           torch.zeros(3 * width + 3, dtype=torch.float64, device=dev)))   # the kernel fills 3 width + 2
        allreduce_sum_(mom, group)   # `mom` holds this rank's first moments (pass 0, launched by the caller)
        W, Sx, Sy = mom.tolist()[:3]
        ctx.lit = W > 0.0
        W = max(W, 1e-12)
        centroid = target is None
        cx, cy = (Sx / W, Sy / W) if centroid else (float(target[0]), float(target[1]))
        mom = mom[:4].clone()
        mom[3] = 0.0
        _spot_moments(x, y, w, mask, 1, cx, cy, W, mom)
        loss = None   # this rank's sum; it travels with the first gradient chunk, one exchange for both
        dtheta = torch.zeros(table.shape[2], dtype=torch.float64, device=dev)
        if ctx.lit:   # the same decision on every rank (W is the global sum): the exchanges below pair up
            for k0, used, width, tan_dev, out in staged:
                if n:   # a rank whose shard is empty contributes zeros
                    with torch.cuda.device(dev):
                        code = N.lib().rtt_trace_seq_spot_grad_f32(
                            N.ptr(rays_in._pos), N.ptr(rays_in._dir), N.ptr(rays_in.intensity), n, n,
                            N.ptr(dp.dev), dp.host_ptr, dp.n, N.ptr(tan_dev), width, int(sensor_slot), cx, cy, W,
                            N.ptr(out), N.stream_handle(dev))
                    N.check(code, "rtt_trace_seq_spot_grad_f32")
                if loss is None:
                    out[3 * width + 2] = mom[3]
                allreduce_sum_(out, group)
                if loss is None:
                    loss = out[3 * width + 2:3 * width + 3].clone()
                A, Bx, By = out[:width], out[width:2 * width], out[2 * width:3 * width]
                d = A - out[3 * width] * Bx - out[3 * width + 1] * By if centroid else A
                dtheta[k0:k0 + used] = d[:used]
        if loss is None:   # nothing lit reached the sensor on any rank
            loss = allreduce_sum_(mom[3:4].clone(), group)
        ctx.dtheta, ctx.thetas = dtheta, thetas
        ctx.param_meta = [(p.shape, p.dtype, p.device) for p in params]
        ctx.param_ids = [id(p) for p in params]
        lit = torch.tensor(1.0 if ctx.lit else 0.0, device=dev)
        ctx.mark_non_differentiable(lit)
        return loss.reshape(()).to(torch.float32), lit

    @staticmethod
    def backward(ctx, g, _glit=None):
        grads = _assemble_grads(ctx.dtheta * g.to(torch.float64), ctx.thetas, ctx.param_meta, ctx.param_ids, ctx.derived)
        return (None,) * 9 + tuple(grads)


_scalars_cache = {}


def _cached_scalars(elements, params=None):
    """trainable_scalars(elements), re-derived only when the elements or their requires_grad flags change (walking
    the module trees costs as much as launching a trace; a scene passes the parameter list it already keeps)."""
    import weakref
    if params is None:
        params = [p for el in elements for p in el.parameters()]
    key = (tuple(id(el) for el in elements), tuple(id(p) for p in params), tuple(p.requires_grad for p in params))
    hit = _scalars_cache.get(key[0])
    if hit is None or hit[0] != key or any(r() is None for r in hit[1]):
        if len(_scalars_cache) > 64:
            _scalars_cache.clear()
        hit = _scalars_cache[key[0]] = (key, [weakref.ref(p) for p in params], trainable_scalars(elements))
    return hit[2]


def fused_spot_loss(elements, rays, prog, sensor_slot, sens, target=None, group=None, params=None, table_cache=None,
                    mom=None):
    """Differentiable spot-size loss of a bundle already traced (forward only, no graph) through `prog`;
    `sens` = (hit [3,N], w [N], mask [N]) of the sensor slot, `rays` the *input* bundle of that trace.
    Returns (loss, lit): `lit` is a 0/1 tensor on the device, 0 when nothing lit reached the sensor on any rank (the
    reference skips such a bundle, optim/goals.py:165-167; the loss is then 0 with zero gradients).  Neither is read
    on the host here, so the caller's step stays asynchronous."""
    hit, w, mask = sens[:3]
    if mom is None:   # (a caller whose trace already summed the first moments hands them in: six float64, three filled)
        # first moments right behind the trace; the host work below overlaps both kernels
        mom = torch.zeros(6, device=w.device, dtype=torch.float64)
        _spot_moments(hit[0], hit[1], w, mask, 0, 0.0, 0.0, 1.0, mom)
    thetas = _cached_scalars(elements, params)
    # `table_cache` = (dict owned by the scene, key of the scene's parameter state): the slot Jacobians are derived
    # again only when a parameter changed (for device-resident parameters deriving them costs a device-to-host
    # transfer, i.e. a synchronisation)
    table = derived = None
    if table_cache is not None:
        holder, state = table_cache
        key = (state, tuple((id(p), i) for p, i in thetas))
        kept = holder.get("_tangent_table")
        if kept is not None and kept[0] == key:
            table, derived = kept[1], kept[2]
    if table is None:
        table, derived = tangent_table(elements, thetas), {}
        if table_cache is not None:
            holder["_tangent_table"] = (key, table, derived)
    params, seen = [], set()
    for p, _ in thetas:
        if id(p) not in seen:
            seen.add(id(p))
            params.append(p)
    return FusedSpotLossFn.apply(rays, prog, (table, derived), thetas, group, target, sensor_slot, sens, mom, *params)
