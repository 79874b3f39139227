"""Python launchers for the C-ABI entry points of ``librtt_hip.so``.

Each function validates devices / shapes / dtypes on the host, allocates outputs with torch (the
C side never allocates) and enqueues one kernel on the current torch stream of the rays' device.
"""
import weakref

import numpy as np
import torch

from . import _native as N
from . import program as P
from .rays.ray import Rays


_checked = [(None, None, None, -1), (None, None, None, -1)]   # the last two plane triples that passed (weak refs) + ray count


def _check_rays(rays: Rays):
    p, d, w = rays._pos, rays._dir, rays.intensity
    for c in _checked:
        # the very tensors of an earlier call (a stepping loop, an optimisation loop, an in/out pair): already checked
        # (a tensor object keeps its identity when it is given other data, `t.data = ...`: what would let a launch read
        # host memory or another type is looked at again, it costs a few attribute reads)
        if (c[0] is not None and c[0]() is p and c[1]() is d and c[2]() is w and p.shape[1] == c[3]
                and p.is_cuda and d.is_cuda and w.is_cuda and p.dtype is torch.float32 and d.dtype is torch.float32
                and w.dtype is torch.float32 and w.shape[0] == c[3] and d.shape[1] == c[3]):
            return c[3]
    n = _check_rays_full(rays, p, d, w)
    _checked[1] = _checked[0]
    _checked[0] = (weakref.ref(p), weakref.ref(d), weakref.ref(w), n)
    return n


def _check_rays_full(rays, p, d, w):
    n = p.shape[1] if p.ndim == 2 else -1
    f32 = torch.float32
    if (p.dtype is f32 and d.dtype is f32 and w.dtype is f32 and p.is_cuda and d.is_cuda and w.is_cuda
            and p.shape == (3, n) and d.shape == (3, n) and w.shape == (n,)
            and p.is_contiguous() and d.is_contiguous() and w.is_contiguous()):
        return n   # the usual case, decided without building messages
    N.require_device(rays._pos, rays._dir, rays.intensity)
    n = len(rays)
    for name, t, shape in (("pos", rays._pos, (3, n)), ("dir", rays._dir, (3, n)),
                           ("intensity", rays.intensity, (n,))):
        if t.dtype != torch.float32:
            raise TypeError(f"rays.{name} must be float32 for the HIP path, got {t.dtype}")
        if tuple(t.shape) != shape or not t.is_contiguous():
            raise ValueError(f"rays.{name} must be a contiguous tensor of shape {shape}")
    return n


def _call_on(dev, fn, *args):
    """Call an entry point with `dev` as the current device (launches go to the current device); the switch is made
    only when another device is current, the context manager costs as much as a small launch."""
    if dev.index is None or torch.cuda.current_device() == dev.index:
        return fn(*args)
    with torch.cuda.device(dev):
        return fn(*args)


def _sensor_buffers(prog, n, device, zero_mask=True):
    """Dense sensor record of one launch.  The sequential trace writes the mask entry of every ray on every sensor
    surface (`zero_mask=False`); a nearest-hit bounce only those of the rays that went to a sensor."""
    k = P.n_sensor_slots(prog)
    if k == 0:
        return None, None, None
    hit = torch.empty((k, 3, n), device=device, dtype=torch.float32)
    w = torch.empty((k, n), device=device, dtype=torch.float32)
    mask = (torch.zeros if zero_mask else torch.empty)((k, n), device=device, dtype=torch.uint8)
    return hit, w, mask


class SensorRecord:
    """Dense sensor record of one launch in ONE allocation: hit planes [K,3,N] float32, incoming intensities [K,N]
    float32 and hit masks [K,N] uint8 behind one another.  The launch only needs the three addresses; the tensor
    views are made when somebody reads them (a view costs as much as an allocation, and a small trace is all host
    time)."""
    __slots__ = ("buf", "k", "n", "_views")

    def __init__(self, k, n, device):
        self.k, self.n = k, n
        self.buf = torch.empty((4 * k * n + (k * n + 3) // 4,), device=device, dtype=torch.float32)
        self._views = None

    def pointers(self):
        base = self.buf.data_ptr()
        return base, base + 12 * self.k * self.n, base + 16 * self.k * self.n

    def views(self):
        """(hit [K,3,N], w [K,N], mask [K,N] uint8)"""
        v = self._views
        if v is None:
            k, n, f = self.k, self.n, self.buf
            v = self._views = (f[:3 * k * n].view(k, 3, n), f[3 * k * n:4 * k * n].view(k, n),
                               f[4 * k * n:].view(torch.uint8)[:k * n].view(k, n))
        return v


class SensorVisit:
    """What one sensor slot recorded in one launch, as the 4-sequence (hit planes [3,N], w [N], mask [N], ids [N]) the
    sensors keep per visit; the views into the launch's record are made on first use."""
    __slots__ = ("record", "slot", "ids", "_items")

    def __init__(self, record, slot, ids):
        self.record, self.slot, self.ids, self._items = record, slot, ids, None

    def _resolve(self):
        it = self._items
        if it is None:
            hit, w, mask = self.record.views()
            it = self._items = (hit[self.slot], w[self.slot], mask[self.slot], self.ids)
        return it

    def __iter__(self):
        return iter(self._resolve())

    def __getitem__(self, i):
        return self._resolve()[i]

    def __len__(self):
        return 4

    @property
    def n(self):
        return self.record.n


def _prepared(prog, kind, src, dst, n, dev):
    """The prepared launch of `prog` (a `Program` with facts) on the planes of `src` -> `dst`: found again by the identity
    of the program and of the six plane tensors (a stepping or optimisation loop passes the same objects every time),
    prepared on first use.  A few per program are kept; they go with the program.  Returns None for a plain record array."""
    facts = P.facts_of(prog)
    if facts is None:
        return None
    kept = facts.get(("launches", kind))
    if kept is None:
        kept = facts[("launches", kind)] = []
    a, b, c, d, e, f = src._pos, src._dir, src.intensity, dst._pos, dst._dir, dst.intensity
    for entry in kept:
        r = entry[0]
        if r[0]() is a and r[1]() is b and r[2]() is c and r[3]() is d and r[4]() is e and r[5]() is f and entry[1] == n:
            # the handle holds addresses: a tensor object whose storage was exchanged (`t.data = ...`, `set_`) is another launch
            if entry[4] == (a.data_ptr(), b.data_ptr(), c.data_ptr(), d.data_ptr(), e.data_ptr(), f.data_ptr()):
                return entry[2]
            kept.remove(entry)
            break
    dp = N.DeviceProgram.upload(prog, dev)
    lib = N.lib()
    where = (a.data_ptr(), b.data_ptr(), c.data_ptr(), d.data_ptr(), e.data_ptr(), f.data_ptr())
    with torch.cuda.device(dev):
        handle = N.PreparedLaunch(lib.rtt_prepare_step_nearest_f32 if kind else lib.rtt_prepare_trace_seq_f32,
                                  where[0], where[1], where[2], n, where[3], where[4], where[5], n, n,
                                  dp.dev.data_ptr(), dp.host_ptr, dp.n, n)
    if len(kept) >= 4:
        del kept[0]
    kept.append((tuple(weakref.ref(t) for t in (a, b, c, d, e, f)), n, handle, dp, where))   # (dp: the device copy the handle reads)
    return handle


def _launch_prepared(handle, rec, dev, what, retry):
    code = _call_on(dev, handle.launch, rec.buf.data_ptr() if rec is not None else None, N.stream_handle(dev))
    if code == -3 and retry is not None:   # prepared under other rtt_tune settings: forget the handles and prepare again
        return retry()
    if code:
        N.check(code, what)
    return rec


def _trace_pieces(rays, prog, out, n, dev, hit_p, w_p, mask_p):
    """SequentialScene.simulate of a program longer than one launch takes: the pieces of `P.split_program` one after the
    other, the first from `rays` into `out`, the others in place in `out`; every piece records into the planes of its
    own sensor slots (hit_p, w_p, mask_p: addresses of the whole record's [K,3,N], [K,N], [K,N] planes, 0 = none)."""
    lib = N.lib()
    src = rays
    for piece, _, slots in P.split_program(prog):
        dp = N.DeviceProgram.upload(piece, dev)
        has = P.n_sensor_slots(piece) > 0
        code = _call_on(dev, lib.rtt_trace_seq_f32,
                        src._pos.data_ptr(), src._dir.data_ptr(), src.intensity.data_ptr(), n,
                        out._pos.data_ptr(), out._dir.data_ptr(), out.intensity.data_ptr(), n, n,
                        dp.dev.data_ptr(), dp.host_ptr, dp.n,
                        hit_p + 12 * slots * n if has else None, w_p + 4 * slots * n if has else None,
                        mask_p + slots * n if has else None, n, N.stream_handle(dev))
        if code:
            N.check(code, "rtt_trace_seq_f32")
        src = out


def trace_sequential_record(rays: Rays, prog: np.ndarray, out: Rays = None):
    """trace_sequential for the plain call (no moments): returns the launch's SensorRecord, or None for a program
    without sensor slots."""
    n = _check_rays(rays)
    if out is None:
        out = rays
    elif _check_rays(out) != n:
        raise ValueError("`out` must hold the same number of rays")
    k = P.n_sensor_slots(prog)
    if n == 0 or prog.shape[0] == 0:
        if out is not rays:
            out._pos.copy_(rays._pos); out._dir.copy_(rays._dir); out.intensity.copy_(rays.intensity)
        return SensorRecord(k, n, rays.device) if k else None
    dev = rays.device
    if prog.shape[0] > P.MAX_SURFACES:   # a launch takes 64 surfaces: one launch per piece, each continuing in `out`
        rec = SensorRecord(k, n, dev) if k else None
        _trace_pieces(rays, prog, out, n, dev, *(rec.pointers() if k else (0, 0, 0)))
        return rec
    handle = _prepared(prog, 0, rays, out, n, dev)
    if handle is not None:
        rec = SensorRecord(k, n, dev) if k else None

        def again():
            P.facts_of(prog).pop(("launches", 0), None)
            return _launch_prepared(_prepared(prog, 0, rays, out, n, dev), rec, dev, "rtt_launch_prepared", None)
        return _launch_prepared(handle, rec, dev, "rtt_launch_prepared", again)
    dp = N.DeviceProgram.upload(prog, dev)
    if k:
        rec = SensorRecord(k, n, dev)
        hit_p, w_p, mask_p = rec.pointers()
    else:
        rec, hit_p, w_p, mask_p = None, None, None, None
    code = _call_on(dev, N.lib().rtt_trace_seq_f32,
                    rays._pos.data_ptr(), rays._dir.data_ptr(), rays.intensity.data_ptr(), n,
                    out._pos.data_ptr(), out._dir.data_ptr(), out.intensity.data_ptr(), n, n,
                    dp.dev.data_ptr(), dp.host_ptr, dp.n, hit_p, w_p, mask_p, n, N.stream_handle(dev))
    if code:
        N.check(code, "rtt_trace_seq_f32")
    return rec


class Tape:
    """What a trace kept for its reverse sweep (`rtt_trace_seq_tape_f32`): an opaque device buffer, valid for the rays
    and the program of that trace."""
    __slots__ = ("buf", "n", "n_surf")

    def __init__(self, n, n_surf, device):
        self.n, self.n_surf = n, n_surf
        self.buf = torch.empty(int(N.lib().rtt_trace_seq_tape_bytes(n, n_surf)), dtype=torch.uint8, device=device)


def tape_slots(n_surf: int) -> int:
    """Slot sums one reverse sweep from a tape can return for a program of `n_surf` surfaces; 0 = such a program is not
    swept from a tape."""
    return int(N.lib().rtt_trace_seq_adjoint_tape_max_slots(int(n_surf))) if 1 <= n_surf <= P.MAX_SURFACES else 0


def trace_sequential(rays: Rays, prog: np.ndarray, out: Rays = None, moments=None, store: bool = True, tape: bool = False):
    """SequentialScene.simulate on the device.  Writes into `out` (default: in place into `rays`).
    Returns (sens_hit [K,3,N], sens_w [K,N], sens_mask [K,N] uint8) or (None, None, None).
    `moments` = (sensor slot, float64 device tensor of at least 3 entries): the same launch also leaves the first
    spot-size moments (sum w, sum w x, sum w y over the lit hits of that slot) in the tensor's first three entries.
    With `store=False` (moments only) the traced rays are not written: `rays` keeps its planes and `out` is ignored.
    With `tape` the launch also keeps what the reverse sweep needs and a fourth value is returned, the `Tape` (None for
    an empty bundle or program); the result must then go out of place (`out` is not `rays`), since the sweep reads the
    input positions."""
    if tape:
        return _trace_sequential_tape(rays, prog, out, moments, store)
    n = _check_rays(rays)
    if not store:
        if moments is None:
            raise ValueError("only the trace that sums moments can leave the traced rays unwritten")
        out = None
    elif out is None:
        out = rays
    elif _check_rays(out) != n:
        raise ValueError("`out` must hold the same number of rays")
    dev = rays.device
    sens_hit, sens_w, sens_mask = _sensor_buffers(prog, n, dev, zero_mask=False)
    if n == 0 or prog.shape[0] == 0:
        if out is not None and out is not rays:
            out._pos.copy_(rays._pos); out._dir.copy_(rays._dir); out.intensity.copy_(rays.intensity)
        if moments is not None:
            moments[1][:3] = 0.0
        return sens_hit, sens_w, sens_mask
    if prog.shape[0] > P.MAX_SURFACES:
        if moments is not None:
            raise ValueError("the trace that sums moments is one launch: at most %d surfaces" % P.MAX_SURFACES)
        _trace_pieces(rays, prog, out, n, dev, N.ptr(sens_hit) or 0, N.ptr(sens_w) or 0, N.ptr(sens_mask) or 0)
        return sens_hit, sens_w, sens_mask
    dp = N.DeviceProgram.upload(prog, dev)
    if moments is not None:
        slot, mom = moments
        if mom.dtype != torch.float64 or mom.numel() < 3 or not mom.is_contiguous() or mom.device != dev:
            raise ValueError("`moments` needs a contiguous float64 tensor of at least 3 entries on the rays' device")
        ws_bytes = int(N.lib().rtt_trace_seq_moments_workspace_bytes(n))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            code = N.lib().rtt_trace_seq_moments_f32(
                N.ptr(rays._pos), N.ptr(rays._dir), N.ptr(rays.intensity), n,
                N.ptr(out._pos) if store else None, N.ptr(out._dir) if store else None,
                N.ptr(out.intensity) if store else None, n, n,
                N.ptr(dp.dev), dp.host_ptr, dp.n,
                N.ptr(sens_hit), N.ptr(sens_w), N.ptr(sens_mask), n, int(slot), N.ptr(mom), N.ptr(ws), ws_bytes,
                N.stream_handle(dev))
        N.check(code, "rtt_trace_seq_moments_f32")
        return sens_hit, sens_w, sens_mask
    code = _call_on(dev, N.lib().rtt_trace_seq_f32,
                    rays._pos.data_ptr(), rays._dir.data_ptr(), rays.intensity.data_ptr(), n,
                    out._pos.data_ptr(), out._dir.data_ptr(), out.intensity.data_ptr(), n, n,
                    dp.dev.data_ptr(), dp.host_ptr, dp.n,
                    N.ptr(sens_hit), N.ptr(sens_w), N.ptr(sens_mask), n, N.stream_handle(dev))
    if code:
        N.check(code, "rtt_trace_seq_f32")
    return sens_hit, sens_w, sens_mask


def _trace_sequential_tape(rays, prog, out, moments, store):
    n = _check_rays(rays)
    if not store:
        if moments is None:
            raise ValueError("only the trace that sums moments can leave the traced rays unwritten")
        out = None
    else:
        if out is None or out is rays or out._pos is rays._pos:
            raise ValueError("a trace that keeps a tape writes its result out of place: the reverse sweep reads the input positions")
        if _check_rays(out) != n:
            raise ValueError("`out` must hold the same number of rays")
    dev = rays.device
    sens_hit, sens_w, sens_mask = _sensor_buffers(prog, n, dev, zero_mask=False)
    if n == 0 or prog.shape[0] == 0:
        if out is not None:
            out._pos.copy_(rays._pos); out._dir.copy_(rays._dir); out.intensity.copy_(rays.intensity)
        if moments is not None:
            moments[1][:3] = 0.0
        return sens_hit, sens_w, sens_mask, None
    dp = N.DeviceProgram.upload(prog, dev)
    lib = N.lib()
    kept = Tape(n, dp.n, dev)
    slot, mom, ws, ws_bytes = -1, None, None, 0
    if moments is not None:
        slot, mom = moments
        if mom.dtype != torch.float64 or mom.numel() < 3 or not mom.is_contiguous() or mom.device != dev:
            raise ValueError("`moments` needs a contiguous float64 tensor of at least 3 entries on the rays' device")
        ws_bytes = int(lib.rtt_trace_seq_moments_workspace_bytes(n))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        code = lib.rtt_trace_seq_tape_f32(
            N.ptr(rays._pos), N.ptr(rays._dir), N.ptr(rays.intensity), n,
            N.ptr(out._pos) if store else None, N.ptr(out._dir) if store else None, N.ptr(out.intensity) if store else None, n, n,
            N.ptr(dp.dev), dp.host_ptr, dp.n, N.ptr(sens_hit), N.ptr(sens_w), N.ptr(sens_mask), n, int(slot), N.ptr(mom),
            N.ptr(ws), ws_bytes, N.ptr(kept.buf), kept.buf.numel(), N.stream_handle(dev))
    N.check(code, "rtt_trace_seq_tape_f32")
    return sens_hit, sens_w, sens_mask, kept


def trace_sequential_history(rays: Rays, prog: np.ndarray):
    """SequentialScene.simulate in place plus the position of every ray after each surface.
    Returns (sens_hit, sens_w, sens_mask, hist [S,3,N], hit_any [S] int32)."""
    n = _check_rays(rays)
    dev = rays.device
    s = prog.shape[0]
    sens_hit, sens_w, sens_mask = _sensor_buffers(prog, n, dev)
    hist = torch.empty((s, 3, n), device=dev, dtype=torch.float32)
    hit_any = torch.zeros((s,), device=dev, dtype=torch.int32)
    if n and s:
        for piece, first, slots in P.split_program(prog):   # (one piece unless the program is longer than a launch takes)
            dp = N.DeviceProgram.upload(piece, dev)
            has = P.n_sensor_slots(piece) > 0
            with torch.cuda.device(dev):
                code = N.lib().rtt_trace_seq_history_f32(
                    N.ptr(rays._pos), N.ptr(rays._dir), N.ptr(rays.intensity), n,
                    N.ptr(rays._pos), N.ptr(rays._dir), N.ptr(rays.intensity), n, n,
                    N.ptr(dp.dev), dp.host_ptr, dp.n,
                    N.ptr(sens_hit[slots:]) if has else None, N.ptr(sens_w[slots:]) if has else None,
                    N.ptr(sens_mask[slots:]) if has else None, n,
                    N.ptr(hist[first:]), N.ptr(hit_any[first:]), n, N.stream_handle(dev))
            N.check(code, "rtt_trace_seq_history_f32")
    return sens_hit, sens_w, sens_mask, hist, hit_any


def intersect_test(rays: Rays, prog: np.ndarray) -> torch.Tensor:
    """[N, S] matrix of distances (inf = miss), returned as a transposed view of [S, N] planes."""
    n = _check_rays(rays)
    s = prog.shape[0]
    t = torch.empty((s, n), device=rays.device, dtype=torch.float32)
    if n and s:
        for piece, first, _ in P.split_program(prog):   # (one piece unless the program is longer than a launch takes)
            dp = N.DeviceProgram.upload(piece, rays.device)
            with torch.cuda.device(rays.device):
                code = N.lib().rtt_intersect_test_f32(
                    N.ptr(rays._pos), N.ptr(rays._dir), n, n, N.ptr(dp.dev), dp.host_ptr, dp.n,
                    N.ptr(t[first:]), n, N.stream_handle(rays.device))
            N.check(code, "rtt_intersect_test_f32")
    return t.t()


def surface_forward(rays: Rays, prog: np.ndarray, index: int, want_phys: bool):
    """Forward of ONE surface on every ray.  Returns dict with t [N], hit_global / normal_global /
    hit_local [N,3] views, and when `want_phys` new_dir [N,3], imult [N]."""
    n = _check_rays(rays)
    dev = rays.device
    mk3 = lambda: torch.empty((3, n), device=dev, dtype=torch.float32)
    t = torch.empty((n,), device=dev, dtype=torch.float32)
    hit_g, nrm, hit_l = mk3(), mk3(), mk3()
    new_dir = mk3() if want_phys else None
    imult = torch.empty((n,), device=dev, dtype=torch.float32) if want_phys else None
    if n:
        dp = N.DeviceProgram.upload(prog, dev)
        with torch.cuda.device(dev):
            code = N.lib().rtt_surface_forward_f32(
                N.ptr(rays._pos), N.ptr(rays._dir), n, n, N.ptr(dp.dev), dp.host_ptr, int(index),
                N.ptr(t), N.ptr(hit_g), N.ptr(nrm), N.ptr(hit_l), N.ptr(new_dir), N.ptr(imult), n,
                N.stream_handle(dev))
        N.check(code, "rtt_surface_forward_f32")
    out = dict(t=t, hit_global=hit_g.t(), normal_global=nrm.t(), hit_local=hit_l.t())
    if want_phys:
        out.update(new_dir=new_dir.t(), imult=imult)
    return out


def ray_cast(rays: Rays, prog: np.ndarray):
    """winner [N] int32 (-1 = miss) and tmin [N]."""
    n = _check_rays(rays)
    dev = rays.device
    if prog.shape[0] > P.MAX_SURFACES:
        return _ray_cast_pieces(rays, prog, n, dev)
    winner = torch.full((n,), -1, device=dev, dtype=torch.int32)
    tmin = torch.full((n,), float("inf"), device=dev, dtype=torch.float32)
    if n and prog.shape[0]:
        dp = N.DeviceProgram.upload(prog, dev)
        with torch.cuda.device(dev):
            code = N.lib().rtt_ray_cast_f32(
                N.ptr(rays._pos), N.ptr(rays._dir), n, n, N.ptr(dp.dev), dp.host_ptr, dp.n,
                N.ptr(winner), N.ptr(tmin), N.stream_handle(dev))
        N.check(code, "rtt_ray_cast_f32")
    return winner, tmin


def _ray_cast_pieces(rays, prog, n, dev):
    """Scene.ray_cast over more surfaces than one launch takes: a cast per piece of the program, then the row minimum
    over the pieces by the rule of torch.min over the whole row (scene/base.py:165-170): the smaller distance wins, the
    earlier surface on equal distances, and a NaN beats every number (the first one wins)."""
    winner = torch.full((n,), -1, device=dev, dtype=torch.int32)
    tmin = torch.full((n,), float("inf"), device=dev, dtype=torch.float32)
    for piece, first, _ in P.split_program(prog):
        w, t = ray_cast(rays, piece)
        t_nan, m_nan = torch.isnan(t), torch.isnan(tmin)
        take = (t_nan & ~m_nan) | (~t_nan & ~m_nan & (t < tmin))
        tmin = torch.where(take, t, tmin)
        winner = torch.where(take, w + first, winner)
    return torch.where(tmin < float("inf"), winner, torch.full_like(winner, -1)), tmin


def _step_pieces(rays, prog, out, n, dev, want_winner):
    """Scene.step over more surfaces than one launch takes.  The cast over all pieces names every ray's surface; then one
    bounce per piece on the rays whose surface lies in it (the others ride along with intensity 0, which a bounce leaves
    alone), and the moved rays of all pieces are put together."""
    winner, _ = _ray_cast_pieces(rays, prog, n, dev)
    sens_hit, sens_w, sens_mask = _sensor_buffers(prog, n, dev, zero_mask=True)
    new_pos, new_dir, new_int = rays._pos.clone(), rays._dir.clone(), rays.intensity.clone()
    lib = N.lib()
    for piece, first, slots in P.split_program(prog):
        mine = (winner >= first) & (winner < first + piece.shape[0])
        lit = torch.where(mine, rays.intensity, torch.zeros_like(rays.intensity))
        pos_c, dir_c, int_c = torch.empty_like(rays._pos), torch.empty_like(rays._dir), torch.empty_like(lit)
        dp = N.DeviceProgram.upload(piece, dev)
        has = P.n_sensor_slots(piece) > 0
        with torch.cuda.device(dev):
            code = lib.rtt_step_nearest_f32(
                N.ptr(rays._pos), N.ptr(rays._dir), N.ptr(lit), n, N.ptr(pos_c), N.ptr(dir_c), N.ptr(int_c), n, n,
                N.ptr(dp.dev), dp.host_ptr, dp.n, None,
                N.ptr(sens_hit[slots:]) if has else None, N.ptr(sens_w[slots:]) if has else None,
                N.ptr(sens_mask[slots:]) if has else None, n, N.stream_handle(dev))
        N.check(code, "rtt_step_nearest_f32")
        new_pos = torch.where(mine[None, :], pos_c, new_pos)
        new_dir = torch.where(mine[None, :], dir_c, new_dir)
        new_int = torch.where(mine, int_c, new_int)
    out._pos.copy_(new_pos); out._dir.copy_(new_dir); out.intensity.copy_(new_int)
    return (winner if want_winner else None), sens_hit, sens_w, sens_mask


def step_nearest_record(rays: Rays, prog: np.ndarray):
    """One Scene.step bounce in place, the plain call (no winners wanted): returns the launch's SensorRecord, or None
    for a program without sensor slots.  The entry point clears the record's masks itself."""
    n = _check_rays(rays)
    k = P.n_sensor_slots(prog)
    dev = rays.device
    if n == 0 or prog.shape[0] == 0:
        rec = SensorRecord(k, n, dev) if k else None
        if rec is not None:
            rec.views()[2].zero_()
        return rec
    handle = _prepared(prog, 1, rays, rays, n, dev)
    if handle is not None:
        rec = SensorRecord(k, n, dev) if k else None

        def again():
            P.facts_of(prog).pop(("launches", 1), None)
            return _launch_prepared(_prepared(prog, 1, rays, rays, n, dev), rec, dev, "rtt_launch_prepared", None)
        return _launch_prepared(handle, rec, dev, "rtt_launch_prepared", again)
    dp = N.DeviceProgram.upload(prog, dev)
    if k:
        rec = SensorRecord(k, n, dev)
        hit_p, w_p, mask_p = rec.pointers()
    else:
        rec, hit_p, w_p, mask_p = None, None, None, None
    pos, d, inten = rays._pos.data_ptr(), rays._dir.data_ptr(), rays.intensity.data_ptr()
    code = _call_on(dev, N.lib().rtt_step_nearest_f32, pos, d, inten, n, pos, d, inten, n, n,
                    dp.dev.data_ptr(), dp.host_ptr, dp.n, None, hit_p, w_p, mask_p, n, N.stream_handle(dev))
    if code:
        N.check(code, "rtt_step_nearest_f32")
    return rec


def step_nearest(rays: Rays, prog: np.ndarray, out: Rays = None, want_winner: bool = False):
    """One Scene.step bounce on the device (in place by default)."""
    n = _check_rays(rays)
    if out is None:
        out = rays
    elif _check_rays(out) != n:
        raise ValueError("`out` must hold the same number of rays")
    dev = rays.device
    if n and prog.shape[0] > P.MAX_SURFACES:
        return _step_pieces(rays, prog, out, n, dev, want_winner)
    launches = bool(n and prog.shape[0])
    sens_hit, sens_w, sens_mask = _sensor_buffers(prog, n, dev, zero_mask=not launches)   # the entry point clears the masks
    winner = torch.full((n,), -1, device=dev, dtype=torch.int32) if want_winner else None
    if launches:
        dp = N.DeviceProgram.upload(prog, dev)
        with torch.cuda.device(dev):
            code = N.lib().rtt_step_nearest_f32(
                N.ptr(rays._pos), N.ptr(rays._dir), N.ptr(rays.intensity), n,
                N.ptr(out._pos), N.ptr(out._dir), N.ptr(out.intensity), n, n,
                N.ptr(dp.dev), dp.host_ptr, dp.n, N.ptr(winner),
                N.ptr(sens_hit), N.ptr(sens_w), N.ptr(sens_mask), n, N.stream_handle(dev))
        N.check(code, "rtt_step_nearest_f32")
    elif out is not rays:
        out._pos.copy_(rays._pos); out._dir.copy_(rays._dir); out.intensity.copy_(rays.intensity)
    return winner, sens_hit, sens_w, sens_mask


HIT_COUNTER_STRIDE = 32   # int32 entries between the counters of two parts of the hit list (the C interface's layout)


def hit_list_parts(n) -> int:
    """Parts of the sensor hit list of a fused bounce loop: with many rays, thousands of wavefronts meet a sensor in
    the same bounce and a single counter makes them queue; a few thousand rays fill only a few blocks, whose parts
    would overflow for nothing."""
    return 64 if n >= (1 << 16) else 1


def _launch_simulate_nearest(src, out, dp, n, max_bounces, bounces, cap, dev):
    """One launch of the fused bounce loop from the planes `src` into the planes `out` (both (pos, dir, intensity));
    returns the append list (hit [parts * cap_p, 4], keys [parts * cap_p, 2], count [parts, 32], parts) or Nones for a
    program without sensors; `cap` is the capacity of the whole list."""
    has_sensor = cap is not None
    parts = hit_list_parts(n) if has_sensor else 1
    if has_sensor:
        cap = max(parts, (cap + parts - 1) // parts * parts)
    hit = torch.empty((cap, 4), device=dev, dtype=torch.float32) if has_sensor else None
    keys = torch.empty((cap, 2), device=dev, dtype=torch.int32) if has_sensor else None
    count = torch.zeros((parts, HIT_COUNTER_STRIDE), device=dev, dtype=torch.int32) if has_sensor else None
    args = (N.ptr(src[0]), N.ptr(src[1]), N.ptr(src[2]), n, N.ptr(out[0]), N.ptr(out[1]), N.ptr(out[2]), n, n,
            N.ptr(dp.dev), dp.host_ptr, dp.n, int(max_bounces), N.ptr(bounces),
            N.ptr(hit), N.ptr(keys), N.ptr(count), cap if has_sensor else 0, parts)
    with torch.cuda.device(dev):
        if staged_bounce_loop(n, max_bounces, dp.host):
            ws_bytes = int(N.lib().rtt_simulate_nearest_staged_workspace_bytes(n))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            code = N.lib().rtt_simulate_nearest_staged_f32(*args, N.ptr(ws), ws_bytes, N.stream_handle(dev))
            N.check(code, "rtt_simulate_nearest_staged_f32")
        else:
            code = N.lib().rtt_simulate_nearest_f32(*args, N.stream_handle(dev))
            N.check(code, "rtt_simulate_nearest_f32")
    return hit, keys, count, parts


STAGED_MIN_RAYS = 1 << 17    # below, the launches of the stages cost more than the thinned-out wavefronts


def staged_bounce_loop(n, max_bounces, prog) -> bool:
    """Whether the bounce loop runs in stages (rays that still move are packed together every few bounces): many rays,
    more bounces than the first stage holds, and no functor whose random draws are keyed by the place in the launch."""
    return n >= STAGED_MIN_RAYS and max_bounces > 5 and not bool((prog["phys_kind"] == P.PHYS_FRESNEL).any())


class PendingHits:
    """The sensor crossings of one fused bounce loop, still as the launch left them: an unsorted append list and its
    counter in device memory.  Nothing has waited for the launch yet.  `resolve()` reads the count (the first host
    synchronisation), repeats the launch from the kept input planes with a longer list should the first one have been
    too short — the rays the loop produced do not depend on the list's capacity, only the list itself is cut — and
    returns (hit [M,4], keys [M,2])."""

    def __init__(self, src, dp, n, max_bounces, cap, dev, lists):
        self.src, self.dp, self.n, self.max_bounces, self.cap, self.dev = src, dp, n, max_bounces, cap, dev
        self.lists = lists
        self.result = None

    def nbytes(self) -> int:
        """Device memory this entry keeps alive until it is resolved or dropped: the lists and the input planes."""
        if self.lists is None:
            return 0
        held = [t for t in self.lists[:3] if t is not None] + [t for t in self.src if t is not None]
        return sum(t.numel() * t.element_size() for t in held)

    def resolve(self):
        if self.result is None:
            hit, keys, count, parts = self.lists
            counts = count[:, 0]
            most = int(counts.max().item())   # the first wait for the launch
            while most > hit.shape[0] // parts:
                # a part was too short: trace again from the untouched input, into scratch planes.  Which part a ray
                # of a later stage appends to depends on the order in which wavefronts finished the stage before, so
                # the parts get some room beyond the fullest one of the last attempt, and the check is repeated.
                scratch = tuple(torch.empty_like(t) for t in self.src)
                room = most if parts == 1 else most + most // 4 + 64
                hit, keys, count, parts = _launch_simulate_nearest(self.src, scratch, self.dp, self.n, self.max_bounces, None,
                                                                   parts * room, self.dev)
                counts = count[:, 0]
                most = int(counts.max().item())
            if parts == 1:
                m = int(counts[0].item())
                hit, keys = hit[:m], keys[:m]
            else:
                cap_p = hit.shape[0] // parts
                filled = (torch.arange(cap_p, device=self.dev)[None, :] < counts[:, None]).reshape(-1)
                hit, keys = hit[filled], keys[filled]
            self.result = (hit, keys)
            self.src = self.lists = None   # the input planes are no longer needed
        return self.result


def simulate_nearest(rays: Rays, prog: np.ndarray, max_bounces: int, want_bounces: bool = False,
                     first_capacity: int = None, defer: bool = False):
    """Scene.simulate in one launch (every ray runs its own bounce loop in registers); `rays` receives new planes.
    Returns (bounces [N] int32 or None, hits) where hits is None for a program without sensors, else
    (hit [M,4] = hit_local xyz + intensity before the surface, keys [M,2] = (bounce*64 + slot, ray index)),
    unsorted.  With `defer` the call does not wait for the device: hits is a `PendingHits` whose `resolve()` gives
    the pair."""
    n = _check_rays(rays)
    dev = rays.device
    bounces = torch.zeros((n,), device=dev, dtype=torch.int32) if want_bounces else None
    has_sensor = P.n_sensor_slots(prog) > 0
    if n == 0 or prog.shape[0] == 0 or max_bounces <= 0:
        empty = (torch.empty((0, 4), device=dev), torch.empty((0, 2), device=dev, dtype=torch.int32))
        return bounces, (empty if has_sensor else None)
    dp = N.DeviceProgram.upload(prog, dev)
    src = (rays._pos, rays._dir, rays.intensity)
    out = (torch.empty_like(rays._pos), torch.empty_like(rays._dir), torch.empty_like(rays.intensity))
    cap = (2 * n if first_capacity is None else int(first_capacity)) if has_sensor else None
    lists = _launch_simulate_nearest(src, out, dp, n, max_bounces, bounces, cap, dev)
    rays._pos, rays._dir, rays.intensity = out
    if not has_sensor:
        return bounces, None
    pending = PendingHits(src, dp, n, max_bounces, cap, dev, lists)
    return bounces, (pending if defer else pending.resolve())


def simulate_nearest_tangent(rays: Rays, dp, max_bounces: int, tangents: np.ndarray, first_capacity: int = None,
                             seed_rays: bool = False):
    """Scene.simulate with dual-number rays over the 2, 4, 6 or 8 parameter directions of `tangents`
    ([n_surf, 32, K] float32), out of place.  Returns (pos [3,N], dir [3,N], intensity [N], hit [M,4], keys [M,2],
    ray_tan [6,K,N], hit_tan [M,3,K]); the hit list is sorted by (bounce, sensor slot, ray), which makes the row order
    the same in every call on the same rays and program.  With `seed_rays` (K >= 6) directions 0..5 are the
    components of each ray's own input, so ray_tan[:, :6] is the per-ray Jacobian."""
    n = _check_rays(rays)
    dev = rays.device
    width = int(tangents.shape[2])
    has_sensor = P.n_sensor_slots(dp.host) > 0
    out_pos, out_dir, out_int = torch.empty_like(rays._pos), torch.empty_like(rays._dir), torch.empty_like(rays.intensity)
    ray_tan = torch.zeros((6, width, n), device=dev, dtype=torch.float32)
    tan_dev = torch.from_numpy(np.ascontiguousarray(tangents, dtype=np.float32)).to(dev)
    cap = (2 * n if first_capacity is None else int(first_capacity)) if has_sensor else 0
    while True:
        hit = torch.empty((cap, 4), device=dev, dtype=torch.float32)
        keys = torch.empty((cap, 2), device=dev, dtype=torch.int32)
        hit_tan = torch.empty((cap, 3, width), device=dev, dtype=torch.float32)
        count = torch.zeros((1,), device=dev, dtype=torch.int32)
        if n and dp.n and max_bounces > 0:
            with torch.cuda.device(dev):
                code = N.lib().rtt_simulate_nearest_tangent_f32(
                    N.ptr(rays._pos), N.ptr(rays._dir), N.ptr(rays.intensity), n,
                    N.ptr(out_pos), N.ptr(out_dir), N.ptr(out_int), n, n,
                    N.ptr(dp.dev), dp.host_ptr, dp.n, int(max_bounces), N.ptr(tan_dev), width, int(bool(seed_rays)), N.ptr(ray_tan),
                    N.ptr(hit), N.ptr(keys), N.ptr(hit_tan), N.ptr(count), cap, N.stream_handle(dev))
            N.check(code, "rtt_simulate_nearest_tangent_f32")
        else:
            out_pos.copy_(rays._pos); out_dir.copy_(rays._dir); out_int.copy_(rays.intensity)
            if seed_rays:
                for c in range(6):
                    ray_tan[c, c] = 1.0
        m = int(count.item())
        if m <= cap:
            break
        cap = m     # more sensor crossings than room: trace again
    hit, keys, hit_tan = hit[:m], keys[:m], hit_tan[:m]
    if m:
        order = torch.argsort(keys[:, 0].to(torch.int64) * (n + 1) + keys[:, 1].to(torch.int64))
        hit, keys, hit_tan = hit[order], keys[order], hit_tan[order]
    return out_pos, out_dir, out_int, hit, keys, ray_tan, hit_tan


def apply_physics(rec: np.ndarray, hit_local, ray_dir, normal):
    """One physics functor on [N,3] tensors -> (new_dir [N,3], imult [N])."""
    N.require_device(hit_local, ray_dir, normal)
    n = ray_dir.shape[0]
    dev = ray_dir.device
    planes = torch.stack([v.to(torch.float32).t() for v in (hit_local, ray_dir, normal)]).contiguous()
    new_dir = torch.empty((3, n), device=dev, dtype=torch.float32)
    imult = torch.empty((n,), device=dev, dtype=torch.float32)
    if n:
        dp = N.DeviceProgram.upload(np.asarray(rec).reshape(1), dev)
        with torch.cuda.device(dev):
            code = N.lib().rtt_apply_physics_f32(
                N.ptr(planes[0]), N.ptr(planes[1]), N.ptr(planes[2]), n, n, N.ptr(dp.dev), dp.host_ptr,
                N.ptr(new_dir), N.ptr(imult), N.stream_handle(dev))
        N.check(code, "rtt_apply_physics_f32")
    return new_dir.t(), imult


def spot_moments(x, y, w, mask, pass_id: int, cx: float, cy: float, wsum: float, out: torch.Tensor):
    """Accumulate spot-size moments into `out` (4 float64 on the device)."""
    N.require_device(x, y, w, out)
    n = x.numel()
    if out.dtype != torch.float64 or out.numel() not in ((6,) if pass_id >= 2 else (4, 6)) or not out.is_contiguous():
        raise ValueError("`out` must be 4 or 6 float64 values (6 for the passes that also sum the centroid term)")
    for t in (x, y, w):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
            raise ValueError("x, y, w must be contiguous float32 tensors of equal length")
    if mask is not None and (mask.dtype != torch.uint8 or mask.numel() != n or not mask.is_contiguous()):
        raise ValueError("mask must be a contiguous uint8 tensor of the same length")
    if n:
        with torch.cuda.device(x.device):
            code = N.lib().rtt_spot_moments_f32(N.ptr(x), N.ptr(y), N.ptr(w), N.ptr(mask), n, int(pass_id),
                                                float(cx), float(cy), float(wsum), N.ptr(out),
                                                N.stream_handle(x.device))
        N.check(code, "rtt_spot_moments_f32")
    return out


SRC_PENCIL, SRC_DISK, SRC_LINE, SRC_FAN, SRC_POINT = range(5)


def sample_bundle(kind: int, par, rot, trans, ray_id: int, seed: int, offset: int, n: int, device,
                  area_kind: int = 0, emit_kind: int = 0, emit_par=()) -> Rays:
    """Bundle.sample(N) on the device: local sample, bundle pose, normalisation, ids (one write-only kernel).
    `rot` is the 3x3 local->global rotation, `trans` the translation; rays get the Philox values of the
    global indices offset .. offset+n-1.  The rays are generated when they are first read (`SampledRays`);
    `SequentialScene.simulate` generates and traces them in one launch."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("sample_bundle generates rays on a HIP device only")
    import ctypes
    from .rays.ray import SampledRays
    src = N.RttSource()
    src.kind, src.ray_id = int(kind), int(ray_id)
    par = [float(v) for v in par] + [0.0] * (4 - len(par))
    src.par = (ctypes.c_float * 4)(*par)
    src.area_kind, src.emit_kind = int(area_kind), int(emit_kind)
    src.epar = (ctypes.c_float * 4)(*([float(v) for v in emit_par] + [0.0] * (4 - len(emit_par))))
    src.R = (ctypes.c_float * 9)(*[float(v) for v in torch.as_tensor(rot, dtype=torch.float32).reshape(-1).tolist()])
    src.t = (ctypes.c_float * 3)(*[float(v) for v in torch.as_tensor(trans, dtype=torch.float32).reshape(-1).tolist()])
    src.seed, src.offset = int(seed) & (2 ** 64 - 1), int(offset)
    return SampledRays.pending(src, int(n), device)


def _sample_planes(n, device):
    return (torch.empty((3, n), device=device, dtype=torch.float32), torch.empty((3, n), device=device, dtype=torch.float32),
            torch.empty((n,), device=device, dtype=torch.float32), torch.empty((n,), device=device, dtype=torch.int8),
            torch.empty((n,), device=device, dtype=torch.float32))


def generate_sample(pend):
    """The planes of a pending sample: (pos [3,N], dir [3,N], intensity [N], id [N] int8, wavelength [N])."""
    import ctypes
    n, device = pend.n, pend.device
    pos, dir_, inten, ids, wl = _sample_planes(n, device)
    if n:
        with torch.cuda.device(device):
            code = N.lib().rtt_sample_bundle_f32(ctypes.addressof(pend.src), N.ptr(pos), N.ptr(dir_), n, N.ptr(inten),
                                                 N.ptr(ids), N.ptr(wl), n, N.stream_handle(device))
        N.check(code, "rtt_sample_bundle_f32")
    return pos, dir_, inten, ids, wl


def can_generate_in_trace(pend, prog) -> bool:
    """Whether rtt_sample_trace_seq_f32 takes this sample: a source of two uniforms per ray, an even global offset
    (a lane's pair of rays is one Philox block), and something to trace."""
    from .rays.ray import PendingSample
    return (type(pend) is PendingSample and pend.n > 0 and 0 < prog.shape[0] <= P.MAX_SURFACES and 0 <= pend.src.kind <= 6
            and pend.src.offset % 2 == 0)


def sample_trace_sequential(rays, prog: np.ndarray):
    """Bundle.sample + SequentialScene.simulate in one launch for `rays` that are still pending (`SampledRays`): fills
    them with the traced planes and returns the launch's SensorRecord (None for a program without sensor slots)."""
    import ctypes
    pend = rays._pending()
    n, dev = pend.n, pend.device
    pos, dir_, inten, ids, wl = _sample_planes(n, dev)
    dp = N.DeviceProgram.upload(prog, dev)
    k = P.n_sensor_slots(prog)
    if k:
        rec = SensorRecord(k, n, dev)
        hit_p, w_p, mask_p = rec.pointers()
    else:
        rec, hit_p, w_p, mask_p = None, None, None, None
    code = _call_on(dev, N.lib().rtt_sample_trace_seq_f32, ctypes.addressof(pend.src),
                    pos.data_ptr(), dir_.data_ptr(), inten.data_ptr(), n, ids.data_ptr(), wl.data_ptr(), n,
                    dp.dev.data_ptr(), dp.host_ptr, dp.n, hit_p, w_p, mask_p, n, N.stream_handle(dev))
    if code:
        N.check(code, "rtt_sample_trace_seq_f32")   # raises; `rays` is still the pending sample it was
    rays._take()
    rays._pos, rays._dir, rays.intensity, rays.id, rays.wavelength = pos, dir_, inten, ids, wl
    return rec


def spot_groups(x, y, w, mask, ids, pass_id: int, centers, norm_ord: float, out: torch.Tensor):
    """Accumulate per-id sums of recorded hits into `out` ([256, 3] float64 on the device)."""
    N.require_device(x, y, w, ids, out)
    n = x.numel()
    if out.dtype != torch.float64 or tuple(out.shape) != (256, 3) or not out.is_contiguous():
        raise ValueError("`out` must be a contiguous [256, 3] float64 tensor")
    if ids.dtype != torch.int8 or ids.numel() != n or not ids.is_contiguous():
        raise ValueError("ids must be a contiguous int8 tensor with one entry per hit slot")
    for t in (x, y, w):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
            raise ValueError("x, y, w must be contiguous float32 tensors of equal length")
    if mask is not None and (mask.dtype != torch.uint8 or mask.numel() != n or not mask.is_contiguous()):
        raise ValueError("mask must be a contiguous uint8 tensor of the same length")
    if centers is not None and (centers.dtype != torch.float64 or tuple(centers.shape) != (256, 2) or not centers.is_contiguous()):
        raise ValueError("centers must be a contiguous [256, 2] float64 tensor")
    if n:
        with torch.cuda.device(x.device):
            code = N.lib().rtt_spot_groups_f32(N.ptr(x), N.ptr(y), N.ptr(w), N.ptr(mask), N.ptr(ids), n, int(pass_id),
                                               N.ptr(centers), float(norm_ord), N.ptr(out), N.stream_handle(x.device))
        N.check(code, "rtt_spot_groups_f32")
    return out
