"""CPU oracle for the ray-propagation hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, on plain CPU torch tensors, the per-ray arithmetic of the reference's
sequential and nearest-hit trace so that the HIP kernels can be checked against it.  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
it; the product package never does.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function here against golden
vectors in ``tests/golden/*.npz`` that were produced by running the unmodified reference
(``oracle/make_goldens.py``), and against the known answers of the reference's own
``tests/test_primitive.py``.

The oracle interprets the same packed *surface program* the kernels consume (one 256-byte record
per (element, surface), layout declared in ``include/rtt_hip.h``), but shares no code with the
kernels or with the host packer: the record layout and the enum values are restated below.

All citations are file:line relative to the reference tree.
"""
import math

import numpy as np
import torch

INF = float("inf")
EPS = 1e-6  # geom/primitives.py:6 (intersectEpsilon) and :21 (Surface.epsilon)

SURFACE_DTYPE = np.dtype([
    ("surf_kind", "<i4"), ("bound_kind", "<i4"), ("shape_kind", "<i4"), ("phys_kind", "<i4"),
    ("flags", "<i4"), ("element_id", "<i4"), ("surface_id", "<i4"), ("sensor_slot", "<i4"),
    ("Rs", "<f4", (9,)), ("ts", "<f4", (3,)), ("Rj", "<f4", (9,)), ("tj", "<f4", (3,)),
    ("geom", "<f4", (4,)), ("bnd", "<f4", (4,)), ("shp", "<f4", (8,)), ("phys", "<f4", (4,)),
    ("phys_R", "<f4", (9,)), ("seed", "<u4", (2,)), ("pad", "<f4", (1,)),
])
assert SURFACE_DTYPE.itemsize == 256

PLANE, QUADRIC, QUADRIC_ZY, CYLINDER, SPHERE, CONE = range(6)
B_NONE, B_DISK, B_RECT, B_ELLIPSE, B_HALF, B_HALF_DISK, B_SINGLE_CONE, B_ALWAYS = range(8)
S_NONE, S_SPHERIC_FACE, S_SPHERIC_EDGE, S_CYL_FACE, S_CYL_EDGE, S_POLYHEDRON = range(6)
P_TRANSMIT, P_BLOCK, P_REFLECT, P_SNELL, P_APERTURE, P_LINEAR, P_FRESNEL = range(7)
F_HAS_SHAPE, F_SHAPE_ROT_ID, F_SURF_ROT_ID, F_INVERT, F_SENSOR, F_PHYS_ROT_SURF = 1, 2, 4, 8, 16, 32


class _Rec:
    """One surface record with its float fields as python floats / torch tensors of `dtype`."""

    def __init__(self, rec, dtype):
        self.surf_kind = int(rec["surf_kind"]); self.bound_kind = int(rec["bound_kind"])
        self.shape_kind = int(rec["shape_kind"]); self.phys_kind = int(rec["phys_kind"])
        self.flags = int(rec["flags"]); self.element_id = int(rec["element_id"])
        self.surface_id = int(rec["surface_id"]); self.sensor_slot = int(rec["sensor_slot"])
        t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32)).to(dtype)
        self.Rs = t(rec["Rs"]).reshape(3, 3); self.ts = t(rec["ts"])
        self.Rj = t(rec["Rj"]).reshape(3, 3); self.tj = t(rec["tj"])
        self.geom = t(rec["geom"]); self.bnd = t(rec["bnd"]); self.shp = t(rec["shp"])
        self.phys = t(rec["phys"])
        self.phys_R = t(rec["phys_R"]).reshape(3, 3)
        self.seed = int(rec["seed"][0]) | (int(rec["seed"][1]) << 32)
        self.dtype = dtype


def _rowmat(v, R):
    """v @ R for row vectors, spelled out so the summation order is fixed (geom/transform.py:92-93)."""
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    return torch.stack([x * R[0, j] + y * R[1, j] + z * R[2, j] for j in range(3)], dim=1)


def _rowmat_T(v, R):
    """v @ R.T (geom/transform.py:116-117, geom/primitives.py:94, geom/shape.py:85)."""
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    return torch.stack([x * R[j, 0] + y * R[j, 1] + z * R[j, 2] for j in range(3)], dim=1)


def _normalize(d):
    """Rays.__post_init__ (rays/ray.py:22-25): F.normalize(dir, p=2, dim=1), eps 1e-12."""
    n = torch.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])
    return d / torch.clamp_min(n, 1e-12).unsqueeze(1)


def _nanmin(a, b):
    """torch.min over a stacked dim propagates NaN (geom/primitives.py:34, geom/bounded.py:34)."""
    return torch.min(torch.stack([a, b]), dim=0)[0]


# --------------------------------------------------------------------------------------------
# roots of the ray/surface equation in the surface frame
# --------------------------------------------------------------------------------------------
def solve_roots(S, pl, dl):
    ox, oy, oz = pl[:, 0], pl[:, 1], pl[:, 2]
    dx, dy, dz = dl[:, 0], dl[:, 1], dl[:, 2]
    eps = torch.tensor(EPS, dtype=torch.float32).to(S.dtype)
    inf = torch.full_like(ox, INF)
    k = S.surf_kind
    if k == PLANE:  # geom/primitives.py:124-136
        safe = torch.where(torch.abs(dz) < eps, torch.tensor(1e-8, dtype=S.dtype), dz)
        return [-oz / safe]
    if k == SPHERE:  # geom/primitives.py:155-181
        b = 2.0 * (ox * dx + oy * dy + oz * dz)
        c = (ox * ox + oy * oy + oz * oz) - S.geom[1]
        disc = b * b - 4 * c
        hit = disc >= 0
        sq = torch.sqrt(torch.where(hit, disc, torch.zeros_like(disc)))
        t1 = (-b - sq) / 2.0
        t2 = (-b + sq) / 2.0
        return [torch.where(hit, t1, inf), torch.where(hit, t2, inf)]
    if k == CYLINDER:  # geom/primitives.py:201-231
        A = dx * dx + dy * dy
        B = 2.0 * (ox * dx + oy * dy)
        C = (ox * ox + oy * oy) - S.geom[1]
        disc = B * B - 4.0 * A * C
        hit = disc >= 0
        sq = torch.sqrt(torch.abs(disc))
        t1 = (-B - sq) / (2.0 * A)
        t2 = (-B + sq) / (2.0 * A)
        return [torch.where(hit, t1, inf), torch.where(hit, t2, inf)]
    if k in (QUADRIC, QUADRIC_ZY):  # geom/primitives.py:266-328, :356-377
        c, c1k = S.geom[0], S.geom[2]
        two_c, two_c1k = 2 * c, 2 * c1k
        if k == QUADRIC:
            A = c * (dx * dx + dy * dy) + c1k * (dz * dz)
            B = two_c * (ox * dx + oy * dy) + two_c1k * oz * dz - 2 * dz
            C = c * (ox * ox + oy * oy) + c1k * (oz * oz) - 2 * oz
        else:
            A = c * (dy * dy) + c1k * (dz * dz)
            B = two_c * (oy * dy) + two_c1k * oz * dz - 2 * dz
            C = c * (oy * oy) + c1k * (oz * oz) - 2 * oz
        return _solve_quadratic_linear_fallback(A, B, C, eps, inf, abs_disc=True)
    if k == CONE:  # geom/primitives.py:416-466
        k2 = S.geom[1]
        A = dz * dz - k2 * (dx * dx + dy * dy)
        B = 2.0 * (oz * dz - k2 * (ox * dx + oy * dy))
        C = oz * oz - k2 * (ox * ox + oy * oy)
        return _solve_quadratic_linear_fallback(A, B, C, eps, inf, abs_disc=False)
    raise ValueError(f"unknown surface kind {k}")


def _solve_quadratic_linear_fallback(A, B, C, eps, inf, abs_disc):
    """Quadric._solve_quadratic (geom/primitives.py:290-320) / Cone (:436-466)."""
    disc = B * B - 4 * A * C
    hit = disc >= 0
    lin = torch.abs(A) < eps
    if abs_disc:
        sq = torch.sqrt(torch.abs(disc))
    else:
        sq = torch.sqrt(torch.where(hit, disc, torch.zeros_like(disc)))
    A_safe = torch.where(lin, torch.ones_like(A), A)
    t1 = (-B - sq) / (2.0 * A_safe)
    t2 = (-B + sq) / (2.0 * A_safe)
    B_safe = torch.where(torch.abs(B) < eps, eps.expand_as(B), B)
    t_lin = -C / B_safe
    t1 = torch.where(hit, t1, inf)
    t2 = torch.where(hit, t2, inf)
    return [torch.where(lin, t_lin, t1), torch.where(lin, t_lin, t2)]


# --------------------------------------------------------------------------------------------
# bounds
# --------------------------------------------------------------------------------------------
def surface_in_bounds(S, h):
    """SurfaceBounded.inBounds variants, surface frame, without `invert` (geom/bounded.py)."""
    x, y, z = h[:, 0], h[:, 1], h[:, 2]
    b = S.bound_kind
    one_eps = torch.tensor(1 + EPS, dtype=torch.float32).to(S.dtype)
    if b == B_DISK:  # bounded.py:60-64
        return (x * x + y * y) <= S.bnd[0]
    if b == B_RECT:  # bounded.py:77-82
        return (torch.abs(x) <= S.bnd[0]) & (torch.abs(y) <= S.bnd[1])
    if b == B_ELLIPSE:  # bounded.py:98-106
        cs, sn, rM, rm = S.bnd
        dM = x * cs - y * sn
        dm = x * sn + y * cs
        return ((dM / rM) ** 2 + (dm / rm) ** 2) <= 1.0
    if b == B_HALF:  # bounded.py:123-127, :171-174
        return torch.abs(z * S.geom[0]) < one_eps
    if b == B_HALF_DISK:  # bounded.py:151-159
        return (torch.abs(z * S.geom[0]) < one_eps) & ((x * x + y * y) <= S.bnd[0])
    if b == B_SINGLE_CONE:  # bounded.py:208-217
        return (z * S.geom[0]) >= torch.tensor(-EPS, dtype=torch.float32).to(S.dtype)
    if b == B_ALWAYS:
        return torch.ones_like(x, dtype=torch.bool)
    raise ValueError(f"surface_in_bounds on bound kind {b}")


def _sag(c, tz, y):
    """HalfCyl.sagittalZ with a per-ray height (geom/bounded.py:176-186)."""
    r_sq = y * y
    term = torch.relu(1.0 - (c * c) * r_sq)
    return (c * r_sq) / (1.0 + torch.sqrt(term)) + tz


def shape_in_bounds(S, he):
    """Shape.inBounds(hit, j) in the element frame (geom/spherics.py:27-46, geom/cylindrics.py:23-55)."""
    x, y, z = he[:, 0], he[:, 1], he[:, 2]
    k = S.shape_kind
    if k == S_SPHERIC_FACE:
        return (x * x + y * y) <= S.shp[0]
    if k == S_SPHERIC_EDGE:
        return (z >= S.shp[0]) & (z <= S.shp[1])
    if k in (S_CYL_FACE, S_CYL_EDGE):
        # shp[0..3] = x_max+1e-5, x_min-1e-5, y_max+1e-5, y_min-1e-5
        ap = (x <= S.shp[0]) & (x >= S.shp[1]) & (y <= S.shp[2]) & (y >= S.shp[3])
        if k == S_CYL_FACE:
            return ap
        zf = _sag(S.shp[4], S.shp[5], y)
        zb = _sag(S.shp[6], S.shp[7], y)
        in_z = (z >= zf + 1e-4) & (z <= zb - 1e-4)
        return in_z & ap
    if k == S_POLYHEDRON:  # geom/shape.py:122-132; S.siblings = the element's plane records in surface order
        inside = torch.ones_like(x, dtype=torch.bool)
        for q, Q in enumerate(S.siblings):
            if q == S.surface_id:
                continue
            nq, tq = Q.Rj[2, :], Q.tj
            lz = nq[0] * (x - tq[0]) + nq[1] * (y - tq[1]) + nq[2] * (z - tq[2])
            inside = inside & (lz < 1e-4)
        return inside
    raise ValueError(f"shape_in_bounds on shape kind {k}")


# --------------------------------------------------------------------------------------------
# one surface: frames, t, test column, forward, physics
# --------------------------------------------------------------------------------------------
def to_frames(S, p, d):
    """Global -> element frame (+ re-normalisation) -> surface frame.
    geom/shape.py:37-38, rays/ray.py:84-97,:25, geom/transform.py:90-93, geom/primitives.py:50."""
    if S.flags & F_HAS_SHAPE:
        pe = _rowmat(p - S.ts[None, :], S.Rs)
        de = _rowmat(d, S.Rs)
        den = _normalize(de)
    else:
        pe, de, den = p, d, d
    pl = _rowmat(pe - S.tj[None, :], S.Rj)
    dl = _rowmat(den, S.Rj)
    return pe, de, pl, dl


def solve_t(S, pl, dl):
    """Surface._check_t (geom/primitives.py:28-36) / SurfaceBounded._check_t (geom/bounded.py:20-36)."""
    roots = solve_roots(S, pl, dl)
    eps = torch.tensor(EPS, dtype=torch.float32).to(S.dtype)
    out = []
    for r in roots:
        bad = r <= eps
        if S.bound_kind != B_NONE:
            h = pl + r[:, None] * dl
            keep = surface_in_bounds(S, h)
            if S.flags & F_INVERT:
                keep = ~keep
            bad = bad | ~keep
        out.append(torch.where(bad, torch.full_like(r, INF), r))
    t = out[0]
    for r in out[1:]:
        t = _nanmin(t, r)
    return t


def test_column(S, p, d):
    """Column of Element.intersectTest for this surface (geom/shape.py:42-56 or primitives.py:38-57)."""
    pe, de, pl, dl = to_frames(S, p, d)
    t = solve_t(S, pl, dl)
    if S.flags & F_HAS_SHAPE:
        he = pe + t[:, None] * de  # un-normalised element-frame direction, shape.py:47
        valid = (t < INF) & shape_in_bounds(S, he)
        t = torch.where(valid, t, torch.full_like(t, INF))
    return t


def surface_normal(S, hl):
    """_getNormal in the surface frame."""
    x, y, z = hl[:, 0], hl[:, 1], hl[:, 2]
    k = S.surf_kind
    zero = torch.zeros_like(x)
    if k == PLANE:  # primitives.py:138-143
        return torch.stack([zero, zero, torch.ones_like(x)], dim=1)
    if k == SPHERE:  # primitives.py:183-187
        return hl / S.geom[0]
    if k == CYLINDER:  # primitives.py:233-241
        return torch.stack([x / S.geom[0], y / S.geom[0], zero], dim=1)
    if k in (QUADRIC, QUADRIC_ZY):  # primitives.py:330-343, :379-395
        two_c, two_c1k = 2 * S.geom[0], 2 * S.geom[2]
        nx = two_c * x if k == QUADRIC else zero
        ny = two_c * y
        nz = two_c1k * z - 2.0
        raw = torch.stack([nx, ny, nz], dim=1)
        ln = torch.sqrt(nx * nx + ny * ny + nz * nz)
        return -(raw / (ln[:, None] + 1e-8))
    if k == CONE:  # primitives.py:468-497
        k2 = S.geom[1]
        raw = torch.stack([-k2 * x, -k2 * y, z], dim=1)
        ln = torch.sqrt(raw[:, 0] ** 2 + raw[:, 1] ** 2 + raw[:, 2] ** 2)
        unit = raw / (ln[:, None] + 1e-8)
        up = torch.tensor([0.0, 0.0, 1.0], dtype=S.dtype).expand_as(raw)
        return torch.where((ln > 1e-8)[:, None], unit, up)
    raise ValueError(k)


def surface_forward(S, p, d):
    """Shape.forward / Surface.forward for every ray (geom/shape.py:61-87, primitives.py:59-96).
    Returns t, hit_global, normal_global, hit_local."""
    pe, de, pl, dl = to_frames(S, p, d)
    t = solve_t(S, pl, dl)
    hit_g = p + t[:, None] * d  # shape.py:81 / primitives.py:85
    hl = pl + t[:, None] * dl   # primitives.py:86
    n = _rowmat_T(surface_normal(S, hl), S.Rj)  # primitives.py:94
    if S.flags & F_HAS_SHAPE:
        n = _rowmat_T(n, S.Rs)  # shape.py:85
    return t, hit_g, n, hl


def fresnel_uniforms(S, n, u=None, bounce=0):
    """The uniforms of RefractFresnel's choice (phys/std.py:200): given ones (pinning against the
    reference's torch.rand values), or the Philox words the kernels use for rays 0..n-1 of a call in bounce
    `bounce` of a launch: word 2*(i&1) of the block with counter (i >> 1, 0, bounce) under key
    seed + (element_id << 8) + surface_id."""
    if u is not None:
        return _as(u, S.dtype).reshape(-1, 1)
    key = (S.seed + (S.element_id << 8) + S.surface_id) & (2 ** 64 - 1)
    ua, _ = philox_uniforms(key, 0, n, word3=bounce)
    return torch.from_numpy(ua).to(S.dtype).reshape(-1, 1)


def apply_physics(S, hl, d, n, u=None):
    """SurfaceFunction.forward(hit_local, dir_global, normal_global) -> (new_dir, imult).
    `u` optionally supplies the uniforms of the stochastic functor."""
    k = S.phys_kind
    ones = torch.ones_like(d[:, 0])
    if k == P_TRANSMIT:  # std.py:227-235
        return d, ones
    if k == P_BLOCK:  # std.py:243-254
        return torch.zeros_like(d), torch.zeros_like(ones)
    if k == P_REFLECT:  # std.py:97-108
        cos = (d[:, 0] * n[:, 0] + d[:, 1] * n[:, 1] + d[:, 2] * n[:, 2])[:, None]
        return d - 2 * cos * n, ones
    if k == P_SNELL:  # std.py:123-145
        dot = (d[:, 0] * n[:, 0] + d[:, 1] * n[:, 1] + d[:, 2] * n[:, 2])[:, None]
        ent = dot < 0
        ne = torch.where(ent, n, -n)
        c1 = torch.abs(dot)
        mu = torch.where(ent, S.phys[2], S.phys[3])
        term = 1.0 - mu * mu * (1.0 - c1 * c1)
        tir = term < 0
        c2 = torch.sqrt(torch.relu(term))
        refr = mu * d + (mu * c1 - c2) * ne
        refl = d - 2 * dot * n
        return torch.where(tir, refl, refr), ones
    if k == P_APERTURE:  # filter.py:24-33
        m = surface_in_bounds(S, hl).to(d.dtype)
        return d * m[:, None], m
    if k == P_FRESNEL:  # std.py:177-219
        dot = (d[:, 0] * n[:, 0] + d[:, 1] * n[:, 1] + d[:, 2] * n[:, 2])[:, None]
        ent = dot < 0
        ne = torch.where(ent, n, -n)
        ci = torch.abs(dot)
        n1 = torch.where(ent, S.phys[0], S.phys[1])
        n2 = torch.where(ent, S.phys[1], S.phys[0])
        mu = n2 / n1
        sin2 = (mu ** 2) * (1.0 - ci ** 2)
        tir = sin2 > 1.0
        ct = torch.sqrt(torch.relu(1.0 - sin2))
        rs = ((n1 * ci - n2 * ct) / (n1 * ci + n2 * ct + 1e-8)) ** 2
        rp = ((n1 * ct - n2 * ci) / (n1 * ct + n2 * ci + 1e-8)) ** 2
        R = torch.where(tir, torch.ones_like(rs), 0.5 * (rs + rp))
        reflect = fresnel_uniforms(S, d.shape[0], u) < R
        refl = d - 2 * dot * n
        refr = mu * d + (mu * ci - ct) * ne
        return torch.where(reflect, refl, refr), ones
    if k == P_LINEAR:  # std.py:72-88
        Rp = S.Rj if (S.flags & F_PHYS_ROT_SURF) else S.phys_R
        dl = _rowmat(d, Rp)
        dl = dl / dl[:, 2][:, None]
        v = torch.stack([S.phys[0] * hl[:, 0] + S.phys[2] * dl[:, 0], S.phys[1] * hl[:, 1] + S.phys[3] * dl[:, 1],
                         torch.ones_like(hl[:, 0])], dim=1)
        return _rowmat_T(_normalize(v), Rp), ones
    raise ValueError(k)


def element_forward(S, p, d):
    """Element.forward (elements/parent.py:44-58) -> new_pos, new_dir, imult, hit_local."""
    t, hit_g, n, hl = surface_forward(S, p, d)
    nd, m = apply_physics(S, hl, d, n)
    return hit_g, nd, m, hl


# --------------------------------------------------------------------------------------------
# drivers
# --------------------------------------------------------------------------------------------
def _records(prog, dtype):
    prog = np.asarray(prog)
    assert prog.dtype == SURFACE_DTYPE
    recs = [_Rec(prog[i], dtype) for i in range(prog.shape[0])]
    for i, S in enumerate(recs):
        if S.shape_kind == S_POLYHEDRON:
            first = i - S.surface_id
            S.siblings = recs[first:first + int(S.shp[0])]
    return recs


def _as(x, dtype):
    return torch.as_tensor(np.asarray(x)).to(dtype).clone()


def trace_sequential(prog, pos, dir_, inten, dtype=torch.float32):
    """SequentialScene.simulate (scene/sequential.py:12-36) on [N,3] pos/dir and [N] intensity.

    Returns dict(pos, dir, intensity, sensor=[(slot, mask, hit_local, intensity_before), ...]).
    Predicated (`where`) instead of gather/scatter: identical arithmetic per ray."""
    p, d, I = _as(pos, dtype), _as(dir_, dtype), _as(inten, dtype)
    sensor = []
    for S in _records(prog, dtype):
        t = test_column(S, p, d)
        mask = t < INF
        hit_g, nd, m, hl = element_forward(S, p, d)
        if S.flags & F_SENSOR:  # elements/sensor.py:35-37: intensity before the multiplier
            sensor.append((S.sensor_slot, mask.clone(), hl.clone(), I.clone()))
        p = torch.where(mask[:, None], hit_g, p)
        d = torch.where(mask[:, None], nd, d)
        I = torch.where(mask, I * m, I)  # rays/ray.py:38-40
    return dict(pos=p, dir=d, intensity=I, sensor=sensor)


def intersect_matrix(prog, pos, dir_, dtype=torch.float32):
    """cat of Element.intersectTest over all elements: [N, S] (scene/base.py:164-168)."""
    p, d = _as(pos, dtype), _as(dir_, dtype)
    return torch.stack([test_column(S, p, d) for S in _records(prog, dtype)], dim=1)


def ray_cast(prog, pos, dir_, dtype=torch.float32):
    """Scene.ray_cast (scene/base.py:144-178): (tmin[N], winner[N] int64 global surface index,
    -1 where no surface is hit)."""
    tm = intersect_matrix(prog, pos, dir_, dtype)
    tmin, idx = torch.min(tm, dim=1)
    win = torch.where(tmin < INF, idx, torch.full_like(idx, -1))
    return tmin, win, tm


def step_nearest(prog, pos, dir_, inten, dtype=torch.float32):
    """Scene.step (scene/base.py:180-235)."""
    p, d, I = _as(pos, dtype), _as(dir_, dtype), _as(inten, dtype)
    tmin, win, _ = ray_cast(prog, p, d, dtype)
    active = (win >= 0) & (I > 0)
    np_, nd_, nI = p.clone(), d.clone(), I.clone()
    sensor = []
    for s, S in enumerate(_records(prog, dtype)):
        m = active & (win == s)
        if not bool(m.any()):
            if S.flags & F_SENSOR:
                sensor.append((S.sensor_slot, m, torch.zeros_like(p), I.clone()))
            continue
        hit_g, nd, im, hl = element_forward(S, p, d)
        if S.flags & F_SENSOR:
            sensor.append((S.sensor_slot, m.clone(), hl.clone(), I.clone()))
        np_ = torch.where(m[:, None], hit_g, np_)
        nd_ = torch.where(m[:, None], nd, nd_)
        nI = torch.where(m, I * im, nI)
    return dict(pos=np_, dir=nd_, intensity=nI, winner=win, tmin=tmin, sensor=sensor)


def spot_size_loss(x, y, w, target_xy=None):
    """SpotSizeLoss.forward inner formula (optim/goals.py:165-183) for one bundle."""
    x, y, w = (torch.as_tensor(a) for a in (x, y, w))
    act = w > 0
    if not bool(act.any()):
        return None
    x, y, w = x[act], y[act], w[act]
    ws = w.sum().clamp(min=1e-12)
    if target_xy is None:
        cx, cy = (x * w).sum() / ws, (y * w).sum() / ws
    else:
        cx, cy = target_xy
    return torch.sqrt(((x - cx) ** 2 + (y - cy) ** 2) * (w / ws)).sum()


# ------------------------------------------------------------------------------------------------
# Ray sources (rays/bundle.py).  The reference draws uniforms with torch.distributions.Uniform; its
# formulas are restated here as functions of the unit uniforms so that they can be pinned against the
# reference (fed with the very torch.rand values it consumed) and then drive the check of the device
# sampler (fed with the Philox values the kernel uses).
# ------------------------------------------------------------------------------------------------
SRC_PENCIL, SRC_DISK, SRC_LINE, SRC_FAN, SRC_POINT, SRC_GAUSS, SRC_RAYLEIGH, SRC_PANEL, SRC_SPHERE = range(9)
AREA_RECT, AREA_RING = 0, 1
EMIT_LAMBERT, EMIT_CONE = 0, 1


def philox4x32_10(counter, key, word2=0, word3=0):
    """Philox4x32-10 (Salmon et al., SC'11; Random123): counter [M] uint64 placed in words 0-1, word 2 =
    `word2`, word 3 = `word3`; key uint64.  Returns [M, 4] uint32."""
    counter = np.asarray(counter, dtype=np.uint64)
    m32 = np.uint64(0xFFFFFFFF)
    c = [counter & m32, counter >> np.uint64(32), np.full_like(counter, word2), np.full_like(counter, word3)]
    k0, k1 = np.uint64(int(key) & 0xFFFFFFFF), np.uint64((int(key) >> 32) & 0xFFFFFFFF)
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & m32, p1 & m32, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & m32, p0 & m32]
        k0 = (k0 + np.uint64(0x9E3779B9)) & m32
        k1 = (k1 + np.uint64(0xBB67AE85)) & m32
    return np.stack(c, axis=1).astype(np.uint32)


def philox_uniforms(seed, offset, n, word3=0):
    """The two unit uniforms of rays offset .. offset+n-1: the pair of rays (2j, 2j+1) shares block j,
    ray 2j takes words (0, 1), ray 2j+1 words (2, 3); u = (x >> 8) * 2^-24."""
    g = np.arange(offset, offset + n, dtype=np.uint64)
    blocks = philox4x32_10(g >> np.uint64(1), seed, word3=word3)
    half = (g & np.uint64(1)).astype(np.int64)
    rows = np.arange(n)
    to_u = lambda x: ((x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24))
    return to_u(blocks[rows, 2 * half]), to_u(blocks[rows, 2 * half + 1])


def _local_only(kind, par, ua, ub, dtype):
    return sample_bundle(kind, par, np.eye(3), np.zeros(3), ua, ub, dtype=dtype, local=True)


def philox_uniforms4(seed, offset, n):
    """The four unit uniforms of rays offset .. offset+n-1 of a source that draws four per ray: block
    (index lo, index hi, 1, 0), words in order."""
    g = np.arange(offset, offset + n, dtype=np.uint64)
    blocks = philox4x32_10(g, seed, word2=1)
    to_u = lambda x: ((x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24))
    return tuple(to_u(blocks[:, k]) for k in range(4))


def _lambert_local(u1, u2):
    """rays/panels.py:27-37, rays/particle.py:62-68."""
    r = torch.sqrt(u1)
    phi = 2.0 * math.pi * u2
    return torch.stack([r * torch.cos(phi), r * torch.sin(phi), torch.sqrt((1.0 - u1).clamp(min=0.0))], dim=1)


def sample_bundle(kind, par, R, t, ua, ub, dtype=torch.float32, uc=None, ud=None, area=AREA_RECT, emit=EMIT_LAMBERT,
                  epar=(), local=False):
    """Bundle.sample (rays/bundle.py:30-37) as a function of the unit uniforms ua (first draw) and ub
    (second draw): local sample (:40-171), RayTransformBundle.transform_ (geom/transform.py:259-262),
    F.normalize of Rays.__post_init__ (rays/ray.py:22-25).  Sources with a position and a direction sampler
    (rays/panels.py, rays/particle.py:9-87) take two more, uc and ud, for the direction.  Returns (pos [N,3],
    dir [N,3]); `local` stops before the bundle pose."""
    ua, ub = _as(ua, dtype).reshape(-1), _as(ub, dtype).reshape(-1)
    par = [torch.tensor(float(v), dtype=dtype) for v in list(par) + [0.0] * (4 - len(par))]
    R = _as(R, dtype).reshape(3, 3)
    t = _as(t, dtype).reshape(3)
    n = ua.shape[0]
    zero, one = torch.zeros(n, dtype=dtype), torch.ones(n, dtype=dtype)
    between = lambda lo, hi, u: lo + u * (hi - lo)   # torch.distributions.Uniform.rsample
    p = torch.stack([zero, zero, zero], dim=1)
    d = torch.stack([zero, zero, one], dim=1)
    if kind == SRC_DISK:        # rays/bundle.py:47-56 (theta first, then r)
        theta = between(par[2], par[3], ua)
        r = torch.sqrt(between(par[0], par[1], ub))
        p = torch.stack([r * torch.cos(theta), r * torch.sin(theta), zero], dim=1)
    elif kind == SRC_LINE:      # rays/bundle.py:111-118
        p = torch.stack([between(par[0], par[1], ua), zero, zero], dim=1)
    elif kind == SRC_FAN:       # rays/bundle.py:136-140
        theta = between(par[0], par[1], ua)
        d = torch.stack([zero, torch.sin(theta), torch.cos(theta)], dim=1)
    elif kind == SRC_POINT:     # rays/bundle.py:65-75, :162-171 (phi first, then theta)
        phi = torch.acos(-2 * between(par[0], par[1], ua) + 1)
        theta = between(par[2], par[3], ub)
        dr = torch.sin(phi)
        d = torch.stack([torch.cos(theta) * dr, torch.sin(theta) * dr, torch.cos(phi)], dim=1)
    elif kind == SRC_GAUSS:     # rays/beam.py:37-41; standard normals by Box-Muller of (ua, ub)
        r = torch.sqrt(-2.0 * torch.log(1.0 - ua))
        p = torch.stack([par[0] * (r * torch.cos(2.0 * math.pi * ub)), par[1] * (r * torch.sin(2.0 * math.pi * ub)), zero], dim=1)
    elif kind == SRC_RAYLEIGH:  # rays/particle.py:111-125
        half_q = 2.0 - 4.0 * ua
        A = (torch.sqrt(half_q ** 2 + 1.0) - half_q).pow(1.0 / 3.0)
        ct = A - 1.0 / A
        st = torch.sqrt((1.0 - ct ** 2).clamp(min=0.0))
        phi = 2.0 * math.pi * ub
        d = torch.stack([st * torch.cos(phi), st * torch.sin(phi), ct], dim=1)
    elif kind == SRC_PANEL:     # rays/panels.py:65-154: position draws first, then the emission model's
        uc, ud = _as(uc, dtype).reshape(-1), _as(ud, dtype).reshape(-1)
        if area == AREA_RING:
            p, _ = _local_only(SRC_DISK, par, ua, ub, dtype)
        else:
            p = torch.stack([between(par[0], par[1], ua), between(par[2], par[3], ub), zero], dim=1)
        if emit == EMIT_CONE:
            _, d = _local_only(SRC_POINT, epar, uc, ud, dtype)
        else:
            d = _lambert_local(uc, ud)
    elif kind == SRC_SPHERE:    # rays/particle.py:34-87
        uc, ud = _as(uc, dtype).reshape(-1), _as(ud, dtype).reshape(-1)
        ct = 2.0 * ua - 1.0
        st = torch.sqrt((1.0 - ct ** 2).clamp(min=0.0))
        phi = 2.0 * math.pi * ub
        nrm = torch.stack([st * torch.cos(phi), st * torch.sin(phi), ct], dim=1)
        p = nrm * par[0]
        l = _lambert_local(uc, ud)
        sign = torch.sign(nrm[:, 2] + 1e-10)
        a = -1.0 / (sign + nrm[:, 2])
        b = nrm[:, 0] * nrm[:, 1] * a
        tg = torch.stack([1.0 + sign * nrm[:, 0] ** 2 * a, sign * b, -sign * nrm[:, 0]], dim=1)
        bt = torch.stack([b, sign + nrm[:, 1] ** 2 * a, -nrm[:, 1]], dim=1)
        d = l[:, 0:1] * tg + l[:, 1:2] * bt + l[:, 2:3] * nrm
    if local:
        return p, d
    pg = _rowmat_T(p, R) + t[None, :]
    dg = _rowmat_T(d, R)
    return pg, _normalize(dg)
