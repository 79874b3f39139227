"""Scene compiler: flattens ``elements -> surfaces`` into the packed *surface program* the HIP
kernels execute (one 256-byte ``RttSurface`` record per (element, surface); layout declared in
``include/rtt_hip.h``).

The reference walks its ``nn.Module`` tree once per surface visit from Python
(scene/sequential.py:17-34, scene/base.py:164,213-226); here the tree is walked once per call on the
host and everything per-ray happens inside one kernel.  Records are re-packed from the live
``nn.Parameter`` values on every call; host copies of parameter values are cached by
``(data_ptr, _version)`` so unchanged parameters cost no device synchronisation.
"""
import weakref

import numpy as np
import torch

SURFACE_DTYPE = np.dtype([
    ("surf_kind", "<i4"), ("bound_kind", "<i4"), ("shape_kind", "<i4"), ("phys_kind", "<i4"),
    ("flags", "<i4"), ("element_id", "<i4"), ("surface_id", "<i4"), ("sensor_slot", "<i4"),
    ("Rs", "<f4", (9,)), ("ts", "<f4", (3,)), ("Rj", "<f4", (9,)), ("tj", "<f4", (3,)),
    ("geom", "<f4", (4,)), ("bnd", "<f4", (4,)), ("shp", "<f4", (8,)), ("phys", "<f4", (4,)),
    ("phys_R", "<f4", (9,)), ("seed", "<u4", (2,)), ("pad", "<f4", (1,)),
])
RECORD_BYTES = SURFACE_DTYPE.itemsize
assert RECORD_BYTES == 256
MAX_SURFACES = 64

# enum mirrors of include/rtt_hip.h
SURF_PLANE, SURF_QUADRIC, SURF_QUADRIC_ZY, SURF_CYLINDER, SURF_SPHERE, SURF_CONE = range(6)
(BND_NONE, BND_DISK, BND_RECT, BND_ELLIPSE, BND_HALF, BND_HALF_DISK, BND_SINGLE_CONE,
 BND_ALWAYS) = range(8)
SHP_NONE, SHP_SPHERIC_FACE, SHP_SPHERIC_EDGE, SHP_CYL_FACE, SHP_CYL_EDGE, SHP_POLYHEDRON = range(6)
PHYS_TRANSMIT, PHYS_BLOCK, PHYS_REFLECT, PHYS_SNELL, PHYS_APERTURE, PHYS_LINEAR, PHYS_FRESNEL = range(7)
F_HAS_SHAPE, F_SHAPE_ROT_ID, F_SURF_ROT_ID, F_INVERT, F_SENSOR, F_PHYS_ROT_SURF = 1, 2, 4, 8, 16, 32

_IDENTITY9 = np.eye(3, dtype=np.float32).reshape(9)
f32 = np.float32

_host_cache = {}

try:   # host-only C++ helper (raytracetorch_amd/csrc/fastparams.cpp): one call reads what the cache checks below need
    from . import _fastparams
    snapshot = _fastparams.snapshot
except ImportError:   # not built: the same facts, read tensor by tensor
    def snapshot(params):
        """(what identifies the storage of every tensor, whether any requires grad, what identifies the state of the
        tensors outside host memory, the bytes of the host-resident ones)"""
        host = [q for q in params if q.device.type == "cpu"]
        return (tuple((q.data_ptr(), q.numel(), q.dtype) for q in params), any(q.requires_grad for q in params),
                tuple(q._version for q in params if q.device.type != "cpu"),
                b"".join([q.detach().contiguous().reshape(-1).numpy().tobytes() for q in host]))


def host_value(p) -> np.ndarray:
    """float32 host copy of a (possibly device-resident) parameter / tensor / python number."""
    if not isinstance(p, torch.Tensor):
        return np.asarray(p, dtype=np.float32)
    if p.device.type == "cpu":
        return p.detach().to(torch.float32).numpy()
    hit = _cached(p)
    if hit is not None:
        return hit
    v = p.detach().to(device="cpu", dtype=torch.float32).numpy()
    _remember(p, v)
    return v


def _cached(p):
    """Host copy of device tensor `p` if one is held for this very tensor object in its current state.  The entry is
    keyed by id() and therefore also holds a weak reference: a temporary that has been freed can hand its id, its
    storage address and its version counter (0) to an unrelated new tensor."""
    hit = _host_cache.get(id(p))
    if hit is not None and hit[2]() is p and hit[0] == (p.data_ptr(), p._version):
        return hit[1]
    return None


def forget_host_copy(p) -> None:
    _host_cache.pop(id(p), None)


def _remember(p, v):
    if len(_host_cache) > 4096:
        _host_cache.clear()
    _host_cache[id(p)] = ((p.data_ptr(), p._version), v, weakref.ref(p))


def prefetch_host(params) -> None:
    """Bring every device-resident parameter whose host copy is stale to the host in ONE transfer (packing a program
    reads them one by one; a transfer each would synchronise with the device once per parameter)."""
    stale = [p for p in params if isinstance(p, torch.Tensor) and p.device.type != "cpu" and _cached(p) is None]
    if len(stale) < 2:
        return
    flat = torch.cat([p.detach().reshape(-1).to(torch.float32) for p in stale]).cpu().numpy()
    at = 0
    for p in stale:
        n = p.numel()
        _remember(p, flat[at:at + n].reshape(tuple(p.shape)).copy())
        at += n


def scalar(p) -> np.float32:
    return f32(host_value(p).reshape(-1)[0])


def new_record() -> np.ndarray:
    rec = np.zeros((), dtype=SURFACE_DTYPE)
    rec["Rs"] = _IDENTITY9
    rec["Rj"] = _IDENTITY9
    rec["phys_R"] = _IDENTITY9
    rec["sensor_slot"] = -1
    return rec


def set_transform(rec, which: str, transform):
    """Fill (R, t) of the element ('s') or surface ('j') frame from a RayTransform and set the
    identity-rotation flag when R is bit-exactly the identity."""
    if getattr(transform, "noisy", False):
        raise NotImplementedError(
            "RayTransformNoisy with a non-zero deviation: the reference raises IndexError in every transform of this "
            "class (geom/transform.py:184), so there is no behaviour to reproduce")
    R = host_value(transform.rot).reshape(9)
    t = host_value(transform.trans).reshape(3)
    rec["R" + which] = R
    rec["t" + which] = t
    if np.array_equal(R, _IDENTITY9):
        rec["flags"] |= F_SHAPE_ROT_ID if which == "s" else F_SURF_ROT_ID


def sag_f32(c, tz, r) -> np.float32:
    """HalfSphere/HalfCyl.sagittalZ in float32, op for op (geom/bounded.py:129-139, :176-186)."""
    c, tz, r = f32(c), f32(tz), f32(r)
    r_sq = r * r
    term = f32(1.0) - (c * c) * r_sq
    term = term if term > 0 else f32(0.0)
    denom = f32(1.0) + np.sqrt(term, dtype=np.float32)
    return f32((c * r_sq) / denom + tz)


class Program(np.ndarray):
    """The packed records of a scene, as `compile_elements` returns them: a structured array that also carries facts
    derived from it once (`facts`: number of sensor slots, presence of a stochastic functor, device copies), so that a
    repeated launch of an unchanged program does not derive them again.  Slices and copies start without facts.  Once
    a device copy has been derived the records are made read-only (`flags.writeable = False`): a changed program is a
    new `Program` (`prog.copy()`), never an edit in place, which would leave the facts stale."""

    def __array_finalize__(self, obj):
        self.facts = None


def facts_of(prog):
    """The dict of derived facts of a `Program` (created on first use), or None for a plain record array."""
    if type(prog) is not Program:
        return None
    if prog.facts is None:
        prog.facts = {}
    return prog.facts


def compile_elements(elements) -> np.ndarray:
    """Records for every (element, surface) in scene order (scene/base.py:113-123)."""
    recs = []
    n_sensor = 0
    for k, el in enumerate(elements):
        for j in range(len(el.shape)):
            rec = el._record(j)
            rec["element_id"] = k
            rec["surface_id"] = j
            if int(rec["flags"]) & F_SENSOR:
                rec["sensor_slot"] = n_sensor
                n_sensor += 1
            recs.append(rec)
    if not recs:
        return np.zeros((0,), dtype=SURFACE_DTYPE).view(Program)
    return np.stack(recs).view(Program)


def split_program(prog, limit: int = MAX_SURFACES):
    """A program of any length as pieces the kernels accept (at most `limit` records each, cut between elements: the
    records of a polyhedron refer to their siblings): [(piece, index of its first record in `prog`, number of sensor
    slots in front of it)].  The sensor slots of a piece are numbered from 0; a launch is given the planes of its first
    slot.  The pieces of a `Program` are made once and kept with it."""
    if prog.shape[0] <= limit:
        return [(prog, 0, 0)]
    facts = facts_of(prog)
    key = ("pieces", limit)
    if facts is not None and key in facts:
        return facts[key]
    plain = np.asarray(prog)
    ids = plain["element_id"]
    cuts = [0] + [int(i) for i in np.nonzero(ids[1:] != ids[:-1])[0] + 1] + [plain.shape[0]]   # first record of every element
    pieces, start, slots = [], 0, 0
    while start < plain.shape[0]:
        end = max((c for c in cuts if start < c <= start + limit), default=None)
        if end is None:
            raise ValueError(f"an element of the scene has more than {limit} surfaces")
        piece = plain[start:end].copy()
        sensors = (piece["flags"] & F_SENSOR) != 0
        piece["sensor_slot"][sensors] -= slots
        pieces.append((piece.view(Program), start, slots))
        slots += int(sensors.sum())
        start = end
    if facts is not None:
        facts[key] = pieces
    return pieces


def n_sensor_slots(prog) -> int:
    facts = facts_of(prog)
    if facts is None:
        return int(((prog["flags"] & F_SENSOR) != 0).sum())
    k = facts.get("sensor slots")
    if k is None:
        k = facts["sensor slots"] = int(((np.asarray(prog)["flags"] & F_SENSOR) != 0).sum())
    return k


def has_stochastic_functor(prog) -> bool:
    facts = facts_of(prog)
    if facts is None:
        return bool((prog["phys_kind"] == PHYS_FRESNEL).any())
    v = facts.get("stochastic")
    if v is None:
        v = facts["stochastic"] = bool((np.asarray(prog)["phys_kind"] == PHYS_FRESNEL).any())
    return v
