"""ctypes binding of ``librtt_hip.so`` (C ABI in ``include/rtt_hip.h``).

There is deliberately no fallback: if the shared library is missing, fails to load, or a tensor
is not on a HIP device, the call raises.  Nothing in this package computes ray geometry on the CPU.
"""
import ctypes
import os

import numpy as np
import torch

from . import program as P

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTT_HIP_LIB selects another build of the same ABI (the strict-fp build, an A/B candidate)
LIB_PATH = os.environ.get("RTT_HIP_LIB") or os.path.join(_HERE, "csrc", "librtt_hip.so")

_lib = None

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_dbl = ctypes.c_double

_SIGNATURES = {
    "rtt_trace_seq_f32": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _int,
                          _vp, _vp, _vp, _i64, _vp],
    "rtt_trace_seq_moments_f32": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _int,
                                  _vp, _vp, _vp, _i64, _int, _vp, _vp, _i64, _vp],
    "rtt_trace_seq_history_f32": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _int,
                                  _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp],
    "rtt_trace_seq_tangent_f32": [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _int, _vp, _int, _vp, _vp, _i64, _vp, _i64,
                                  _vp, _vp],
    "rtt_trace_seq_spot_grad_f32": [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _int, _vp, _int, _int, _dbl, _dbl, _dbl,
                                    _vp, _vp],
    "rtt_trace_seq_adjoint_f32": [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _int, _vp, _vp, _vp, _i64, _vp, _i64,
                                  _int, _int, _dbl, _dbl, _dbl, _vp, _vp, _i64, _vp, _vp, _i64, _vp],
    "rtt_trace_seq_adjoint_static_f32": [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _int, _vp, _vp, _vp, _i64, _vp, _i64,
                                         _int, _int, _dbl, _dbl, _dbl, _vp, _vp, _i64, _vp, _vp],
    "rtt_trace_seq_tape_f32": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _int,
                               _vp, _vp, _vp, _i64, _int, _vp, _vp, _i64, _vp, _i64, _vp],
    "rtt_trace_seq_adjoint_tape_f32": [_vp, _i64, _i64, _vp, _vp, _int, _vp, _vp, _vp, _i64, _vp, _i64,
                                       _int, _int, _dbl, _dbl, _dbl, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp],
    "rtt_ray_cast_f32": [_vp, _vp, _i64, _i64, _vp, _vp, _int, _vp, _vp, _vp],
    "rtt_step_nearest_f32": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _int,
                             _vp, _vp, _vp, _vp, _i64, _vp],
    "rtt_intersect_test_f32": [_vp, _vp, _i64, _i64, _vp, _vp, _int, _vp, _i64, _vp],
    "rtt_surface_forward_f32": [_vp, _vp, _i64, _i64, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp,
                                _i64, _vp],
    "rtt_apply_physics_f32": [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp],
    "rtt_spot_moments_f32": [_vp, _vp, _vp, _vp, _i64, _int, _dbl, _dbl, _dbl, _vp, _vp],
    "rtt_simulate_nearest_f32": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _int, _int,
                                 _vp, _vp, _vp, _vp, _i64, _int, _vp],
    "rtt_simulate_nearest_staged_f32": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _int, _int,
                                        _vp, _vp, _vp, _vp, _i64, _int, _vp, _i64, _vp],
    "rtt_simulate_nearest_tangent_f32": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _int, _int,
                                         _vp, _int, _int, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "rtt_trace_seq_ray_adjoint_f32": [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _int, _vp, _vp, _vp, _i64, _vp, _i64,
                                      _vp, _i64, _vp],
    "rtt_spot_groups_f32": [_vp, _vp, _vp, _vp, _vp, _i64, _int, _vp, _dbl, _vp, _vp],
    "rtt_render_f32": [_vp, _vp, _i64, _i64, _vp, _vp, _int, _vp, ctypes.c_uint64, _vp, _vp, _vp, _vp],
    "rtt_sample_bundle_f32": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp],
    "rtt_sample_trace_seq_f32": [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _int, _vp, _vp, _vp, _i64, _vp],
}
EXPORTS = tuple(_SIGNATURES) + ("rtt_strerror", "rtt_abi_version", "rtt_surface_record_bytes", "rtt_tune",
                                "rtt_trace_seq_adjoint_max_slots", "rtt_trace_seq_adjoint_workspace_bytes",
                                "rtt_trace_seq_moments_workspace_bytes", "rtt_simulate_nearest_staged_workspace_bytes",
                                "rtt_trace_seq_tape_bytes", "rtt_trace_seq_adjoint_tape_max_slots",
                                "rtt_prepare_trace_seq_f32", "rtt_prepare_step_nearest_f32", "rtt_launch_prepared",
                                "rtt_release_prepared", "rtt_sensor_record_bytes")


class RttSource(ctypes.Structure):
    """Mirror of `RttSource` in include/rtt_hip.h (passed by pointer, read on the host)."""
    _fields_ = [("kind", ctypes.c_int32), ("ray_id", ctypes.c_int32), ("par", ctypes.c_float * 4),
                ("R", ctypes.c_float * 9), ("t", ctypes.c_float * 3), ("seed", ctypes.c_uint64),
                ("offset", ctypes.c_uint64), ("area_kind", ctypes.c_int32), ("emit_kind", ctypes.c_int32),
                ("epar", ctypes.c_float * 4)]


class NativeLibraryError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; raise loudly when it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C raytracetorch_amd/csrc`. There is no CPU fallback.")
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, args in _SIGNATURES.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = _int
    L.rtt_strerror.argtypes = [_int]
    L.rtt_strerror.restype = ctypes.c_char_p
    L.rtt_abi_version.restype = _int
    L.rtt_tune.argtypes = [_int, _int]
    L.rtt_tune.restype = _int
    L.rtt_surface_record_bytes.restype = _int
    L.rtt_trace_seq_adjoint_max_slots.argtypes = [_int]
    L.rtt_trace_seq_adjoint_max_slots.restype = _int
    L.rtt_trace_seq_adjoint_workspace_bytes.argtypes = [_int]
    L.rtt_trace_seq_adjoint_workspace_bytes.restype = _i64
    L.rtt_trace_seq_moments_workspace_bytes.argtypes = [_i64]
    L.rtt_trace_seq_moments_workspace_bytes.restype = _i64
    L.rtt_simulate_nearest_staged_workspace_bytes.argtypes = [_i64]
    L.rtt_simulate_nearest_staged_workspace_bytes.restype = _i64
    L.rtt_trace_seq_tape_bytes.argtypes = [_i64, _int]
    L.rtt_trace_seq_tape_bytes.restype = _i64
    L.rtt_trace_seq_adjoint_tape_max_slots.argtypes = [_int]
    L.rtt_trace_seq_adjoint_tape_max_slots.restype = _int
    L.rtt_prepare_trace_seq_f32.argtypes = [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _int, _i64, ctypes.POINTER(_vp)]
    L.rtt_prepare_trace_seq_f32.restype = _int
    L.rtt_prepare_step_nearest_f32.argtypes = L.rtt_prepare_trace_seq_f32.argtypes
    L.rtt_prepare_step_nearest_f32.restype = _int
    L.rtt_launch_prepared.argtypes = [_vp, _vp, _vp]
    L.rtt_launch_prepared.restype = _int
    L.rtt_release_prepared.argtypes = [_vp]
    L.rtt_release_prepared.restype = None
    L.rtt_sensor_record_bytes.argtypes = [_int, _i64]
    L.rtt_sensor_record_bytes.restype = _i64
    if L.rtt_surface_record_bytes() != P.RECORD_BYTES:
        raise NativeLibraryError("RttSurface size mismatch between librtt_hip.so and program.py")
    _lib = L
    return L


def check(code: int, what: str):
    if code != 0:
        raise RuntimeError(f"{what} failed: {lib().rtt_strerror(code).decode()} ({code})")


def require_device(*tensors):
    for t in tensors:
        if t is not None and t.device.type != "cuda":
            raise RuntimeError(
                "raytracetorch_amd runs ray geometry only on a HIP device (torch device 'cuda'); "
                f"got a tensor on '{t.device}'. There is no CPU fallback.")


def ptr(t):
    return 0 if t is None else t.data_ptr()


def stream_handle(device):
    """Raw handle of torch's current stream on `device` (the launch stream of every entry point)."""
    if type(device) is not torch.device:
        device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(index)


class PreparedLaunch:
    """Handle of `rtt_prepare_*`: one launch with its arguments validated and bound (ray planes, device program, sensor
    leading dimension).  `launch(sens_base, stream)` enqueues it; the handle is released with the object."""
    __slots__ = ("handle", "_launch", "_release")

    def __init__(self, prepare, *args):
        L = lib()
        out = _vp()
        check(prepare(*args, ctypes.byref(out)), "rtt_prepare")
        self.handle, self._launch, self._release = out.value, L.rtt_launch_prepared, L.rtt_release_prepared

    def launch(self, sens_base, stream):
        return self._launch(self.handle, sens_base, stream)

    def __del__(self):
        h, self.handle = self.handle, None
        if h:
            self._release(h)


class DeviceProgram:
    """Host records + their device copy.  Uploads are skipped for the programs used most recently (a few per
    device, record count and use), so scenes that take turns do not upload before every launch."""

    _cache = {}
    _KEEP = 8

    def __init__(self, host: np.ndarray, dev: torch.Tensor):
        self.host = host
        self.dev = dev

    @classmethod
    def upload(cls, prog: np.ndarray, device, tag: str = "") -> "DeviceProgram":
        device = torch.device(device)
        index = device.index if device.index is not None else torch.cuda.current_device()
        facts = P.facts_of(prog)   # a program packed by a scene remembers its device copies
        if facts is not None:
            dp = facts.get(("device copy", index, tag))
            if dp is not None:
                return dp
            dp = cls._upload(np.asarray(prog), device, index, tag)
            facts[("device copy", index, tag)] = dp
            # the device copy is found again by the identity of `prog`, not by its bytes: from here on the records
            # are read-only, so that an edit in place raises instead of tracing the stale copy (edit a `.copy()`)
            prog.flags.writeable = False
            return dp
        return cls._upload(prog, device, index, tag)

    @classmethod
    def _upload(cls, prog, device, index, tag):
        prog = np.ascontiguousarray(prog)
        key = (index, prog.shape[0], tag)
        raw = prog.tobytes()
        recent = cls._cache.setdefault(key, {})
        dp = recent.get(raw)
        if dp is not None:
            return dp
        dev = torch.from_numpy(prog.view(np.uint8).reshape(-1).copy()).to(device)
        dp = cls(prog, dev)
        if len(recent) >= cls._KEEP:   # drop the oldest copy (dicts keep insertion order)
            del recent[next(iter(recent))]
        recent[raw] = dp
        return dp

    @property
    def n(self):
        return self.host.shape[0]

    @property
    def host_ptr(self):
        p = self.__dict__.get("_host_ptr")
        if p is None:   # (numpy builds the ctypes view on every access)
            p = self.__dict__["_host_ptr"] = self.host.ctypes.data
        return p
