"""CPU: the C-ABI library loads and exports what include/rtt_hip.h declares, the host-side mirror of
the reference API behaves like the reference, and nothing falls back to a CPU computation."""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

import scenes
from raytracetorch_amd import _native, program as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rtt_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rtt_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _native.lib()
    names = declared_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in rtt_hip.h but not exported"
    assert set(names) == set(_native.EXPORTS)
    assert lib.rtt_abi_version() == 6
    assert lib.rtt_surface_record_bytes() == 256 == P.RECORD_BYTES


def test_record_layout_matches_header():
    """Field offsets of the numpy record equal offsetof() of the C struct."""
    fields = [n for n in P.SURFACE_DTYPE.names if n != "pad"]
    prog = "#include <stdio.h>\n#include <stddef.h>\n#include \"rtt_hip.h\"\nint main(){\n"
    for f in fields:
        prog += f'  printf("{f} %zu\\n", offsetof(RttSurface, {f}));\n'
    prog += '  printf("sizeof %zu\\n", sizeof(RttSurface));\n  return 0;\n}\n'
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "o.c"), os.path.join(d, "o")
        open(src, "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        out = dict(l.split() for l in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines())
    for f in fields:
        assert int(out[f]) == P.SURFACE_DTYPE.fields[f][1], f
    assert int(out["sizeof"]) == 256


def test_entry_points_validate_before_touching_the_gpu():
    """Argument errors are reported as codes; no kernel is launched (this runs without a GPU)."""
    lib = _native.lib()
    rec = P.new_record()[None].copy()
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)
    hp = rec.ctypes.data
    E_NULL, E_SIZE, E_KIND, E_ALIGN = -1, -2, -3, -5
    assert lib.rtt_trace_seq_f32(0, p, p, 8, p, p, p, 8, 8, p, hp, 1, 0, 0, 0, 0, 0) == E_NULL
    assert lib.rtt_trace_seq_f32(p, p, p, 4, p, p, p, 8, 8, p, hp, 1, 0, 0, 0, 0, 0) == E_SIZE      # ld < n
    assert lib.rtt_trace_seq_f32(p, p, p, 8, p, p, p, 8, 8, p, hp, 0, 0, 0, 0, 0, 0) == E_SIZE      # n_surf < 1
    assert lib.rtt_trace_seq_f32(p, p, p, 8, p, p, p, 8, 8, p, hp, 65, 0, 0, 0, 0, 0) == E_SIZE     # n_surf > 64
    assert lib.rtt_trace_seq_f32(p + 2, p, p, 8, p, p, p, 8, 8, p, hp, 1, 0, 0, 0, 0, 0) == E_ALIGN
    bad = rec.copy(); bad["surf_kind"] = 17
    assert lib.rtt_trace_seq_f32(p, p, p, 8, p, p, p, 8, 8, p, bad.ctypes.data, 1, 0, 0, 0, 0, 0) == E_KIND
    sens = rec.copy(); sens["flags"] |= P.F_SENSOR; sens["sensor_slot"] = 0
    assert lib.rtt_trace_seq_f32(p, p, p, 8, p, p, p, 8, 8, p, sens.ctypes.data, 1, 0, 0, 0, 8, 0) == E_NULL  # no sensor buffers
    assert lib.rtt_trace_seq_f32(p, p, p, 8, p, p, p, 8, 0, p, hp, 1, 0, 0, 0, 0, 0) == 0            # n == 0: nothing to do
    assert lib.rtt_ray_cast_f32(p, p, 8, 0, p, hp, 1, 0, 0, 0) == 0
    assert lib.rtt_intersect_test_f32(p, p, 8, 8, p, hp, 1, 0, 8, 0) == E_NULL
    assert lib.rtt_surface_forward_f32(p, p, 8, 8, p, hp, 64, 0, 0, 0, 0, 0, 0, 8, 0) == E_SIZE
    assert lib.rtt_spot_moments_f32(p, p, p, 0, 8, 5, 0.0, 0.0, 1.0, p, 0) == E_SIZE
    # reverse-mode backward: host program required, slot budget, workspace rows
    mask = (ctypes.c_uint32 * 1)(0x07000000)   # geom 0-2 of the single surface
    mp = ctypes.addressof(mask)
    adj = lambda **kw: lib.rtt_trace_seq_adjoint_f32(*[kw.get(k, v) for k, v in (
        ("pos", p), ("dir", p), ("inten", p), ("ld", 8), ("n", 8), ("dev", p), ("host", hp), ("n_surf", 1), ("mask", mp),
        ("adj_pos", p), ("adj_dir", 0), ("ld_adj", 8), ("adj_sens", 0), ("sens_ld", 0), ("spot", -1), ("centroid", 0), ("cx", 0.0), ("cy", 0.0),
        ("wsum", 1.0), ("spot_g", 0), ("ray_adj", 0), ("ld_ray", 0), ("out", p), ("ws", p), ("ws_bytes", 1 << 20), ("stream", 0))])
    assert adj(host=0) == E_NULL
    assert adj(ld_adj=4) == E_SIZE
    assert adj(out=0) == E_NULL                       # slots asked for, nowhere to put them
    assert adj(ws_bytes=16) == E_SIZE                 # fewer workspace rows than blocks
    assert adj(spot=0, wsum=0.0) == E_SIZE
    full = (ctypes.c_uint32 * 64)(*([0xFFFFFFFF] * 64))
    # all 32 slots of the one surface: accepted as a request, refused for its workspace
    assert lib.rtt_trace_seq_adjoint_f32(p, p, p, 8, 8, p, hp, 1, ctypes.addressof(full), p, 0, 8, 0, 0, -1, 0, 0.0, 0.0, 1.0, 0, 0, 0,
                                         p, p, 16, 0) == E_SIZE
    assert adj(n=0, mask=0) == 0
    assert lib.rtt_trace_seq_adjoint_max_slots(4) == 128 and lib.rtt_trace_seq_adjoint_max_slots(20) == 20
    assert lib.rtt_trace_seq_adjoint_max_slots(23) == 0 and lib.rtt_trace_seq_adjoint_max_slots(0) == 0
    assert lib.rtt_trace_seq_adjoint_workspace_bytes(10) == 8 * 2048 * 10
    # differentiable bounce loop: host copy of the program required, n_theta in {2,4,6,8}, ray seeds need >= 6
    args = lambda hp_, k, seed: (p, p, p, 8, p, p, p, 8, 8, p, hp_, 1, 3, p, k, seed, p, p, p, p, p, 8, 0)
    assert lib.rtt_simulate_nearest_tangent_f32(*args(0, 8, 0)) == E_NULL
    assert lib.rtt_simulate_nearest_tangent_f32(*args(hp, 5, 0)) == E_SIZE
    assert lib.rtt_simulate_nearest_tangent_f32(*args(hp, 4, 1)) == E_SIZE
    assert lib.rtt_simulate_nearest_tangent_f32(p, p, p, 8, p, p, p, 8, 0, p, hp, 1, 3, p, 8, 1, p, p, p, p, p, 8, 0) == 0
    # per-ray adjoint of the sequential trace
    assert lib.rtt_trace_seq_ray_adjoint_f32(p, p, p, 8, 8, p, hp, 1, 0, p, p, 8, 0, 0, p, 8, 0) == E_NULL     # no zero table
    assert lib.rtt_trace_seq_ray_adjoint_f32(p, p, p, 8, 8, p, hp, 1, p, p, p, 4, 0, 0, p, 8, 0) == E_SIZE     # ld_adj < n
    assert lib.rtt_trace_seq_ray_adjoint_f32(p, p, p, 8, 0, p, hp, 1, p, p, p, 8, 0, 0, p, 8, 0) == 0
    # sources: kinds, and the sub-kinds of a panel
    src = _native.RttSource()
    sp = ctypes.addressof(src)
    src.kind = 9
    assert lib.rtt_sample_bundle_f32(sp, p, p, 8, 0, 0, 0, 8, 0) == E_KIND
    src.kind, src.area_kind = 7, 2
    assert lib.rtt_sample_bundle_f32(sp, p, p, 8, 0, 0, 0, 8, 0) == E_KIND
    src.area_kind, src.emit_kind = 1, 1
    assert lib.rtt_sample_bundle_f32(sp, p, p, 8, 0, 0, 0, 0, 0) == 0
    # prepared launches: the arguments are checked when the handle is made; a handle made before an rtt_tune call is stale
    h = ctypes.c_void_p()
    prep = lambda *a: lib.rtt_prepare_trace_seq_f32(*a, ctypes.byref(h))
    assert lib.rtt_prepare_trace_seq_f32(p, p, p, 8, p, p, p, 8, 8, p, hp, 1, 8, None) == E_NULL    # nowhere to put the handle
    assert prep(0, p, p, 8, p, p, p, 8, 8, p, hp, 1, 8) == E_NULL and not h.value
    assert prep(p, p, p, 8, p, p, p, 8, 8, p, 0, 1, 8) == E_NULL                                     # needs the host copy
    assert prep(p, p, p, 8, p, p, p, 8, 0, p, hp, 1, 8) == E_SIZE                                    # nothing to launch
    assert prep(p, p, p, 4, p, p, p, 8, 8, p, hp, 1, 8) == E_SIZE
    assert prep(p + 2, p, p, 8, p, p, p, 8, 8, p, hp, 1, 8) == E_ALIGN
    assert prep(p, p, p, 8, p, p, p, 8, 8, p, bad.ctypes.data, 1, 8) == E_KIND
    assert prep(p, p, p, 8, p, p, p, 8, 8, p, sens.ctypes.data, 1, 4) == E_SIZE                      # sensor planes shorter than the bundle
    assert lib.rtt_prepare_step_nearest_f32(p, p, p, 8, p, p, p, 8, 8, p, sens.ctypes.data, 1, 8, ctypes.byref(h)) == 0 and h.value
    assert lib.rtt_launch_prepared(None, p, 0) == E_NULL
    assert lib.rtt_launch_prepared(h, None, 0) == E_NULL                                             # a sensor surface, no record
    assert lib.rtt_tune(3, 1) == 0                                                                   # (the default value: only the epoch moves)
    assert lib.rtt_launch_prepared(h, p, 0) == E_KIND                                                # stale: prepare again
    lib.rtt_release_prepared(h)
    lib.rtt_release_prepared(None)
    assert lib.rtt_sensor_record_bytes(2, 10) == 340 and lib.rtt_sensor_record_bytes(1, 3) == 52 and lib.rtt_sensor_record_bytes(0, 8) == 0
    assert lib.rtt_tune(10, 2) == E_SIZE and lib.rtt_tune(10, 0) == 0 and lib.rtt_tune(11, 1) == 0 and lib.rtt_tune(12, 1) == 0 and lib.rtt_tune(13, 0) == E_SIZE
    # the staged bounce loop: distinct output planes, an aligned workspace of the stated size, no Fresnel surface
    assert lib.rtt_simulate_nearest_staged_workspace_bytes(7) == 256 + 768 + (8 * 4 + 1) * 8   # counters, two sets of compact planes (to 256 B), records and marks of 8 rays
    big = (ctypes.c_float * 1024)()
    ws = (ctypes.addressof(big) + 255) & ~255
    q = p + 128
    st = lambda *a: lib.rtt_simulate_nearest_staged_f32(*a)
    assert st(p, p, p, 8, q, q, q, 8, 0, p, hp, 1, 3, 0, 0, 0, 0, 0, 1, ws, 4096, 0) == 0
    assert st(p, p, p, 8, p, q, q, 8, 0, p, hp, 1, 3, 0, 0, 0, 0, 0, 1, ws, 4096, 0) == E_ALIGN      # in place
    assert st(p, p, p, 8, q, q, q, 8, 0, p, hp, 1, 3, 0, 0, 0, 0, 0, 1, ws + 4, 4096, 0) == E_ALIGN
    assert st(p, p, p, 8, q, q, q, 8, 8, p, hp, 1, 3, 0, 0, 0, 0, 0, 1, ws, 256, 0) == E_SIZE        # workspace too small
    assert st(p, p, p, 8, q, q, q, 8, 0, p, 0, 1, 3, 0, 0, 0, 0, 0, 1, ws, 4096, 0) == E_NULL        # needs the host copy
    assert st(p, p, p, 8, q, q, q, 8, 0, p, hp, 1, 3, 0, 0, 0, 0, 0, 3, ws, 4096, 0) == E_SIZE       # parts: a power of two
    assert st(p, p, p, 8, q, q, q, 8, 0, p, hp, 1, 3, 0, 0, 0, 0, 0, 128, ws, 4096, 0) == E_SIZE     # at most 64
    assert lib.rtt_simulate_nearest_f32(p, p, p, 8, q, q, q, 8, 0, p, hp, 1, 3, 0, 0, 0, 0, 0, 0, 0) == E_SIZE
    assert lib.rtt_simulate_nearest_f32(p, p, p, 8, q, q, q, 8, 0, p, hp, 1, 3, 0, 0, 0, 0, 0, 64, 0) == 0
    fres = rec.copy()
    fres["phys_kind"] = P.PHYS_FRESNEL
    assert st(p, p, p, 8, q, q, q, 8, 0, p, fres.ctypes.data, 1, 3, 0, 0, 0, 0, 0, 1, ws, 4096, 0) == E_KIND
    # generate-and-trace: two-uniform sources with an even offset only
    src.kind, src.offset = 7, 0
    assert lib.rtt_sample_trace_seq_f32(sp, p, p, p, 8, 0, 0, 8, p, hp, 1, 0, 0, 0, 0, 0) == E_KIND
    src.kind, src.offset = 1, 3
    assert lib.rtt_sample_trace_seq_f32(sp, p, p, p, 8, 0, 0, 8, p, hp, 1, 0, 0, 0, 0, 0) == E_ALIGN
    src.offset = 2
    assert lib.rtt_sample_trace_seq_f32(sp, p, p, p, 4, 0, 0, 8, p, hp, 1, 0, 0, 0, 0, 0) == E_SIZE
    assert lib.rtt_sample_trace_seq_f32(sp, p, 0, p, 8, 0, 0, 8, p, hp, 1, 0, 0, 0, 0, 0) == E_NULL
    assert lib.rtt_sample_trace_seq_f32(sp, p, p, p, 8, 0, 0, 0, p, hp, 1, 0, 0, 0, 0, 0) == 0
    assert lib.rtt_strerror(E_KIND).decode().startswith("surface record")
    assert lib.rtt_tune(0, 3) == E_SIZE and lib.rtt_tune(0, 2) == 0


def test_no_cpu_fallback(ns):
    """CPU tensors are refused loudly by every compute entry of the product API."""
    torch.manual_seed(0)
    lens, sensor = scenes.spot_system(ns, grads=False)
    rays = scenes.disk_bundle(ns).sample(16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ns.SequentialScene([lens, sensor]).simulate(rays)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lens.intersectTest(rays)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lens(rays, 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lens.shape.surfaces[0](rays)
    scene = ns.Scene(); scene.add_element(lens); scene.rays = rays
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        scene.step()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ns.RefractSnell(1.0, 1.5)(rays.pos, rays.dir, rays.dir)


def test_missing_library_is_an_error(monkeypatch):
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", "/nonexistent/librtt_hip.so")
    with pytest.raises(_native.NativeLibraryError, match="no CPU fallback"):
        _native.lib()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "raytracetorch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "rtt_oracle" not in src and "import oracle" not in src and "ref_loader" not in src, f


# ---- Rays container (rays/ray.py:7-97 of the reference) -------------------------------------------
def test_rays_container_semantics(ns):
    r = ns.Rays.initialize([[0, 0, -10], [1, 1, -5], [2, 0, 0]], [[0, 0, 2], [0, 3, 4], [0, 0, 0]], ray_id=5)
    assert r.pos.shape == (3, 3) and r.dir.shape == (3, 3) and r.batch_size == torch.Size([3])
    assert r.id.dtype == torch.int8 and int(r.id[0]) == 5
    assert torch.equal(r.intensity, torch.ones(3)) and torch.equal(r.wavelength, torch.zeros(3))
    assert torch.allclose(r.dir, torch.tensor([[0, 0, 1.0], [0, 0.6, 0.8], [0, 0, 0]]))  # normalised, zero stays zero
    assert r._pos.is_contiguous() and r._pos.shape == (3, 3)  # planes
    assert r.pos.data_ptr() == r._pos.data_ptr()  # [N,3] view, no copy
    sub = r[torch.tensor([True, False, True])]
    assert sub.pos.shape == (2, 3) and torch.equal(sub.pos[1], torch.tensor([2.0, 0, 0]))
    r.scatter_update(torch.tensor([False, True, False]), torch.tensor([[9.0, 9, 9]]), torch.tensor([[0.0, 2, 0]]),
                     torch.tensor([0.5]))
    assert torch.equal(r.pos[1], torch.tensor([9.0, 9, 9])) and torch.equal(r.dir[1], torch.tensor([0.0, 2, 0]))
    assert torch.equal(r.intensity, torch.tensor([1.0, 0.5, 1.0]))
    w = r.with_coords(r.pos + 1, r.dir * 3)
    assert torch.allclose(w.dir[1], torch.tensor([0.0, 1.0, 0.0])) and w.intensity is r.intensity
    r.pos = torch.zeros(3, 3)
    assert torch.equal(r._pos, torch.zeros(3, 3))
    with pytest.raises(ValueError):
        ns.Rays(pos=torch.zeros(3, 3), dir=torch.zeros(3, 3), intensity=torch.ones(3), id=torch.zeros(3),
                wavelength=torch.zeros(3), batch_size=[4])


def test_bundles_sample_expected_geometry(ns):
    torch.manual_seed(0)
    r = scenes.disk_bundle(ns, radius=4.0, z=-30.0).sample(20000)
    rad = r.pos[:, :2].norm(dim=1)
    assert rad.max() <= 4.0 and torch.all(r.pos[:, 2] == -30.0) and torch.all(r.dir[:, 2] == 1.0)
    assert abs(float((rad < 2.0).float().mean()) - 0.25) < 0.02  # uniform in area
    tilted = scenes.disk_bundle(ns, rot=[-0.087, 0.0, 0.0]).sample(8)
    assert torch.allclose(tilted.dir[:, 1], torch.full((8,), float(np.sin(0.087))), atol=1e-6)
    ps = ns.PointSource(NA=0.5, ray_id=1).sample(5000)
    assert torch.allclose(ps.dir.norm(dim=1), torch.ones(5000), atol=1e-6) and ps.dir[:, 2].min() >= np.sqrt(0.75) - 1e-6
    fan = ns.Fan(angle=0.4, ray_id=0).sample(100)
    assert torch.all(fan.dir[:, 0] == 0) and fan.dir[:, 1].abs().max() <= np.sin(0.2) + 1e-6
    line = ns.CollimatedLine(length=2.0, ray_id=0).sample(100)
    assert line.pos[:, 0].abs().max() <= 2.0 and torch.all(line.pos[:, 1:] == 0)


# ---- scene compiler ------------------------------------------------------------------------------------
def test_singlet_program(ns):
    lens = scenes.singlet(ns, transform=scenes.T(ns, 7.0))
    prog = P.compile_elements([lens])
    assert prog.shape == (3,) and prog.dtype == P.SURFACE_DTYPE
    assert list(prog["surf_kind"]) == [P.SURF_QUADRIC, P.SURF_QUADRIC, P.SURF_CYLINDER]
    assert list(prog["bound_kind"]) == [P.BND_HALF, P.BND_HALF, P.BND_NONE]
    assert list(prog["shape_kind"]) == [P.SHP_SPHERIC_FACE, P.SHP_SPHERIC_FACE, P.SHP_SPHERIC_EDGE]
    assert list(prog["phys_kind"]) == [P.PHYS_SNELL, P.PHYS_SNELL, P.PHYS_BLOCK]
    allid = P.F_HAS_SHAPE | P.F_SHAPE_ROT_ID | P.F_SURF_ROT_ID
    assert all(int(f) == allid for f in prog["flags"])
    assert np.allclose(prog["ts"], [[0, 0, 7.0]] * 3)
    assert np.allclose(prog["tj"][:, 2], [-1.5, 1.5, 0.0])
    # elements/lens.py:41-49: front face ior_in = glass, ior_out = media; back face the other way round
    assert np.allclose(prog["phys"][0], [1.5, 1.0, 1.0 / 1.5, 1.5])
    assert np.allclose(prog["phys"][1], [1.0, 1.5, 1.5, 1.0 / 1.5])
    assert np.allclose(prog["geom"][0][:3], [0.05, 0.0, 0.05]) and np.allclose(prog["geom"][2][:2], [5.0, 25.0])
    assert np.isclose(prog["shp"][0][0], 25.0)
    sag = 0.05 * 25 / (1 + np.sqrt(1 - 0.05 ** 2 * 25))
    assert np.allclose(prog["shp"][2][:2], [sag - 1.5, 1.5 - sag], atol=1e-6)


def test_program_tracks_live_parameters(ns):
    lens = scenes.singlet(ns, c1_grad=True)
    a = P.compile_elements([lens])
    with torch.no_grad():
        lens.shape.surfaces[0].c.fill_(0.02)
        lens.ior_glass.fill_(1.7)
    b = P.compile_elements([lens])
    assert np.isclose(b["geom"][0][0], 0.02) and not np.isclose(a["geom"][0][0], 0.02)
    assert np.isclose(b["phys"][0][0], 1.7) and np.isclose(b["phys"][1][1], 1.7)  # shared Parameter on both faces


def test_system_and_naive_programs(ns):
    prog = P.compile_elements(scenes.system5(ns))
    assert prog.shape[0] == 9
    assert list(prog["element_id"]) == [0, 0, 0, 1, 2, 2, 2, 3, 4]
    stop = prog[3]
    assert stop["surf_kind"] == P.SURF_PLANE and stop["bound_kind"] == P.BND_DISK and stop["phys_kind"] == P.PHYS_APERTURE
    assert int(stop["flags"]) & P.F_INVERT and not int(stop["flags"]) & P.F_HAS_SHAPE
    mirror = prog[7]
    assert mirror["bound_kind"] == P.BND_HALF_DISK and mirror["phys_kind"] == P.PHYS_REFLECT and np.isclose(mirror["bnd"][0], 225.0)
    sensor = prog[8]
    assert int(sensor["flags"]) & P.F_SENSOR and sensor["sensor_slot"] == 0 and P.n_sensor_slots(prog) == 1
    prog = P.compile_elements(scenes.naive32(ns))
    assert prog.shape[0] == 32
    side = prog[20]  # first side plane of the first cylindrical singlet
    assert side["surf_kind"] == P.SURF_PLANE and side["shape_kind"] == P.SHP_CYL_EDGE
    assert not int(side["flags"]) & P.F_SURF_ROT_ID and int(side["flags"]) & P.F_SHAPE_ROT_ID
    assert np.allclose(side["shp"][:4], [7.00001, -7.00001, 7.00001, -7.00001], atol=1e-6)


def test_constructor_validation(ns):
    with pytest.raises(ValueError, match="R1"):
        ns.SingletLens(c1=0.5, c2=-0.05, d=10.0, t=3.0, ior_glass=1.5)
    with pytest.raises(ValueError, match="Thickness"):
        ns.SingletLens(c1=0.05, c2=-0.05, d=10.0, t=0.0, ior_glass=1.5)
    with pytest.raises(ValueError, match="ntersect"):
        ns.SingletLens(c1=0.15, c2=-0.15, d=10.0, t=0.5, ior_glass=1.5)
    fres = ns.SingletLens(c1=0.05, c2=-0.05, d=10.0, t=3.0, ior_glass=1.5, fresnel=True)
    assert [type(f).__name__ for f in fres.surface_functions] == ["RefractFresnel"] * 3   # elements/lens.py:36-56
    assert int(fres._record(0)["phys_kind"]) == P.PHYS_FRESNEL
    # a program may be longer than a launch takes: the host cuts it between elements, sensor slots numbered per piece
    many = [scenes.singlet(ns) for _ in range(21)] + [ns.Sensor(ns.Disk(5.0)), scenes.singlet(ns), ns.Sensor(ns.Disk(5.0))]
    long = P.compile_elements(many)
    assert long.shape[0] == 68 and P.n_sensor_slots(long) == 2
    pieces = P.split_program(long)
    assert [(p.shape[0], first, slots) for p, first, slots in pieces] == [(64, 0, 0), (4, 64, 1)]
    assert P.split_program(long) is pieces                      # made once per program
    assert [int(s) for s in pieces[0][0]["sensor_slot"][-1:]] == [0] and int(pieces[1][0]["sensor_slot"][-1]) == 0
    assert len(P.split_program(long[:30])) == 1
    with pytest.raises(ValueError, match="more than 2 surfaces"):
        P.split_program(long, limit=2)


def test_scene_bookkeeping(ns):
    scene = ns.Scene()
    assert scene.total_surfaces == 0 and scene.ray_cast(None) is None
    els = scenes.system5(ns)
    for el in els:
        scene.add_element(el)
    scene._build_index_maps()
    assert scene.total_surfaces == 9
    assert scene.map_to_element.tolist() == [0, 0, 0, 1, 2, 2, 2, 3, 4] and scene.map_to_surface.tolist() == [0, 1, 2, 0, 0, 1, 2, 0, 0]
    seq = scene.to_sequential()
    zs = [float(e.shape.transform.trans[2]) for e in seq.elements]
    assert zs == sorted(zs) and isinstance(seq, ns.SequentialScene)
    assert len(seq.to_base().elements) == 5
    torch.manual_seed(0)
    scene.add_bundle(scenes.disk_bundle(ns), 10)
    scene.add_bundle(scenes.disk_bundle(ns, ray_id=3), 6)
    scene._build_rays()
    assert len(scene.rays) == 16 and scene.rays.id.tolist() == [0] * 10 + [3] * 6
    scene.clear_bundles()
    assert scene.rays is None


def test_sensor_lists(ns):
    s = ns.Sensor(ns.Disk(5.0))
    hit = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    s._record_dense(hit, torch.tensor([1.0, 0.5, 0.0, 2.0]), torch.tensor([1, 0, 1, 1], dtype=torch.uint8),
                    torch.tensor([0, 0, 1, 1], dtype=torch.int8))
    s._record_dense(hit, torch.ones(4), torch.zeros(4, dtype=torch.uint8), torch.zeros(4, dtype=torch.int8))
    locs, w, ids = s.getHitsTensors()
    assert locs.tolist() == [[0, 4, 8], [2, 6, 10], [3, 7, 11]] and w.tolist() == [1.0, 0.0, 2.0] and ids.tolist() == [0, 1, 1]
    assert len(s.hitLocs) == 1  # the visit nobody hit leaves no entry, like the reference's `continue`
    assert s.getHitsTensors(ray_id=1)[0].shape == (2, 3)
    with pytest.raises(RuntimeError, match="HIP device"):   # the reductions run in a kernel: no CPU path
        s.getSpotSizeParallel_xy([1, 0])
    s.reset()
    assert s.hitLocs == []


def test_names_the_reference_scripts_reach_through_star_imports():
    """The reference's scripts use names its star imports happen to re-export (`rtt.elements.CylSinglet`,
    tests/test_cylindrics.py:106; elements/__init__.py:3-6, phys/__init__.py:1)."""
    import raytracetorch_amd as rtt
    for name in ("RayTransform", "Singlet", "CylSinglet", "Doublet", "Triplet", "Quadric", "QuadricZY", "Disk", "Rectangle",
                 "Ellipse", "HalfSphere", "HalfCyl", "BoundedHalfSphere", "RefractSnell", "RefractFresnel", "Block", "Reflect",
                 "ApertureFilter", "ParaxialDistMat"):
        assert hasattr(rtt.elements, name), name
    assert rtt.phys.RayTransform is rtt.geom.RayTransform


def test_sensor_compacts_pending_records_past_its_limit(ns):
    """Dense per-visit records (17 B per ray) are compacted to the reference's per-hit lists once they pass a size
    limit, in visit order, so a sensor that is never read or reset does not keep whole bundles alive."""
    import torch
    s = ns.Sensor(ns.Disk(1.0))
    s._DENSE_LIMIT = 17 * 25
    for v in range(4):
        hit = torch.arange(30, dtype=torch.float32).reshape(3, 10) + 100 * v
        mask = torch.zeros(10, dtype=torch.uint8)
        mask[v::4] = 1
        s._record_dense(hit, torch.full((10,), float(v + 1)), mask, torch.zeros(10, dtype=torch.int8))
    assert len(s._dense) < 4 and s._dense_bytes == 17 * 10 * len(s._dense)
    locs, inten, _ = s.getHitsTensors()
    assert inten.tolist() == [1.0] * 3 + [2.0] * 3 + [3.0] * 2 + [4.0] * 2
    assert locs[:3, 0].tolist() == [0.0, 4.0, 8.0] and locs[3, 0].item() == 101.0


def test_scene_notices_a_replaced_element_even_if_it_inherits_the_old_id(ns):
    """The packed program is cached per element list; the cache must hold the elements it was built from, or a freed
    element could pass its id on to its replacement and the stale program would be served."""
    import gc
    scene = ns.Scene()
    scene.add_element(ns.SingletLens(c1=0.05, c2=-0.05, d=10.0, t=3.0, ior_glass=1.5))
    first = scene._program().copy()
    for _ in range(50):   # with the cache holding only ids, some iteration reuses the freed element's address
        scene.clear_elements()
        gc.collect()
        scene.add_element(ns.SingletLens(c1=0.02, c2=-0.03, d=10.0, t=3.0, ior_glass=1.5))
        assert abs(float(scene._program()[0]["geom"][0]) - 0.02) < 1e-7
        scene.clear_elements()
        gc.collect()
        scene.add_element(ns.SingletLens(c1=0.05, c2=-0.05, d=10.0, t=3.0, ior_glass=1.5))
        assert abs(float(scene._program()[0]["geom"][0]) - 0.05) < 1e-7
    assert first.shape == scene._program().shape


def test_program_cache_sees_writes_through_data(ns):
    """`p.data.fill_(v)` (the reference's documented way to sweep a parameter, project.py:18-23) leaves the version
    counter of `p` alone; the packed-program cache compares host parameters by value and must see it."""
    import torch
    scene = ns.Scene()
    lens = ns.SingletLens(c1=0.05, c2=-0.05, d=10.0, t=3.0, ior_glass=1.5)
    scene.add_element(lens)
    c = lens.shape.surfaces[0].c
    assert abs(float(scene._program()[0]["geom"][0]) - 0.05) < 1e-7
    v = c._version
    c.data.fill_(0.07)
    assert c._version == v
    assert abs(float(scene._program()[0]["geom"][0]) - 0.07) < 1e-7
    with torch.no_grad():
        c.fill_(0.03)
    assert abs(float(scene._program()[0]["geom"][0]) - 0.03) < 1e-7
    c.data = torch.tensor(0.01)
    assert abs(float(scene._program()[0]["geom"][0]) - 0.01) < 1e-7
    scene.invalidate_program()
    assert abs(float(scene._program()[0]["geom"][0]) - 0.01) < 1e-7


def test_rays_with_planes_still_to_be_made():
    """`SampledRays`: size and device are known without making the planes; the first read of any field makes them once
    and leaves a plain `Rays`; a plain bundle can be turned into a pending one in place (host logic, no GPU)."""
    from raytracetorch_amd.rays.ray import Rays, SampledRays

    class Pending:
        n, device, made = 5, torch.device("cpu"), 0

        def make(self):
            Pending.made += 1
            return (torch.zeros(3, 5), torch.ones(3, 5), torch.full((5,), 2.0), torch.zeros(5, dtype=torch.int8), torch.zeros(5))

    r = SampledRays.pending(None, 5, torch.device("cpu"))
    assert type(r) is SampledRays and len(r) == 5 and r.device.type == "cpu" and tuple(r.batch_size) == (5,)
    assert "N=5" in repr(r) and type(r) is SampledRays            # none of these made anything
    plain = Rays.from_planes(torch.rand(3, 4), torch.rand(3, 4), torch.ones(4), torch.zeros(4, dtype=torch.int8), torch.zeros(4))
    same = SampledRays.defer(plain, Pending())
    assert same is plain and type(plain) is SampledRays and len(plain) == 5 and Pending.made == 0
    assert tuple(plain.pos.shape) == (5, 3) and type(plain) is Rays and Pending.made == 1
    assert float(plain.intensity[0]) == 2.0 and plain.id.dtype == torch.int8 and Pending.made == 1
    other = SampledRays.defer(Rays.from_planes(torch.rand(3, 4), torch.rand(3, 4), torch.ones(4), torch.zeros(4, dtype=torch.int8),
                                               torch.zeros(4)), Pending())
    other.intensity = torch.full((5,), 7.0)                       # a write makes the planes first, then replaces the field
    assert type(other) is Rays and Pending.made == 2 and float(other.intensity[0]) == 7.0 and tuple(other._pos.shape) == (3, 5)
    clone = SampledRays.defer(Rays.from_planes(torch.rand(3, 4), torch.rand(3, 4), torch.ones(4), torch.zeros(4, dtype=torch.int8),
                                               torch.zeros(4)), Pending()).clone()
    assert type(clone) is Rays and len(clone) == 5 and Pending.made == 3


def test_sensor_keeps_pending_hit_lists_in_order(ns):
    """A sensor's entries come out in the order of recording whether a visit is a compacted list, a dense record or
    the pending hit list of a bounce loop nobody has waited for (host logic, no GPU)."""
    s = ns.Sensor(ns.Disk(5.0))
    asked = []

    def entries_of(sensor):
        asked.append(sensor)
        return [(torch.full((2, 3), 1.0), torch.ones(2), torch.zeros(2, dtype=torch.int8)),
                (torch.full((1, 3), 2.0), torch.ones(1), torch.zeros(1, dtype=torch.int8))]
    s._record_hits(torch.zeros(1, 3), torch.ones(1), torch.zeros(1, dtype=torch.int8))
    s._record_pending(entries_of)
    hit = torch.full((3, 4), 3.0)
    s._record_dense(hit, torch.ones(4), torch.tensor([1, 0, 0, 1], dtype=torch.uint8), torch.zeros(4, dtype=torch.int8))
    assert len(s._pending) == 1 and not asked
    locs = s.hitLocs
    assert asked == [s] and not s._pending and not s._dense
    assert [float(l[0, 0]) for l in locs] == [0.0, 1.0, 2.0, 3.0] and [l.shape[0] for l in locs] == [1, 2, 1, 2]
    s._record_pending(entries_of)
    s.reset()
    assert not s._pending and len(asked) == 1 and not s.hitLocs
    for _ in range(s._PENDING_LIMIT + 1):      # never read, never reset: the old entries are settled, the newest stays pending
        s._record_pending(entries_of)
    assert len(s._pending) == 1 and len(s._locs) == 2 * s._PENDING_LIMIT
    # the entries are bounded by the device memory they keep alive, not only by their number: three launches that hold
    # 0.4 GiB each pass the 1 GiB limit with the third, which settles the first two
    s.reset()
    for _ in range(3):
        s._record_pending(entries_of, nbytes=int(0.4 * 2 ** 30))
    assert len(s._pending) == 1 and len(s._locs) == 4 and s._pending_bytes == int(0.4 * 2 ** 30)
    s._record_pending(entries_of, nbytes=2 ** 31)     # one entry larger than the limit is still taken (after settling the rest)
    assert len(s._pending) == 1 and len(s._locs) == 6


def test_long_programs_are_cut_into_stretches_for_the_backward(ns, monkeypatch):
    """Which programs a differentiable trace sweeps in one piece, and where it cuts the others (whole elements, at most
    12 surfaces per stretch in reverse mode, 64 in forward mode; sensor slots numbered per stretch)."""
    import raytracetorch_amd.autograd as AG
    monkeypatch.delenv("RTT_BACKWARD", raising=False)
    short = scenes.triplet_train(ns, 3, grads=True)
    assert AG._stretches(short, P.compile_elements(short)) is None                       # 21 surfaces: the sweep's LDS holds them
    mid = scenes.triplet_train(ns, 4, grads=True, sensors=(1, 3))
    cut = AG._stretches(mid, P.compile_elements(mid))
    assert [(len(e), p.shape[0], P.n_sensor_slots(p)) for e, p in cut] == [(1, 7, 0), (2, 8, 1), (1, 7, 0), (2, 8, 1)]
    assert [e for es, _ in cut for e in es] == mid and all(int(p["sensor_slot"].max()) <= 0 for _, p in cut)
    long = scenes.triplet_train(ns, 10, grads=True, sensors=(2, 9))
    prog = P.compile_elements(long)
    assert sum(p.shape[0] for _, p in AG._stretches(long, prog)) == 72
    assert not AG.one_sweep(prog, len(long)) and AG.one_sweep(prog[:21], 3) and AG.one_sweep(prog[:28], 1)   # (a single element cannot be cut)
    monkeypatch.setenv("RTT_BACKWARD", "tangent")
    assert AG._stretches(mid, P.compile_elements(mid)) is None                           # forward mode has no LDS budget
    assert [p.shape[0] for _, p in AG._stretches(long, prog)] == [64, 8]


def test_rotation_matrix_keeps_its_derivative_ready():
    """exp(skew(v)) of a trainable rotation: the value of torch.linalg.matrix_exp bit for bit, its derivative from the
    three directional derivatives made in the forward (plain, batched as the slot Jacobians ask for it, float64 check)."""
    from raytracetorch_amd.geom.transform import _skew, _skew_exp
    for vec in ([0.3, -0.2, 0.5], [0.0, 0.0, 0.0], [1e-4, 2.0, -1.0]):
        v = torch.tensor(vec, requires_grad=True)
        R, ref = _skew_exp(v), torch.linalg.matrix_exp(_skew(v))
        assert torch.equal(R, ref) and R.grad_fn is not None
        w = torch.randn(3, 3, generator=torch.Generator().manual_seed(1))
        (g,), (g_ref,) = torch.autograd.grad((R * w).sum(), v), torch.autograd.grad((ref * w).sum(), v)
        assert float((g - g_ref).abs().max()) <= 2e-6 * max(1.0, float(g_ref.abs().max()))
        eye = torch.eye(9)
        (j,) = torch.autograd.grad(_skew_exp(v).reshape(-1), v, grad_outputs=eye, is_grads_batched=True)
        (j_ref,) = torch.autograd.grad(torch.linalg.matrix_exp(_skew(v)).reshape(-1), v, grad_outputs=eye, is_grads_batched=True)
        assert float((j - j_ref).abs().max()) <= 2e-6
    v64 = torch.tensor([0.3, -0.2, 0.5], dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(_skew_exp, (v64,))
    with torch.no_grad():
        assert _skew_exp(v).grad_fn is None      # nothing extra is computed when no gradient can be asked for
