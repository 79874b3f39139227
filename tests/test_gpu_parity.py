"""GPU: the HIP path, called through the product API (ctypes -> C ABI -> kernels), reproduces the
golden vectors recorded from the reference and agrees with the CPU oracle.

Tolerance: 1e-5 relative (float32), per ray relative to the length of the reference vector;
discrete results (hit / miss patterns, winners, intensity 0/1) must be identical.  The few cases
ill-conditioned rays on which two correct float32 evaluations differ by more are judged against a
float64 evaluation by the oracle, see util.assert_parity.
"""
import numpy as np
import pytest
import torch

import rtt_oracle as O
import scenes
from functools import partial

import util
from util import INF, PARITY_TOL, golden, is_hard_case, pattern_mismatch, rays_from_golden, rel_err

import raytracetorch_amd.program as P

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

PRIMS = ["plane_c1", "plane_posed", "sphere", "cylinder", "quadric_parabola", "quadric_hyperbola", "quadric_zy",
         "cone", "single_cone", "half_sphere", "bounded_half_sphere", "half_cyl", "disk_inverted", "rectangle",
         "ellipse", "kat_plane", "kat_sphere", "kat_parabola", "kat_implicit"]
SEQ = ["singlet", "singlet_tilted_bundle", "singlet_decentred_bundle", "singlet_posed", "singlet_plano",
       "singlet_clear_edge", "singlet_tir", "singlet_sensor", "system5", "doublet", "triplet", "dead_rays",
       "aperture_plain", "aperture_invert", "mirrors", "cyl_singlet", "ideal_thin_lens", "ideal_mirror", "box"]


def cpu(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("name", PRIMS)
def test_primitive_forward(ns, name):
    assert_parity = partial(util.assert_parity, hard=is_hard_case(name))
    g = golden("prim_" + name)
    cases = dict(scenes.primitive_cases(ns)); cases.update(scenes.kat_rays(ns))
    surf = cases[name][0]()
    rays = rays_from_golden(ns, g, device=DEV)
    t_test = cpu(surf.intersectTest(rays))
    t, hit, nrm, hl = (cpu(x) for x in surf(rays))
    S64 = O._records(surf._surface_record()[None], torch.float64)[0]
    p64, d64 = torch.from_numpy(g["in_pos"]).double(), torch.from_numpy(g["in_dir"]).double()
    tt64 = O.test_column(S64, p64, d64).numpy()
    t64, hit64, nrm64, hl64 = (x.numpy() for x in O.surface_forward(S64, p64, d64))
    assert t_test.shape == g["t_test"].shape
    assert pattern_mismatch(t_test, g["t_test"]) == 0
    assert_parity(t_test[:, 0], g["t_test"][:, 0], tt64, what=f"{name} t_test")
    assert pattern_mismatch(t, g["t"]) == 0
    assert_parity(t, g["t"], t64, what=f"{name} t")
    m = np.isfinite(g["t"])
    for got, ref, tr, w in ((hit, g["hit_global"], hit64, "hit_global"), (nrm, g["normal_global"], nrm64, "normal"),
                            (hl, g["hit_local"], hl64, "hit_local")):
        assert_parity(got[m], ref[m], tr[m], what=f"{name} {w}")


@pytest.mark.parametrize("name", SEQ)
def test_sequential_trace(ns, name):
    assert_parity = partial(util.assert_parity, hard=is_hard_case(name))
    g = golden("seq_" + name)
    elements = scenes.sequential_cases(ns)[name][0]()
    prog = P.compile_elements(elements)
    rays0 = rays_from_golden(ns, g, device=DEV)
    p0, d0, I0 = g["in_pos"], g["in_dir"], g["in_intensity"]
    tm64 = O.intersect_matrix(prog, p0, d0, dtype=torch.float64).numpy()
    # Element.intersectTest per element at the initial state
    col = 0
    for k, el in enumerate(elements):
        got = cpu(el.intersectTest(rays0))
        assert got.shape == g[f"it_{k}"].shape
        assert pattern_mismatch(got, g[f"it_{k}"]) == 0, (name, k)
        assert_parity(got.reshape(-1), g[f"it_{k}"].reshape(-1), tm64[:, col:col + got.shape[1]].reshape(-1),
                      what=f"{name} intersectTest {k}")
        col += got.shape[1]
    # Element.__call__(rays[mask], j) of the first element
    el = elements[0]
    for j in range(len(el.shape)):
        m = g[f"fw_{j}_mask"]
        if not m.any():
            continue
        sub = rays0[torch.from_numpy(m).to(DEV)]
        pos, d, im = el(sub, j)
        if hasattr(el, "reset"):
            el.reset()
        S64 = O._records(prog[j:j + 1], torch.float64)[0]
        hp64, nd64, _, _ = O.element_forward(S64, torch.from_numpy(p0[m]).double(), torch.from_numpy(d0[m]).double())
        assert_parity(cpu(pos), g[f"fw_{j}_pos"], hp64.numpy(), what=f"{name} fw{j} pos")
        assert_parity(cpu(d), g[f"fw_{j}_dir"], nd64.numpy(), what=f"{name} fw{j} dir")
        assert np.array_equal(cpu(im), g[f"fw_{j}_imult"])
    # element-prefix states
    for k in range(len(elements)):
        for e in elements:
            if hasattr(e, "reset"):
                e.reset()
        r = ns.SequentialScene(elements[:k + 1]).simulate(rays0.clone())
        o64 = O.trace_sequential(prog[:sum(len(e.shape) for e in elements[:k + 1])], p0, d0, I0, dtype=torch.float64)
        for f, got in (("pos", r.pos), ("dir", r.dir), ("intensity", r.intensity)):
            assert_parity(cpu(got), g[f"after_{k}_{f}"], o64[f].numpy(), what=f"{name} after {k} {f}")
    # full trace: in place, returns the same object
    for e in elements:
        if hasattr(e, "reset"):
            e.reset()
    r_in = rays0.clone()
    r = ns.SequentialScene(elements).simulate(r_in)
    assert r is r_in
    assert np.array_equal(cpu(r.intensity) > 0, g["out_intensity"] > 0)
    o64 = O.trace_sequential(prog, p0, d0, I0, dtype=torch.float64)
    for f, got in (("pos", r.pos), ("dir", r.dir), ("intensity", r.intensity)):
        assert_parity(cpu(got), g[f"out_{f}"], o64[f].numpy(), what=f"{name} out {f}")
    assert torch.equal(r.id.cpu(), torch.from_numpy(g["out_id"]))
    # sensor hit lists, in ray order (float64 evaluation of the same records as the tie-breaker for hit points
    # whose local coordinates are small differences of large global ones)
    truth64 = [hl[mask].numpy() for _, mask, hl, _ in o64["sensor"]]
    slot = -1
    for k, e in enumerate(elements):
        if not hasattr(e, "getHitsTensors"):
            continue
        slot += 1
        if f"sensor_{k}_locs" not in g:
            assert not e.hitLocs
            continue
        locs, w, ids = e.getHitsTensors()
        assert locs.shape == g[f"sensor_{k}_locs"].shape
        t64 = truth64[slot] if slot < len(truth64) and truth64[slot].shape == g[f"sensor_{k}_locs"].shape else None
        assert_parity(cpu(locs), g[f"sensor_{k}_locs"], t64, what=f"{name} sensor locs")
        assert_parity(cpu(w), g[f"sensor_{k}_w"], what=f"{name} sensor w")
        assert np.array_equal(cpu(ids), g[f"sensor_{k}_id"])
    # agreement with the CPU oracle on the same program (tighter than the golden tolerance is not
    # required; this guards against the oracle and the kernels drifting apart)
    o = O.trace_sequential(prog, p0, d0, I0)
    assert_parity(cpu(r.pos), o["pos"].numpy(), o64["pos"].numpy(), what=f"{name} vs oracle pos")
    assert_parity(cpu(r.dir), o["dir"].numpy(), o64["dir"].numpy(), what=f"{name} vs oracle dir")


def test_out_of_place_equals_in_place(ns):
    g = golden("seq_system5")
    elements = scenes.system5(ns)
    a = rays_from_golden(ns, g, device=DEV)
    b = a.clone()
    out = a.clone()
    scene = ns.SequentialScene(elements)
    r = scene.simulate(a, out=out)
    assert r is out
    assert torch.equal(a.pos, b.pos) and torch.equal(a.dir, b.dir)  # input untouched
    scene.simulate(b)
    assert torch.equal(out.pos, b.pos) and torch.equal(out.dir, b.dir) and torch.equal(out.intensity, b.intensity)


def test_naive_ray_cast_and_step(ns):
    g = golden("naive32")
    scene = ns.Scene()
    for el in scenes.naive32(ns):
        scene.add_element(el)
    scene._build_index_maps()
    assert scene.total_surfaces == 32
    rays = rays_from_golden(ns, g, device=DEV)
    tm = cpu(torch.cat([el.intersectTest(rays) for el in scene.elements], dim=1))
    assert pattern_mismatch(tm, g["t_matrix"]) == 0
    # a benchmark configuration: purely relative bounds (floor 0), no tie-break
    assert rel_err(tm.reshape(-1), g["t_matrix"].reshape(-1), floor=0.0) < PARITY_TOL
    hit, el_ids, sf_ids = scene.ray_cast(rays)
    hit, el_ids, sf_ids = cpu(hit), cpu(el_ids), cpu(sf_ids)
    assert np.array_equal(hit, g["hit_mask"])
    # winners must agree except where the two smallest distances are closer than float32 noise
    srt = np.sort(g["t_matrix"], axis=1)
    tie = np.isfinite(srt[:, 1]) & (np.abs(srt[:, 1] - srt[:, 0]) <= 1e-4 * np.abs(srt[:, 0]))
    ok = hit & ~tie
    assert tie.sum() <= 0.001 * len(tie)
    assert np.array_equal(el_ids[ok], g["el_ids"][ok]) and np.array_equal(sf_ids[ok], g["sf_ids"][ok])
    scene.rays = rays
    scene.step()
    r = scene.rays
    ok = ~tie
    for f, got in (("pos", r.pos), ("dir", r.dir), ("intensity", r.intensity)):
        assert rel_err(cpu(got)[ok], g[f"step1_{f}"][ok], floor=0.0) < PARITY_TOL, f
    sensor = scene.elements[-1]
    if "sensor_locs" in g:
        locs, w, _ = sensor.getHitsTensors()
        assert locs.shape == g["sensor_locs"].shape
        assert rel_err(cpu(locs), g["sensor_locs"]) < PARITY_TOL


def test_spot_size_loss_value(ns):
    g = golden("spot_f32")
    torch.manual_seed(0)
    lens, sensor = scenes.spot_system(ns, grads=False)
    bundle = scenes.disk_bundle(ns)
    loss_fn = ns.SpotSizeLoss(sensor, [bundle], N_rays=65536)
    scene = ns.SequentialScene([lens, sensor])
    loss = loss_fn(scene)
    assert abs(float(loss) - float(g["loss"])) / float(g["loss"]) < PARITY_TOL
    # the base Scene (what the reference's SpotSizeLoss drives) must at least run and return a finite loss
    torch.manual_seed(0)
    base = scene.to_base()
    base.Nbounces = 1
    assert np.isfinite(float(loss_fn(base)))


def test_physics_functors_direct(ns):
    """SurfaceFunction.forward on raw tensors (the reference's tests/test_phys_inf_plane.py usage)."""
    torch.manual_seed(3)
    n = 1000
    d = torch.nn.functional.normalize(torch.randn(n, 3), dim=1)
    nrm = torch.nn.functional.normalize(torch.randn(n, 3) * 0.2 + torch.tensor([0.0, 0.0, 1.0]), dim=1)
    hl = torch.randn(n, 3)
    for mk, kind in ((lambda: ns.Reflect(), O.P_REFLECT), (lambda: ns.RefractSnell(1.0, 1.4), O.P_SNELL),
                     (lambda: ns.RefractSnell(1.8, 1.0), O.P_SNELL), (lambda: ns.Block(), O.P_BLOCK),
                     (lambda: ns.Transmit(), O.P_TRANSMIT)):
        fn = mk()
        rec = P.new_record(); fn._fill_physics(rec)
        S = O._records(rec[None], torch.float32)[0]
        want_d, want_m = O.apply_physics(S, hl, d, nrm)
        got_d, got_m = fn(hl.to(DEV), d.to(DEV), nrm.to(DEV))
        assert rel_err(cpu(got_d), want_d.numpy()) < PARITY_TOL
        assert np.array_equal(cpu(got_m), want_m.numpy())


def test_paths_history(ns):
    """Paths wrapper (rays/ray.py:100-207): the first bounce of the nearest-hit tracer records the reference's
    snapshot; a sequential trace records one snapshot per surface that was hit, equal to the prefix traces."""
    g = golden("paths")
    scene = ns.Scene()
    for el in [scenes.singlet(ns), ns.Sensor(ns.Disk(20.0, transform=scenes.T(ns, 19.0)))]:
        scene.add_element(el)
    scene._build_index_maps()
    scene.rays = ns.Paths(rays_from_golden(ns, g, device=DEV))
    scene.step()
    hist = scene.rays.get_history()
    assert len(hist) == g["step_history"].shape[0] == 2
    assert not hist[0].is_cuda and torch.equal(hist[0], torch.from_numpy(g["in_pos"]))
    assert rel_err(hist[1].numpy(), g["step_history"][1]) < PARITY_TOL
    assert rel_err(cpu(scene.rays.unwrap().pos), g["step1_pos"]) < PARITY_TOL
    # sequential: system5 (9 surfaces); every snapshot equals the trace of the program prefix
    elements = scenes.system5(ns)
    seq = ns.SequentialScene(elements)
    prog = seq._program()
    rays0 = rays_from_golden(ns, g, device=DEV)
    paths = ns.Paths(rays0.clone())
    assert seq.simulate(paths) is paths
    plain = seq.simulate(rays0.clone())
    assert rel_err(cpu(paths.unwrap().pos), cpu(plain.pos)) < PARITY_TOL and rel_err(cpu(paths.unwrap().dir), cpu(plain.dir)) < PARITY_TOL
    hist = paths.get_history()
    p0, d0, I0 = g["in_pos"], g["in_dir"], g["in_intensity"]
    hit_surfaces = [s for s in range(len(prog))
                    if bool((O.intersect_matrix(prog[s:s + 1], *_state_before(prog, s, p0, d0, I0)) < INF).any())]
    assert len(hist) == 1 + len(hit_surfaces)
    for h, s in zip(hist[1:], hit_surfaces):
        want = O.trace_sequential(prog[:s + 1], p0, d0, I0)["pos"].numpy()
        assert rel_err(h.numpy(), want) < PARITY_TOL, s


def _state_before(prog, s, p0, d0, I0):
    if s == 0:
        return p0, d0
    o = O.trace_sequential(prog[:s], p0, d0, I0)
    return o["pos"].numpy(), o["dir"].numpy()


def test_sensor_spot_sizes_per_id(ns):
    """Sensor.getSpotSizeParallel_xy (elements/sensor.py:87-176) through the segmented-reduction kernel, for
    dense trace records and for the per-hit lists of direct Sensor.forward calls."""
    g = golden("sensor_groups")
    lens, sensor = scenes.spot_system(ns, grads=False)
    scene = ns.SequentialScene([lens, sensor])
    for tag in ("a_", "b_"):
        scene.simulate(rays_from_golden(ns, g, prefix=tag, device=DEV))
    tgt = torch.from_numpy(g["target"]).to(DEV)
    for tag, kw in (("centroid_p2", {}), ("centroid_p3", dict(norm_ord=3)), ("target_p2", dict(target_xy=tgt))):
        spot, wsum = sensor.getSpotSizeParallel_xy([3, 0, 7], **kw)
        assert np.array_equal(cpu(wsum), g[f"{tag}_wsum"]), tag
        assert rel_err(cpu(spot), g[f"{tag}_spot"]) < 1e-4, tag     # float32 scatter_add sums on the reference side
    # the same hits as compacted lists (what Sensor.forward on a ray subset records)
    _ = sensor.hitLocs
    assert not sensor._dense and len(sensor._locs) == 2
    spot2, wsum2 = sensor.getSpotSizeParallel_xy([3, 0, 7])
    assert np.array_equal(cpu(wsum2), g["centroid_p2_wsum"]) and rel_err(cpu(spot2), g["centroid_p2_spot"]) < 1e-4
    sensor.reset()
    with pytest.raises(RuntimeError):
        sensor.getSpotSizeParallel_xy([0])


def test_host_copy_cache_ignores_an_entry_left_by_a_freed_tensor():
    """Host copies of device tensors are cached under id(); a freed temporary can pass its id, storage address and
    version counter on to a new tensor of another shape, which must not be served the old value."""
    import weakref
    p = torch.arange(9.0, device=DEV)
    other = torch.zeros(1, device=DEV)
    P._host_cache[id(p)] = ((p.data_ptr(), p._version), np.zeros(1, np.float32), weakref.ref(other))
    assert P.host_value(p).tolist() == list(range(9))
    assert P.host_value(p).tolist() == list(range(9))   # now served from its own entry
    del other


def test_device_parameters_in_place_writes_and_invalidate_program(ns):
    """Device-resident parameters are tracked by storage and version counter: a write through the parameter is seen
    by the next trace, a write through `.data` after `invalidate_program()`."""
    scene = ns.Scene()
    lens = ns.SingletLens(c1=0.05, c2=-0.05, d=10.0, t=3.0, ior_glass=1.5)
    scene.add_element(lens)
    scene.to(DEV)
    c = lens.shape.surfaces[0].c
    assert c.is_cuda and abs(float(scene._program()[0]["geom"][0]) - 0.05) < 1e-7
    with torch.no_grad():
        c.fill_(0.03)
    assert abs(float(scene._program()[0]["geom"][0]) - 0.03) < 1e-7
    c.data.fill_(0.07)
    scene.invalidate_program()
    assert abs(float(scene._program()[0]["geom"][0]) - 0.07) < 1e-7
