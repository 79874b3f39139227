"""GPU: scenes with more surfaces than one launch takes (64).  The host cuts their programs between elements and chains
or merges the launches of the pieces; the results are held against the CPU oracle, which walks the whole program in
one go (scene/sequential.py:12-36, scene/base.py:144-235 of the reference put no limit on the number of surfaces).

Tolerance: 1e-5 relative per ray (util.assert_parity; float64 evaluation of the same records as the tie-break for
ill-conditioned rays); winners, hit patterns and intensities 0/1 identical."""
import numpy as np
import pytest
import torch

import rtt_oracle as O
import scenes
import util
from util import PARITY_TOL, rel_err

import raytracetorch_amd.autograd as AG
import raytracetorch_amd.program as P
from raytracetorch_amd import _native as N

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def cpu(t):
    return t.detach().cpu().numpy()


def _beam(ns, n, seed=0):
    torch.manual_seed(seed)
    return scenes.disk_bundle(ns, radius=4.0, z=-20.0, rot=[0.01, -0.02, 0.0]).sample(n)


@pytest.mark.parametrize("sensors", [(), (2, 9)], ids=["ten_triplets", "ten_triplets_two_sensors"])
def test_sequential_trace_of_seventy_surfaces(ns, sensors):
    els = scenes.triplet_train(ns, 10, sensors=sensors)
    prog = P.compile_elements(els)
    assert prog.shape[0] == 70 + len(sensors) > P.MAX_SURFACES
    assert len(P.split_program(prog)) == 2
    src = _beam(ns, 20000)
    p0, d0, I0 = cpu(src.pos), cpu(src.dir), cpu(src.intensity)
    o = O.trace_sequential(prog, p0, d0, I0)
    o64 = O.trace_sequential(prog, p0, d0, I0, dtype=torch.float64)
    assert float((o["intensity"] > 0).float().mean()) > 0.9      # the beam gets through: every surface is exercised
    scene = ns.SequentialScene(els)
    r_in = src.to(DEV)
    r = scene.simulate(r_in)
    assert r is r_in
    assert np.array_equal(cpu(r.intensity) > 0, o["intensity"].numpy() > 0)
    for f, got in (("pos", r.pos), ("dir", r.dir), ("intensity", r.intensity)):
        util.assert_parity(cpu(got), o[f].numpy(), o64[f].numpy(), what=f"ten triplets {sensors} out {f}")
    # out of place gives the same bits and leaves the input alone
    a, out = src.to(DEV), src.to(DEV)
    for e in els:
        if hasattr(e, "reset"):
            e.reset()
    assert scene.simulate(a, out=out) is out
    assert torch.equal(a.pos, src.pos.to(DEV)) and torch.equal(out.pos, r.pos) and torch.equal(out.dir, r.dir)
    # sensors on either side of the cut
    seen = [e for e in els if hasattr(e, "getHitsTensors")]
    assert len(seen) == len(o["sensor"])
    for e, (slot, mask, hl, w_before), (_, _, hl64, _) in zip(seen, o["sensor"], o64["sensor"]):
        locs, w, ids = e.getHitsTensors()
        assert locs.shape[0] == int(mask.sum())
        util.assert_parity(cpu(locs), hl[mask].numpy(), hl64[mask].numpy(), what=f"ten triplets sensor {slot} sensor locs")
        assert np.array_equal(cpu(w), w_before[mask].numpy())
        assert np.array_equal(cpu(ids), cpu(src.id)[mask.numpy()])
    # the distance matrix of all surfaces at the initial state, and the snapshots of a Paths wrapper
    tm = cpu(torch.cat([el.intersectTest(src.to(DEV)) for el in els], dim=1))
    ref = O.intersect_matrix(prog, p0, d0).numpy()
    assert util.pattern_mismatch(tm, ref) == 0
    util.assert_parity(tm.reshape(-1), ref.reshape(-1), O.intersect_matrix(prog, p0, d0, dtype=torch.float64).numpy().reshape(-1),
                       what="ten triplets distance matrix")
    for e in els:
        if hasattr(e, "reset"):
            e.reset()
    paths = ns.Paths(src.to(DEV))
    scene.simulate(paths)
    # (the kernel that keeps snapshots is another instantiation: the same formulas, not always the same last bit)
    assert float((paths.unwrap().pos - r.pos).abs().max()) <= 2e-6 * float(r.pos.abs().max())
    assert len(paths.get_history()) == 1 + 4 * 10 + len(sensors)    # the four faces of every triplet are hit, its three edges never


def _scattered_rays(ns, n, seed, depth=300.0):
    g = torch.Generator().manual_seed(seed)
    pos = torch.randn(n, 3, generator=g) * torch.tensor([3.0, 3.0, 0.0])
    pos[:, 2] = torch.rand(n, generator=g) * depth - 10.0
    d = torch.randn(n, 3, generator=g) * 0.15
    d[:, 2] = torch.where(torch.rand(n, generator=g) < 0.5, 1.0, -1.0)
    rays = ns.Rays.initialize(pos, d)
    rays.intensity = torch.where(torch.rand(n, generator=g) < 0.05, 0.0, 1.0)    # a few dead rays: a step leaves them alone
    return rays


@pytest.mark.parametrize("copies", [3, 5], ids=["96_surfaces", "160_surfaces"])
def test_ray_cast_and_step_over_ninety_six_surfaces(ns, copies):
    els = sum((scenes.naive32(ns, 100.0 * k) for k in range(copies)), [])
    scene = ns.Scene()
    for el in els:
        scene.add_element(el)
    scene._build_index_maps()
    assert scene.total_surfaces == 32 * copies
    prog = P.compile_elements(els)
    assert [p.shape[0] for p, _, _ in P.split_program(prog)] == {3: [64, 32], 5: [64, 64, 32]}[copies]
    src = _scattered_rays(ns, 30000, 1, depth=100.0 * copies)
    p0, d0, I0 = cpu(src.pos), cpu(src.dir), cpu(src.intensity)
    tmin, win, tm = O.ray_cast(prog, p0, d0)
    win = win.numpy()
    assert min(np.bincount(win[win >= 0] // 32, minlength=copies)) > 0.6 / copies * len(win)     # every copy of the scene wins rays
    rays = src.to(DEV)
    hit, el_ids, sf_ids = scene.ray_cast(rays)
    hit, el_ids, sf_ids = cpu(hit), cpu(el_ids), cpu(sf_ids)
    assert np.array_equal(hit, win >= 0)
    # identical winners, except where the two smallest distances are closer than float32 noise
    with np.errstate(invalid="ignore"):
        srt = np.sort(tm.numpy(), axis=1)
        tie = np.isfinite(srt[:, 1]) & (np.abs(srt[:, 1] - srt[:, 0]) <= 1e-4 * np.abs(srt[:, 0]))
    assert tie.sum() == 0      # (this bundle has no such ray: every winner is comparable)
    ok = hit & ~tie
    el_of, sf_of = cpu(scene.map_to_element), cpu(scene.map_to_surface)
    assert np.array_equal(el_ids[ok], el_of[win[ok]]) and np.array_equal(sf_ids[ok], sf_of[win[ok]])
    o = O.step_nearest(prog, p0, d0, I0)
    o64 = O.step_nearest(prog, p0, d0, I0, dtype=torch.float64)
    scene.rays = rays
    scene.step()
    r = scene.rays
    ok = ~tie
    assert np.array_equal(cpu(r.intensity)[ok] > 0, o["intensity"].numpy()[ok] > 0)
    for f, got in (("pos", r.pos), ("dir", r.dir), ("intensity", r.intensity)):
        util.assert_parity(cpu(got)[ok], o[f].numpy()[ok], o64[f].numpy()[ok], what=f"three naive32 step {f}")
    # the sensors (one per copy; the later ones lie in later pieces) saw what the oracle's did
    sensors = [e for e in els if hasattr(e, "getHitsTensors")]
    assert len(sensors) == copies == len(o["sensor"])
    for e, (slot, mask, hl, w_before) in zip(sensors, o["sensor"]):
        mask = mask.numpy() & ok
        if not mask.any():
            assert not e.hitLocs
            continue
        locs, w, _ = e.getHitsTensors()
        assert locs.shape[0] == int(mask.sum()) > 500
        assert rel_err(cpu(locs), hl.numpy()[mask]) < PARITY_TOL
        assert np.array_equal(cpu(w), w_before.numpy()[mask])
    # simulate(): the bounce loop over the long scene is a loop of such steps (scene/base.py:129-142), bit for bit, and
    # the sensors get one list entry per bounce in which they were hit.  (Later bounces are not held against the oracle:
    # at |z| of hundreds a float32 ray re-finds the surface it stands on at a distance of rounding noise, which the
    # reference's `t <= 1e-6` filter passes or not by the last bit.)
    def run(loop):
        for e in els:
            e.reset() if hasattr(e, "reset") else None
        scene.rays = src.to(DEV)
        loop()
        return scene.rays, [[(cpu(l), cpu(w)) for l, w in zip(e.hitLocs, e.hitIntensity)] for e in sensors]
    scene.Nbounces = 3
    r_sim, seen_sim = run(scene.simulate)
    r_steps, seen_steps = run(lambda: [scene.step() for _ in range(3)])
    assert torch.equal(r_sim.pos, r_steps.pos) and torch.equal(r_sim.dir, r_steps.dir) and torch.equal(r_sim.intensity, r_steps.intensity)
    assert not torch.equal(r_sim.pos, r.pos)      # (the later bounces moved rays)
    for a, b in zip(seen_sim, seen_steps):
        assert len(a) == len(b) >= 1
        for (la, wa), (lb, wb) in zip(a, b):
            assert np.array_equal(la, lb) and np.array_equal(wa, wb)


def test_a_step_of_a_long_scene_under_paths_adds_one_snapshot(ns):
    """Scene.step() with the rays wrapped in Paths (rays/ray.py:100-207): the bounce of a 96-surface scene moves the same
    rays as without the wrapper and leaves one snapshot."""
    els = sum((scenes.naive32(ns, 100.0 * k) for k in range(3)), [])
    src = _scattered_rays(ns, 5000, 2)
    results = []
    for wrap in (False, True):
        scene = ns.Scene()
        for el in els:
            el.reset() if hasattr(el, "reset") else None
            scene.add_element(el)
        scene._build_index_maps()
        scene.rays = ns.Paths(src.to(DEV)) if wrap else src.to(DEV)
        scene.step()
        results.append(scene.rays)
    plain, wrapped = results
    assert torch.equal(wrapped.unwrap().pos, plain.pos) and torch.equal(wrapped.unwrap().intensity, plain.intensity)
    hist = wrapped.get_history()
    assert len(hist) == 2 and torch.equal(hist[1], plain.pos.cpu()) and not torch.equal(hist[0], hist[1])


def _objective(ns, seed):
    def objective(els, rays):
        g = torch.Generator().manual_seed(seed)
        n = len(rays)
        wp, wd, ws = (torch.randn(n, 3, generator=g).to(DEV) for _ in range(3))
        r = ns.SequentialScene(els).simulate(ns.Rays.from_planes(rays._pos, rays._dir, rays.intensity, rays.id, rays.wavelength))
        obj = (r.pos * wp).sum() + (r.dir * wd).sum()
        for el in els:
            if hasattr(el, "getHitsTensors") and el.hitLocs:
                locs, w, _ = el.getHitsTensors()
                obj = obj + (locs * ws[: locs.shape[0]] * w[:, None]).sum()
        return obj
    return objective


def _grads(ns, count, sensors, mode, monkeypatch, calls):
    monkeypatch.setenv("RTT_BACKWARD", mode)
    els = scenes.triplet_train(ns, count, grads=True, sensors=sensors)
    rays = _beam(ns, 6001).to(DEV)
    rays._pos.requires_grad_(True)
    rays._dir.requires_grad_(True)
    lib = N.lib()
    for name in ("rtt_trace_seq_adjoint_f32", "rtt_trace_seq_tangent_f32"):
        def counted(*a, real=getattr(lib, name), name=name):
            calls.append(name)
            return real(*a)
        monkeypatch.setattr(lib, name, counted, raising=False)
    obj = _objective(ns, 5)(els, rays)
    obj.backward()
    out = {k: p.grad.detach().double().cpu().reshape(-1)
           for k, p in torch.nn.ModuleList(els).named_parameters() if p.requires_grad}
    assert all(v is not None for v in out.values())
    out["rays.pos"] = rays._pos.grad.detach().double().cpu().reshape(-1)
    out["rays.dir"] = rays._dir.grad.detach().double().cpu().reshape(-1)
    return float(obj.detach()), out


@pytest.mark.parametrize("count,sensors", [(4, (1, 3)), (10, (2, 9))], ids=["28_surfaces", "72_surfaces"])
def test_long_programs_are_differentiated_in_stretches(ns, count, sensors, monkeypatch):
    """Reverse mode sweeps a long program stretch by stretch (each stretch a launch whose LDS holds its states), with
    the ray adjoint of one stretch as the upstream adjoint of the one in front.  Held against forward mode, which cuts
    the program elsewhere (28 surfaces: not at all) and differentiates the same formulas independently."""
    calls_a, calls_t = [], []
    obj_a, ga = _grads(ns, count, sensors, "adjoint", monkeypatch, calls_a)
    n_stretch = len(AG._stretches(scenes.triplet_train(ns, count, grads=True, sensors=sensors),
                                  P.compile_elements(scenes.triplet_train(ns, count, sensors=sensors))))
    assert n_stretch >= count / 2 and set(calls_a) == {"rtt_trace_seq_adjoint_f32"} and len(calls_a) >= n_stretch
    obj_t, gt = _grads(ns, count, sensors, "tangent", monkeypatch, calls_t)
    assert set(calls_t) == {"rtt_trace_seq_tangent_f32"}
    assert abs(obj_a - obj_t) <= 1e-6 * abs(obj_t)
    assert set(ga) == set(gt) and len(ga) == 3 * count + 2
    big = max(float(v.abs().max()) for v in gt.values())
    for k in ga:
        assert float((ga[k] - gt[k]).abs().max()) <= 3e-4 * max(1.0, big), (k, ga[k], gt[k])
