"""GPU: randomly drawn systems (rotated and decentred elements, doublets, cylindrical lenses, stops,
mirrors, sensor) through the HIP trace, the nearest-hit step and the gradient kernel, against the oracle."""
import math
import os

import numpy as np
import pytest
import torch

import rtt_oracle as O
import scenes
from util import assert_parity

import raytracetorch_amd.program as P

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("gen", ["random_system", "random_system_extra"])
@pytest.mark.parametrize("seed", range(int(os.environ.get("RTT_FUZZ_FIRST", "0")), int(os.environ.get("RTT_FUZZ_SEEDS", "16"))))
def test_random_system_sequential_and_nearest(ns, seed, gen):
    els = getattr(scenes, gen)(ns, np.random.default_rng(seed))
    torch.manual_seed(seed)
    rays = scenes.disk_bundle(ns, radius=5.0, z=-15.0, rot=[0.03 * math.sin(seed), 0.02, 0.0]).sample(3001)
    p0, d0, I0 = rays.pos.clone(), rays.dir.clone(), rays.intensity.clone()
    prog = P.compile_elements(els)
    # hit/miss matrix
    tm = torch.cat([el.intersectTest(rays.to(DEV)) for el in els], dim=1).cpu().numpy()
    tm_o = O.intersect_matrix(prog, p0, d0).numpy()
    flips = np.isfinite(tm) != np.isfinite(tm_o)
    assert flips.sum() <= 3
    clean = ~flips.any(axis=1)
    assert_parity(tm[clean].reshape(-1), tm_o[clean].reshape(-1), O.intersect_matrix(prog, p0, d0, dtype=torch.float64).numpy()[clean].reshape(-1),
                  what=f"seed {seed} t matrix")
    # sequential trace
    with torch.no_grad():   # a Rectangle's half-widths are trainable by default, as in the reference
        out = ns.SequentialScene(els).simulate(rays.to(DEV))
    o32 = O.trace_sequential(prog, p0, d0, I0)
    o64 = O.trace_sequential(prog, p0, d0, I0, dtype=torch.float64)
    same = (o32["intensity"].numpy() > 0) == (out.intensity.cpu().numpy() > 0)
    assert (~same).sum() <= 3
    ok = same & clean & (o32["intensity"].numpy() > 0)
    assert_parity(out.pos.cpu().numpy()[ok], o32["pos"].numpy()[ok], o64["pos"].numpy()[ok], what=f"seed {seed} pos")
    assert_parity(out.dir.cpu().numpy()[ok], o32["dir"].numpy()[ok], o64["dir"].numpy()[ok], what=f"seed {seed} dir")
    # one nearest-hit bounce of the same system
    scene = ns.Scene()
    for el in els:
        if hasattr(el, "reset"):
            el.reset()
        scene.add_element(el)
    scene.rays = rays.to(DEV)
    with torch.no_grad():
        scene.step()
    twin_pos = None
    if any(p.requires_grad for p in scene.parameters()):
        # the differentiable bounce (dual-number visit) gives the values of the plain one
        twin = ns.Scene()
        for el in els:
            twin.add_element(el)
        twin.rays = rays.to(DEV)
        twin.step()
        assert twin.rays._pos.requires_grad
        twin_pos = twin.rays.pos.detach().cpu().numpy()
        for el in els:
            if hasattr(el, "reset"):
                el.reset()
    o = O.step_nearest(prog, p0, d0, I0)
    srt = np.sort(tm_o, axis=1)
    tie = np.isfinite(srt[:, 1]) & (np.abs(srt[:, 1] - srt[:, 0]) <= 1e-4 * np.abs(srt[:, 0]))
    ok = clean & ~tie
    # near-axial rays on a paraboloid cancel in -B - sqrt(disc): two float32 evaluations differ by 1e-4 there and
    # are judged against the float64 evaluation, like the sequential trace above
    o64s = O.step_nearest(prog, p0, d0, I0, dtype=torch.float64)
    assert_parity(scene.rays.pos.cpu().numpy()[ok], o["pos"].numpy()[ok], o64s["pos"].numpy()[ok], tol=2e-5,
                  what=f"seed {seed} step pos")
    assert np.array_equal((scene.rays.intensity.cpu().numpy() > 0)[ok], (o["intensity"].numpy() > 0)[ok])
    if twin_pos is not None:
        assert_parity(twin_pos[ok], o["pos"].numpy()[ok], o64s["pos"].numpy()[ok], tol=2e-5, what=f"seed {seed} differentiable step pos")
