"""Fuzzy filter (phys/filter.py:36-50 of the reference): a user function of the local hit point scales the
intensity.  The kernels record the hit points, the scene applies the function; goldens from the reference."""
import numpy as np
import pytest
import torch

import scenes
from util import PARITY_TOL, golden, rays_from_golden, rel_err

DEV = "cuda:0"


@pytest.mark.gpu
def test_fuzzy_in_a_sequential_trace(ns):
    g = golden("fuzzy")
    els = scenes.fuzzy_system(ns)
    with torch.no_grad():
        r = ns.SequentialScene(els).simulate(rays_from_golden(ns, g, device=DEV))
    assert rel_err(r.pos.cpu().numpy(), g["seq_pos"]) < PARITY_TOL
    assert rel_err(r.dir.cpu().numpy(), g["seq_dir"]) < PARITY_TOL
    assert np.abs(r.intensity.cpu().numpy() - g["seq_intensity"]).max() < 1e-5
    assert ((r.intensity.cpu().numpy() == 0) == (g["seq_intensity"] == 0)).all()
    for k, s in ((0, els[0]), (1, els[3])):
        locs, w, _ = s.getHitsTensors()
        assert locs.shape[0] == g[f"seq_s{k}_locs"].shape[0]
        assert rel_err(locs.cpu().numpy(), g[f"seq_s{k}_locs"]) < PARITY_TOL
        assert np.abs(w.cpu().numpy() - g[f"seq_s{k}_w"]).max() < 1e-5   # the second sensor sees apodised rays


@pytest.mark.gpu
def test_fuzzy_in_nearest_hit_steps_and_simulate(ns):
    """Five bounces (sensor, two lens faces, apodiser, sensor) of the system at 1/100 scale, where float32 rays do
    not re-hit the surface they stand on, as separate steps and as one simulate() (which runs one launch per bounce
    because the filter changes the intensities that decide whether a ray bounces on)."""
    g = golden("fuzzy")
    for whole in (False, True):
        els = scenes.fuzzy_system(ns, small=True)
        scene = ns.Scene()
        for el in els:
            scene.add_element(el)
        scene._build_index_maps()
        scene.rays = rays_from_golden(ns, g, prefix="small_", device=DEV)
        with torch.no_grad():
            if whole:
                scene.Nbounces = 5
                scene.simulate()
            else:
                for _ in range(5):
                    scene.step()
        assert rel_err(scene.rays.pos.cpu().numpy(), g["step5_pos"]) < PARITY_TOL
        assert np.abs(scene.rays.intensity.cpu().numpy() - g["step5_intensity"]).max() < 1e-5
        for k, s in ((0, els[0]), (1, els[3])):
            locs, w, _ = s.getHitsTensors()
            assert locs.shape[0] == g[f"step_s{k}_locs"].shape[0]
            assert rel_err(locs.cpu().numpy(), g[f"step_s{k}_locs"]) < PARITY_TOL
            assert np.abs(w.cpu().numpy() - g[f"step_s{k}_w"]).max() < 1e-5


@pytest.mark.gpu
def test_fuzzy_functor_and_refusals(ns):
    fz = ns.Fuzzy(scenes.soft_edge)
    hl = torch.randn(64, 3, device=DEV) * 2
    d = torch.nn.functional.normalize(torch.randn(64, 3, device=DEV), dim=1)
    nd, im = fz(hl, d, d)
    assert torch.equal(nd, d) and torch.allclose(im, scenes.soft_edge(hl))
    lens = scenes.singlet(ns, c1_grad=True)
    els = [lens, ns.ElementCustom(ns.Disk(6.0, transform=scenes.T(ns, 8.0)), fz, "cpu")]
    rays = scenes.disk_bundle(ns).sample(64).to(DEV)
    with pytest.raises(NotImplementedError):
        ns.SequentialScene(els).simulate(rays)
