"""``.rtt`` project files (project.py / config.py of the reference): the fixture tests/golden/system5.rtt builds
the same objects, samples the same rays and makes the same first bounce here as in the unmodified reference."""
import json
import os

import numpy as np
import pytest
import torch

import raytracetorch_amd as rta
from raytracetorch_amd import config as C
from util import GOLDEN, PARITY_TOL, golden, rays_from_golden, rel_err

RTT = os.path.join(GOLDEN, "system5.rtt")


def test_load_scene_builds_what_the_reference_builds():
    g = golden("project_system5")
    torch.manual_seed(0)
    scene = rta.load_scene(RTT, device="cpu")
    got = [type(el).__name__ for el in scene.elements] + [type(b).__name__ for b in scene.bundles]
    assert got == [str(c) for c in g["classes"]]
    assert scene.element_names + scene.bundle_names == [str(n) for n in g["names"]]
    assert scene.Nbounces == int(g["nbounces"]) == 3
    assert scene.find_element("stop") is scene.elements[1] and scene.find_bundle("point") is scene.bundles[1]
    with pytest.raises(KeyError):
        scene.find_element("nope")
    state = scene.state_dict()
    checked = 0
    for key in g:
        if not key.startswith("param/"):
            continue
        name = key[len("param/"):]
        if name in state:     # same attribute path as the reference's module tree
            assert np.allclose(state[name].numpy(), g[key], rtol=0, atol=1e-7), name
            checked += 1
    assert checked >= 20
    # rays sampled under the same torch seed (CPU bundles draw torch.rand in the reference's order)
    assert rel_err(scene.rays.pos.numpy(), g["in_pos"]) < 2e-6 and rel_err(scene.rays.dir.numpy(), g["in_dir"]) < 2e-6
    assert np.array_equal(scene.rays.id.numpy(), g["in_id"])


def test_project_round_trip_and_validation(tmp_path):
    data = rta.load_project(RTT)
    out = tmp_path / "copy.rtt"
    rta.save_project(str(out), data["elements"], data["bundles"], data["settings"])
    assert rta.load_project(str(out)) == data
    bad = tmp_path / "bad.rtt"
    bad.write_text(json.dumps({"elements": []}))
    with pytest.raises(ValueError, match="Not a valid"):
        rta.load_project(str(bad))
    bad.write_text(json.dumps({"version": "9.9"}))
    with pytest.raises(ValueError, match="Unsupported"):
        rta.load_project(str(bad))
    with pytest.raises(ValueError, match="Cannot find class"):
        C.instantiate_from_config({"class": "NoSuchThing", "params": {}})


def test_type_introspection_helpers():
    from raytracetorch_amd.geom import RayTransform, Shape, Surface
    ctor = C.get_constructor_params(RayTransform)
    assert C.analyze_type(ctor["translation"][0]) == "VEC3" and C.analyze_type(ctor["trans_mask"][0]) == "BOOL3"
    assert C.analyze_type(ctor["trans_grad"][0]) == "BOOL" and C.analyze_type(ctor["dtype"][0]) == "DTYPE"
    lens = C.get_constructor_params(rta.elements.SingletLens)
    assert C.analyze_type(lens["transform"][0]) == "CLASS" and C.get_actual_class(lens["transform"][0]) is RayTransform
    assert C.analyze_type(lens["c1"][0]) == "PRIMITIVE"
    import typing
    assert C.get_all_actual_classes(typing.Union[Shape, Surface]) == [Shape, Surface]
    assert rta.elements.SingletLens in C.get_subclasses(rta.elements.Element)


@pytest.mark.gpu
def test_loaded_scene_steps_like_the_reference(ns):
    g = golden("project_system5")
    scene = rta.load_scene(RTT, device="cuda:0", sample_rays=False)
    assert all(p.is_cuda for p in scene.parameters())
    scene.rays = rays_from_golden(ns, g, device="cuda:0")
    scene.step()
    assert rel_err(scene.rays.pos.cpu().numpy(), g["step1_pos"]) < PARITY_TOL
    assert rel_err(scene.rays.dir.cpu().numpy(), g["step1_dir"]) < PARITY_TOL
    assert np.array_equal(scene.rays.intensity.cpu().numpy() > 0, g["step1_intensity"] > 0)
    # default device + sampling on the device
    torch.manual_seed(1)
    scene2 = rta.load_scene(RTT)
    assert scene2.rays._pos.is_cuda and len(scene2.rays) == 2560
    scene2.simulate()


def test_project_with_the_other_source_families(tmp_path, ns):
    """A project file naming the beam, panel and particle sources (rays/beam.py, rays/panels.py, rays/particle.py of
    the reference) builds them, nested emission model included (CPU: only construction and the torch sampler)."""
    import raytracetorch_amd as rtt
    tf = {"class": "RayTransformBundle", "params": {"translation": [0.0, 0.0, -30.0]}}
    bundles = [
        {"name": "beam", "class": "GaussianBeam", "params": {"diameter_1e2_x": 2.0, "diameter_1e2_y": 1.0, "ray_id": 1, "transform": tf}},
        {"name": "panel", "class": "RectangularPanel",
         "params": {"width": 3.0, "height": 1.0, "ray_id": 2, "transform": tf,
                    "emission": {"class": "SolidAngleEmission", "params": {"cone_angle": 0.3}}}},
        {"name": "ring", "class": "RingSource", "params": {"radius_inner": 1.0, "radius_outer": 2.0, "ray_id": 3, "transform": tf}},
        {"name": "ball", "class": "LambertianSphere", "params": {"radius": 0.5, "ray_id": 4, "transform": tf}},
        {"name": "dust", "class": "RayleighScatter", "params": {"ray_id": 5, "transform": tf}},
    ]
    elements = [{"name": "lens", "class": "SingletLens",
                 "params": {"c1": 0.05, "c2": -0.05, "d": 10.0, "t": 3.0, "ior_glass": 1.5}}]
    path = str(tmp_path / "sources.rtt")
    rtt.save_project(path, [{"config": e} for e in elements], [{"config": b, "N_rays": 64} for b in bundles], {"Nbounces": 3})
    scene = rtt.load_scene(path, device="cpu")
    assert [type(b).__name__ for b in scene.bundles] == [b["class"] for b in bundles]
    panel = scene.bundles[1]
    assert type(panel.emission).__name__ == "SolidAngleEmission" and abs(panel.emission.cone_angle - 0.3) < 1e-12
    assert panel._source()[0] == 7 and panel._source()[3] == 1      # device description: panel source, cone emission
    assert scene.rays is not None and len(scene.rays) == 5 * 64
    assert sorted(set(scene.rays.id.tolist())) == [1, 2, 3, 4, 5]
    assert scene.Nbounces == 3
