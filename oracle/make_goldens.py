"""Generate ``tests/golden/*.npz`` by running the UNMODIFIED upstream reference on CPU.

Container-only tooling (the reference tree does not exist on the GPU box).  Run from the repo root:

    python oracle/make_goldens.py

Only inputs and outputs (arrays) are written; no reference source or bytecode is copied.  The
scene constructors live in ``tests/scenes.py`` and are shared with the tests, which replay the same
constructor calls against ``raytracetorch_amd``.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_loader  # noqa: E402
import scenes  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
INF = float("inf")


def npy(t):
    return t.detach().cpu().numpy()


def rays_arrays(rays, prefix):
    return {prefix + "pos": npy(rays.pos), prefix + "dir": npy(rays.dir),
            prefix + "intensity": npy(rays.intensity), prefix + "id": npy(rays.id)}


def clone_rays(ns, rays):
    return ns.Rays(pos=rays.pos.clone(), dir=rays.dir.clone(), intensity=rays.intensity.clone(),
                   id=rays.id.clone(), wavelength=rays.wavelength.clone(), batch_size=rays.batch_size)


def gen_sequential(ns, only=None):
    for name, (mk_elements, mk_rays, n) in scenes.sequential_cases(ns).items():
        if only and name not in only:
            continue
        torch.manual_seed(0)
        elements = mk_elements()
        rays0 = mk_rays(n)
        out = rays_arrays(rays0, "in_")
        with torch.no_grad():
            for k, el in enumerate(elements):
                out[f"it_{k}"] = npy(el.intersectTest(rays0))
            # Element.forward of the first element on the rays that hit each of its surfaces
            el = elements[0]
            tm = el.intersectTest(rays0)
            for j in range(tm.shape[1]):
                m = tm[:, j] < INF
                out[f"fw_{j}_mask"] = npy(m)
                if bool(m.any()):
                    p, d, im = el(rays0[m], j)
                    out[f"fw_{j}_pos"], out[f"fw_{j}_dir"], out[f"fw_{j}_imult"] = npy(p), npy(d), npy(im)
                    if hasattr(el, "reset"):
                        el.reset()
            # states after each element prefix
            for k in range(len(elements)):
                for e in elements:
                    if hasattr(e, "reset"):
                        e.reset()
                r = ns.SequentialScene(elements[:k + 1]).simulate(clone_rays(ns, rays0))
                out.update(rays_arrays(r, f"after_{k}_"))
            for e in elements:
                if hasattr(e, "reset"):
                    e.reset()
            r = ns.SequentialScene(elements).simulate(clone_rays(ns, rays0))
            out.update(rays_arrays(r, "out_"))
            for k, e in enumerate(elements):
                if hasattr(e, "getHitsTensors") and e.hitLocs:
                    locs, inten, ids = e.getHitsTensors()
                    out[f"sensor_{k}_locs"], out[f"sensor_{k}_w"], out[f"sensor_{k}_id"] = npy(locs), npy(inten), npy(ids)
        np.savez_compressed(os.path.join(OUT, f"seq_{name}.npz"), **out)
        print(f"seq_{name}: N={n} surfaces={sum(len(e.shape) for e in elements)} "
              f"alive={(out['out_intensity'] > 0).sum()}")


def gen_primitives(ns):
    cases = dict(scenes.primitive_cases(ns))
    cases.update(scenes.kat_rays(ns))
    for name, (mk_surf, mk_rays) in cases.items():
        torch.manual_seed(1)
        surf = mk_surf()
        rays = mk_rays()
        with torch.no_grad():
            t_test = surf.intersectTest(rays)
            t, hit, nrm, hl = surf(rays)
        out = rays_arrays(rays, "in_")
        out.update(t_test=npy(t_test), t=npy(t), hit_global=npy(hit), normal_global=npy(nrm), hit_local=npy(hl))
        np.savez_compressed(os.path.join(OUT, f"prim_{name}.npz"), **out)
        print(f"prim_{name}: hits={(out['t'] < INF).sum()}/{len(out['t'])}")


def naive_rays(ns, scene):
    import math
    scene.add_bundle(scenes.disk_bundle(ns, radius=6.5, z=-10.0, rot=[0.03, 0.0, 0.0]), 2048)
    scene.add_bundle(ns.PointSource(NA=0.9, ray_id=1, transform=ns.RayTransformBundle(
        translation=[0.0, 0.5, 44.5])), 2048)
    scene.add_bundle(ns.PointSource(NA=0.8, ray_id=2, transform=ns.RayTransformBundle(
        translation=[1.0, 0.0, 75.0], rotation=[math.pi, 0.0, 0.0])), 2048)
    scene.add_bundle(ns.PointSource(NA=0.99, ray_id=3, transform=ns.RayTransformBundle(
        translation=[0.0, -1.0, 62.0], rotation=[0.0, math.pi / 2, 0.0])), 2048)
    scene._build_rays()


def gen_naive(ns):
    torch.manual_seed(0)
    scene = ns.Scene()
    for el in scenes.naive32(ns):
        scene.add_element(el)
    scene._build_index_maps()
    naive_rays(ns, scene)
    rays0 = clone_rays(ns, scene.rays)
    out = rays_arrays(rays0, "in_")
    with torch.no_grad():
        out["t_matrix"] = npy(torch.cat([el.intersectTest(rays0) for el in scene.elements], dim=1))
        hit, el_ids, sf_ids = scene.ray_cast(rays0)
        out.update(hit_mask=npy(hit), el_ids=npy(el_ids), sf_ids=npy(sf_ids))
        scene.step()
        out.update(rays_arrays(scene.rays, "step1_"))
        sensor = scene.elements[-1]
        if sensor.hitLocs:
            locs, inten, ids = sensor.getHitsTensors()
            out.update(sensor_locs=npy(locs), sensor_w=npy(inten))
    np.savez_compressed(os.path.join(OUT, "naive32.npz"), **out)
    win = out["t_matrix"].argmin(1)[out["hit_mask"]]
    print(f"naive32: N={len(out['hit_mask'])} hit={out['hit_mask'].sum()} distinct winners={len(np.unique(win))}")


def gen_spot(ns, dtype, tag):
    torch.manual_seed(0)
    n = 65536
    lens, sensor = scenes.spot_system(ns, grads=True)
    bundle = ns.CollimatedDisk(radius=4.0, ray_id=0, dtype=dtype,
                               transform=ns.RayTransformBundle(translation=[0.0, 0.0, -30.0], dtype=dtype))
    if dtype == torch.float64:
        lens.double(); sensor.double(); bundle.double()
    rays = bundle.sample(n)
    out = {"first_pos": npy(rays.pos[:64]), "first_dir": npy(rays.dir[:64]),
           "checksum": np.array([rays.pos.double().sum().item(), rays.dir.double().sum().item()])}
    scene = ns.SequentialScene([lens, sensor])
    scene.simulate(rays)
    locs, w, _ = sensor.getHitsTensors()
    xy = locs[:, :2]
    act = w > 0
    xy, w = xy[act], w[act]
    ws = w.sum().clamp(min=1e-12)
    cx, cy = (xy[:, 0] * w).sum() / ws, (xy[:, 1] * w).sum() / ws
    loss = torch.sqrt(((xy[:, 0] - cx) ** 2 + (xy[:, 1] - cy) ** 2) * (w / ws)).sum()
    loss.backward()
    s0, s1 = lens.shape.surfaces[0], lens.shape.surfaces[1]
    out.update(loss=npy(loss), n_hits=np.array(int(act.sum())), centroid=np.array([cx.item(), cy.item()]),
               g_c1=npy(s0.c.grad), g_c2=npy(s1.c.grad), g_trans0=npy(s0.transform.trans.grad),
               g_trans1=npy(s1.transform.trans.grad), g_ior_glass=npy(lens.ior_glass.grad))
    np.savez_compressed(os.path.join(OUT, f"spot_{tag}.npz"), **out)
    print(f"spot_{tag}: loss={loss.item():.6f} hits={int(act.sum())} g_c1={s0.c.grad.item():.4f} "
          f"g_c2={s1.c.grad.item():.4f} g_t0={s0.transform.trans.grad.tolist()} "
          f"g_t1={s1.transform.trans.grad.tolist()} g_ior={lens.ior_glass.grad.item():.4f}")


def gen_grad_kats(ns):
    """tests/test_primitive.py:166-307 of the reference: d(sum hit)/d(translation)."""
    out = {}
    tr = torch.tensor([0.0, 0.0, 5.0], requires_grad=True)
    plane = ns.Plane(transform=ns.RayTransform(translation=tr, trans_grad=True, rot_grad=False))
    d = torch.tensor([[0.0, 1.0, 1.0]]); d = d / d.norm(dim=1, keepdim=True)
    rays = ns.Rays.initialize(torch.tensor([[0.0, 0.0, 0.0]]), d)
    _, hit, _, _ = plane(rays)
    hit.sum().backward()
    out["plane_hit"], out["plane_grad"] = npy(hit), npy(plane.transform.trans.grad)
    tr = torch.tensor([0.0, 0.0, 5.0], requires_grad=True)
    quad = ns.Quadric(transform=ns.RayTransform(translation=tr, trans_grad=True, rot_grad=False), c=0.01, k=0.0, c_grad=True)
    rays = ns.Rays.initialize(torch.tensor([[5.0, 0.0, 0.0]]), torch.tensor([[0.0, 0.0, 1.0]]))
    _, hit, _, _ = quad(rays)
    hit.sum().backward()
    out["quadric_hit"], out["quadric_grad"], out["quadric_grad_c"] = npy(hit), npy(quad.transform.trans.grad), npy(quad.c.grad)
    np.savez_compressed(os.path.join(OUT, "grad_kats.npz"), **out)
    print("grad_kats:", {k: v.tolist() for k, v in out.items()})


def gen_grad_tilt(ns):
    """Gradients of a posed singlet + sensor with every pose component trainable, by the reference's autograd in
    float32 and float64: once with the lens rotation exactly zero (tilt sensitivity at the nominal pose) and once
    tilted.  Objective: fixed random projections of the output rays and of the sensor hit positions."""
    torch.manual_seed(3)
    n = 2048
    rays = scenes.disk_bundle(ns, radius=3.5, rot=[0.02, 0.01, 0.0]).sample(n)
    wp, wd, ws = torch.randn(n, 3), torch.randn(n, 3), torch.randn(n, 3)
    out = rays_arrays(rays, "in_")
    out.update(wp=npy(wp), wd=npy(wd), ws=npy(ws))
    for tag, rot in (("zero", [0.0, 0.0, 0.0]), ("tilted", [0.03, -0.02, 0.1])):
        for dtype, dt in ((torch.float32, "f32"), (torch.float64, "f64")):
            lens, sensor = scenes.tilt_system(ns, rot)
            if dtype == torch.float64:
                lens.double(); sensor.double()
            r = ns.Rays.initialize(rays.pos.to(dtype), rays.dir.to(dtype), dtype=dtype)
            r = ns.SequentialScene([lens, sensor]).simulate(r)
            locs, w, _ = sensor.getHitsTensors()
            obj = (r.pos * wp.to(dtype)).sum() + (r.dir * wd.to(dtype)).sum() + (locs * ws.to(dtype) * w[:, None]).sum()
            obj.backward()
            got = scenes.tilt_gradients(lens, sensor)
            out.update({f"{tag}_{dt}_{k}": npy(v) for k, v in got.items()})
            out[f"{tag}_{dt}_obj"] = npy(obj)
            print("grad_tilt", tag, dt, float(obj), {k: np.round(npy(v), 4).tolist() for k, v in got.items()})
    np.savez_compressed(os.path.join(OUT, "grad_tilt.npz"), **out)


def gen_grad_naive(ns):
    """Gradients through Scene.simulate (nearest-hit bounce loop, scene/base.py:129-235) by the reference's autograd,
    float32 and float64: a folded system in which every ray bounces seven times and crosses the sensor twice.
    Objective: fixed random projections of the final rays and of all sensor hit positions (in the order of
    getHitsTensors: bounce by bounce, rays in order)."""
    torch.manual_seed(5)
    n = 1024
    rays = scenes.folded_bundle(ns).sample(n)
    wp, wd = torch.randn(n, 3), torch.randn(n, 3)
    out = rays_arrays(rays, "in_")
    for dtype, dt in ((torch.float32, "f32"), (torch.float64, "f64")):
        parts = scenes.folded_system(ns)
        scene = ns.Scene()
        for el in parts:
            if dtype == torch.float64:
                el.double()
            scene.add_element(el)
        scene._build_index_maps()
        scene.rays = ns.Rays.initialize(rays.pos.to(dtype), rays.dir.to(dtype), dtype=dtype)
        scene.Nbounces = 12
        scene.simulate()
        locs, w, _ = parts[2].getHitsTensors()
        if dt == "f32":
            torch.manual_seed(6)
            ws = torch.randn(locs.shape[0], 3)
            out.update(wp=npy(wp), wd=npy(wd), ws=npy(ws), n_hits=np.int64(locs.shape[0]), hit_locs=npy(locs))
        r = scene.rays
        obj = (r.pos * wp.to(dtype)).sum() + (r.dir * wd.to(dtype)).sum() + (locs * ws.to(dtype) * w[:, None]).sum()
        obj.backward()
        got = scenes.folded_gradients(*parts)
        out.update({f"{dt}_{k}": npy(v) for k, v in got.items()})
        out[f"{dt}_obj"] = npy(obj.detach())
        out.update(rays_arrays(r, f"{dt}_out_") if dt == "f32" else {})
        print("grad_naive", dt, float(obj.detach()), locs.shape, {k: np.round(npy(v), 3).tolist() for k, v in got.items()})
    np.savez_compressed(os.path.join(OUT, "grad_naive.npz"), **out)


def gen_grad_rays(ns):
    """Gradients that reach the rays' origin: a bundle whose pose is trainable feeds (a) the sequential tilt system
    and (b) the folded nearest-hit system; the reference's autograd gives d(objective)/d(bundle pose) next to the
    element gradients.  Float32 and float64."""
    out = {}
    n = 1024
    torch.manual_seed(8)
    wp, wd = torch.randn(n, 3), torch.randn(n, 3)
    out.update(wp=npy(wp), wd=npy(wd))
    for case in ("seq", "naive"):
        for dtype, dt in ((torch.float32, "f32"), (torch.float64, "f64")):
            bundle = scenes.trainable_bundle(ns, case)
            torch.manual_seed(9)
            rays = bundle.sample(n)
            if dtype == torch.float64:
                rays = ns.Rays.initialize(rays.pos.double(), rays.dir.double(), dtype=dtype)
            if case == "seq":
                parts = scenes.tilt_system(ns, [0.03, -0.02, 0.1])
                scene = ns.SequentialScene(list(parts))
            else:
                parts = scenes.folded_system(ns)
                scene = ns.Scene()
                for el in parts:
                    scene.add_element(el)
                scene._build_index_maps()
                scene.Nbounces = 12
            if dtype == torch.float64:
                for el in parts:
                    el.double()
            if case == "seq":
                r = scene.simulate(rays)
            else:
                scene.rays = rays
                scene.simulate()
                r = scene.rays
            locs, w, _ = parts[-1].getHitsTensors()
            if dt == "f32":
                torch.manual_seed(10)
                ws = torch.randn(locs.shape[0], 3)
                out[f"{case}_ws"] = npy(ws)
            obj = (r.pos * wp.to(dtype)).sum() + (r.dir * wd.to(dtype)).sum() + (locs * ws.to(dtype) * w[:, None]).sum()
            obj.backward()
            got = dict(bundle_trans=bundle.transform.trans.grad, bundle_rot=bundle.transform.rot_vec.grad,
                       c1=parts[0].shape.surfaces[0].c.grad, lens_trans=parts[0].shape.transform.trans.grad)
            out.update({f"{case}_{dt}_{k}": npy(v) for k, v in got.items()})
            out[f"{case}_{dt}_obj"] = npy(obj.detach())
            print("grad_rays", case, dt, float(obj.detach()), {k: np.round(npy(v), 3).tolist() for k, v in got.items()})
    np.savez_compressed(os.path.join(OUT, "grad_rays.npz"), **out)


def gen_paraxial(ns):
    """Paraxial matrices (elements/*.py getParaxial, scene/sequential.py:38-62), FocalLengthLoss with its gradients
    (optim/goals.py:16-39), and the log-barrier constraints (optim/constraints.py) on a system with every element
    kind that has a matrix."""
    els = scenes.paraxial_system(ns)
    out = {}
    for i, el in enumerate(els):
        zs, mats = el.getParaxial()
        out[f"z_{i}"] = np.array([float(z) for z in zs])
        out[f"mat_{i}"] = np.stack([npy(m) for m in mats])
    scene = ns.SequentialScene(els)
    out["system"] = npy(scene.getParaxial())
    loss = ns.FocalLengthLoss(75.0)(scene)
    loss.backward()
    lens = els[0]
    out.update(focal_loss=npy(loss), g_c1=npy(lens.shape.surfaces[0].c.grad), g_c2=npy(lens.shape.surfaces[1].c.grad),
               g_ior=npy(lens.ior_glass.grad), g_focal=npy(els[4].P.grad), g_mirror=npy(els[6].shape.c.grad),
               g_cyl=npy(els[1].shape.surfaces[0].c.grad))
    lenses = els[:4]
    cons = dict(thick=ns.ThicknessConstraint(lenses, 0.5, 5.0, weight=2.0), thick_lb=ns.ThicknessConstraint(els, 0.5),
                space=ns.SpacingConstraint(els, 1.0, weight=0.5), length=ns.SystemLengthConstraint(els, 100.0),
                none=ns.ThicknessConstraint(els[4:], 0.5))
    for p in scene.parameters():
        p.grad = None
    total = sum(c() for c in cons.values())
    total.backward()
    out.update({"con_" + k: npy(c()) for k, c in cons.items()})
    out.update(con_g_t=npy(lens.shape.surfaces[1].transform.trans.grad), con_g_trans=npy(lens.shape.transform.trans.grad))
    np.savez_compressed(os.path.join(OUT, "paraxial.npz"), **out)
    print("paraxial", out["system"][1, 0], float(loss), {k: float(out["con_" + k]) for k in cons})


def gen_spot_target(ns):
    """SpotTargetLoss (optim/goals.py:42-96) on the folded nearest-hit system: two bundles, two targets; value and
    gradients by the reference's autograd, float32."""
    parts = scenes.folded_system(ns)
    scene = ns.Scene()
    for el in parts:
        scene.add_element(el)
    scene._build_index_maps()
    scene.Nbounces = 12
    bundles = [scenes.folded_bundle(ns), scenes.trainable_bundle(ns, "naive")]
    torch.manual_seed(12)
    loss = ns.SpotTargetLoss(parts[2], torch.tensor([[0.001, -0.002], [0.0, 0.003]]))(scene, bundles, N_rays=512)
    loss.backward()
    out = {k: npy(v) for k, v in scenes.folded_gradients(*parts).items()}
    out.update(loss=npy(loss.detach()), bundle_trans=npy(bundles[1].transform.trans.grad),
               bundle_rot=npy(bundles[1].transform.rot_vec.grad))
    np.savez_compressed(os.path.join(OUT, "spot_target.npz"), **out)
    print("spot_target", float(loss), {k: np.round(v, 4).tolist() for k, v in out.items()})


def gen_fuzzy(ns):
    """Fuzzy filter (phys/filter.py:36-50) in a sequential trace and in three nearest-hit steps: final rays and what
    the two sensors recorded."""
    torch.manual_seed(14)
    rays = scenes.disk_bundle(ns, radius=4.5, rot=[0.01, 0.02, 0.0]).sample(2048)
    out = rays_arrays(rays, "in_")
    with torch.no_grad():
        els = scenes.fuzzy_system(ns)
        r = ns.SequentialScene(els).simulate(clone_rays(ns, rays))
        out.update(rays_arrays(r, "seq_"))
        for k, s in ((0, els[0]), (1, els[3])):
            locs, w, _ = s.getHitsTensors()
            out[f"seq_s{k}_locs"], out[f"seq_s{k}_w"] = npy(locs), npy(w)
        els = scenes.fuzzy_system(ns, small=True)
        scene = ns.Scene()
        for el in els:
            scene.add_element(el)
        scene._build_index_maps()
        torch.manual_seed(15)
        small = ns.CollimatedDisk(radius=0.045, ray_id=0, transform=ns.RayTransformBundle(
            translation=[0.0, 0.0, -0.3], rotation=[0.01, 0.02, 0.0])).sample(2048)
        out.update(rays_arrays(small, "small_"))
        scene.rays = clone_rays(ns, small)
        for _ in range(5):   # sensor, lens front, lens back, apodiser, sensor
            scene.step()
        out.update(rays_arrays(scene.rays, "step5_"))
        for k, s in ((0, els[0]), (1, els[3])):
            locs, w, _ = s.getHitsTensors()
            out[f"step_s{k}_locs"], out[f"step_s{k}_w"] = npy(locs), npy(w)
    np.savez_compressed(os.path.join(OUT, "fuzzy.npz"), **out)
    print("fuzzy", out["seq_intensity"].min(), out["seq_intensity"].max(), (out["seq_intensity"] == 0).sum(),
          out["seq_s1_w"].shape, out["step5_intensity"].mean(), out["step_s1_w"].shape, (out["step5_intensity"] == 0).sum())


def gen_manual_spot(ns):
    """tests/test_optimize_singlet.py:66-116 of the reference: loss and gradients of its closure, float32 / float64."""
    torch.manual_seed(16)
    rays = ns.CollimatedDisk(5.0, 1, transform=ns.RayTransformBundle(translation=[0, 0, -10])).sample(4096)
    out = rays_arrays(rays, "in_")
    for dtype, dt in ((torch.float32, "f32"), (torch.float64, "f64")):
        lens = scenes.manual_lens(ns)
        if dtype == torch.float64:
            lens.double()
        r = ns.Rays.initialize(rays.pos.to(dtype), rays.dir.to(dtype), dtype=dtype)
        loss = scenes.manual_spot_loss(ns, lens, r)
        loss.backward()
        s0, s1 = lens.shape.surfaces[0], lens.shape.surfaces[1]
        got = dict(loss=loss.detach(), c1=s0.c.grad, c2=s1.c.grad, t0=s0.transform.trans.grad, t1=s1.transform.trans.grad,
                   ior=lens.ior_glass.grad)
        out.update({f"{dt}_{k}": npy(v) for k, v in got.items()})
        print("manual_spot", dt, {k: np.round(npy(v), 5).tolist() for k, v in got.items()})
    np.savez_compressed(os.path.join(OUT, "manual_spot.npz"), **out)


def gen_grad_plano(ns):
    """Gradients w.r.t. the curvature of a flat lens face (c = 0, where the reference takes the linear root -C/B of
    geom/primitives.py:300-318 and autograd differentiates that formula) and of the curved back face; tilted bundle,
    random projection of the output rays.  Float32 and float64."""
    torch.manual_seed(20)
    n = 2048
    rays = scenes.disk_bundle(ns, radius=3.5, rot=[0.05, -0.03, 0.0]).sample(n)
    wp, wd = torch.randn(n, 3), torch.randn(n, 3)
    out = rays_arrays(rays, "in_")
    out.update(wp=npy(wp), wd=npy(wd))
    for dtype, dt in ((torch.float32, "f32"), (torch.float64, "f64")):
        lens = scenes.singlet(ns, c1=0.0, c2=-0.08, c1_grad=True, c2_grad=True, t_grad=True, ior_glass_grad=True)
        if dtype == torch.float64:
            lens.double()
        r = ns.Rays.initialize(rays.pos.to(dtype), rays.dir.to(dtype), dtype=dtype)
        r = ns.SequentialScene([lens]).simulate(r)
        obj = (r.pos * wp.to(dtype)).sum() + (r.dir * wd.to(dtype)).sum()
        obj.backward()
        s0, s1 = lens.shape.surfaces[0], lens.shape.surfaces[1]
        got = dict(obj=obj.detach(), c1=s0.c.grad, c2=s1.c.grad, t0=s0.transform.trans.grad, ior=lens.ior_glass.grad)
        out.update({f"{dt}_{k}": npy(v) for k, v in got.items()})
        print("grad_plano", dt, {k: np.round(npy(v), 4).tolist() for k, v in got.items()})
    np.savez_compressed(os.path.join(OUT, "grad_plano.npz"), **out)


def gen_grad_fuzz(ns, seeds=range(10), file_name="grad_fuzz", n=512):
    """Gradients of random systems with every parameter trainable, by the reference's autograd (float32 and
    float64): the element kinds of both random generators, poses with and without rotation.  `grad_fuzz` holds seeds
    0-9 with 512 rays, `grad_fuzz_wide` seeds 10-39 with 256 rays."""
    import math
    out = {}
    for gen in ("random_system", "random_system_extra"):
        for seed in seeds:
            torch.manual_seed(100 + seed)
            rays = scenes.disk_bundle(ns, radius=5.0, z=-15.0, rot=[0.03 * math.sin(seed), 0.02, 0.0]).sample(n)
            wp, wd, ws = torch.randn(n, 3), torch.randn(n, 3), torch.randn(n, 3)
            tag = f"{gen}/{seed}"
            out.update({f"{tag}/in_pos": npy(rays.pos), f"{tag}/in_dir": npy(rays.dir), f"{tag}/wp": npy(wp),
                        f"{tag}/wd": npy(wd), f"{tag}/ws": npy(ws)})
            for dtype, dt in ((torch.float32, "f32"), (torch.float64, "f64")):
                els = getattr(scenes, gen)(ns, np.random.default_rng(seed))
                if dtype == torch.float64:
                    for el in els:
                        el.double()
                params = scenes.make_all_trainable(els)
                r = ns.Rays.initialize(rays.pos.to(dtype), rays.dir.to(dtype), dtype=dtype)
                obj = scenes.fuzz_objective(ns, els, r, wp.to(dtype), wd.to(dtype), ws.to(dtype))
                obj.backward()
                out[f"{tag}/{dt}/obj"] = npy(obj.detach())
                for name, p in params.items():
                    out[f"{tag}/{dt}/g/{name}"] = npy(torch.zeros_like(p) if p.grad is None else p.grad)
                # the same system, one nearest-hit bounce
                els = getattr(scenes, gen)(ns, np.random.default_rng(seed))
                if dtype == torch.float64:
                    for el in els:
                        el.double()
                params = scenes.make_all_trainable(els)
                r = ns.Rays.initialize(rays.pos.to(dtype), rays.dir.to(dtype), dtype=dtype)
                obj = scenes.fuzz_step_objective(ns, els, r, wp.to(dtype), wd.to(dtype), ws.to(dtype))
                obj.backward()
                out[f"{tag}/{dt}/step_obj"] = npy(obj.detach())
                for name, p in params.items():
                    out[f"{tag}/{dt}/s/{name}"] = npy(torch.zeros_like(p) if p.grad is None else p.grad)
            print("grad_fuzz", tag, float(out[f"{tag}/f32/obj"]), float(out[f"{tag}/f32/step_obj"]), len(params))
    np.savez_compressed(os.path.join(OUT, file_name + ".npz"), **out)


def gen_bundles(ns):
    """Bundle.sample of the reference, together with the torch.rand values it consumed, in the order the code
    samples (positions first, then directions; one torch.rand of N values per draw, whatever its shape).  The
    Gaussian beam consumes torch.normal values instead: its golden keeps the standard normals z = x / sigma."""
    n = 4096
    for name, case in scenes.bundle_cases(ns).items():
        mk, kind, par_of, draws = case[:4]
        panel = case[4] if len(case) > 4 else {}
        b = mk()
        torch.manual_seed(0)
        rays = b.sample(n)
        torch.manual_seed(0)
        us = [torch.rand((n, 1)).reshape(-1) for _ in range(draws)] + [torch.zeros(n)] * (4 - draws)
        out = rays_arrays(rays, "")
        if name == "gauss":
            torch.manual_seed(0)
            local = b.sample_pos(n)
            sx, sy = par_of(b)
            us[0], us[1] = local[:, 0] / sx, local[:, 1] / sy   # standard normals, not uniforms
        out.update(kind=np.int32(kind), par=np.asarray(par_of(b), dtype=np.float64), ua=npy(us[0]), ub=npy(us[1]),
                   uc=npy(us[2]), ud=npy(us[3]), R=npy(b.transform.rot), t=npy(b.transform.trans),
                   wavelength=npy(rays.wavelength), ray_id=np.int32(b.ray_id),
                   area=np.int32(panel.get("area", 0)), emit=np.int32(panel.get("emit", 0)),
                   epar=np.asarray(panel.get("epar", ()), dtype=np.float64))
        np.savez_compressed(os.path.join(OUT, f"bundle_{name}.npz"), **out)
        print("bundle", name, tuple(rays.pos.shape))


def gen_paths(ns):
    """Paths history (rays/ray.py:100-207) of the first bounce of a small nearest-hit scene.  (The reference's
    SequentialScene cannot take a Paths wrapper: Shape.intersectTest calls rays.with_coords, which Paths lacks.)"""
    torch.manual_seed(0)
    rays = scenes.disk_bundle(ns, radius=6.0).sample(1024)
    out = rays_arrays(rays, "in_")
    scene = ns.Scene()
    for el in [scenes.singlet(ns), ns.Sensor(ns.Disk(20.0, transform=scenes.T(ns, 19.0)))]:
        scene.add_element(el)
    scene._build_index_maps()
    scene.rays = ns.Paths(clone_rays(ns, rays))
    with torch.no_grad():
        scene.step()
    out["step_history"] = np.stack([npy(h) for h in scene.rays.get_history()])
    out.update(rays_arrays(scene.rays.unwrap(), "step1_"))
    np.savez_compressed(os.path.join(OUT, "paths.npz"), **out)
    print("paths", out["step_history"].shape)


def gen_render(ns, only=None):
    """Renderer.render_3d images (render/camera.py:191-257) with the camera rays that produced them."""
    for name in scenes.render_cases(ns):
        if only and name not in only:
            continue
        scene, cam, rend = scenes.render_setup(ns, name)
        rays = cam.generate_rays()
        with torch.no_grad():
            img = rend.render_3d(cam)
            t_matrix = torch.cat([el.intersectTest(rays) for el in scene.elements], dim=1)
        np.savez_compressed(os.path.join(OUT, f"render_{name}.npz"), image=npy(img), t_matrix=npy(t_matrix),
                            **rays_arrays(rays, "in_"))
        print("render", name, tuple(img.shape), float((img != 1.0).any(-1).float().mean()))


def gen_sensor_groups(ns):
    """Sensor.getSpotSizeParallel_xy (elements/sensor.py:87-176) on hits of two bundles with different ids."""
    torch.manual_seed(0)
    lens, sensor = scenes.spot_system(ns, grads=False)
    scene = ns.SequentialScene([lens, sensor])
    a = scenes.disk_bundle(ns, ray_id=0).sample(3000)
    b = scenes.disk_bundle(ns, radius=2.0, x=1.0, y=-0.5, rot=[0.01, 0.02, 0.0], ray_id=3).sample(2000)
    out = {}
    out.update(rays_arrays(a, "a_")); out.update(rays_arrays(b, "b_"))
    with torch.no_grad():
        scene.simulate(a); scene.simulate(b)
        tgt = torch.tensor([[0.1, -0.2], [0.0, 0.0], [0.3, 0.3]])
        for tag, kw in (("centroid_p2", {}), ("centroid_p3", dict(norm_ord=3)), ("target_p2", dict(target_xy=tgt))):
            spot, wsum = sensor.getSpotSizeParallel_xy([3, 0, 7], **kw)
            out[f"{tag}_spot"], out[f"{tag}_wsum"] = npy(spot), npy(wsum)
    out["target"] = npy(tgt)
    np.savez_compressed(os.path.join(OUT, "sensor_groups.npz"), **out)
    print("sensor_groups", out["centroid_p2_spot"], out["centroid_p2_wsum"])


def gen_project(ref):
    """load_scene (project.py:96-179) of tests/golden/system5.rtt by the reference: what it builds (class names,
    parameter values), the rays it samples under seed 0, and the first bounce."""
    torch.manual_seed(0)
    scene = ref.project.load_scene(os.path.join(OUT, "system5.rtt"), device="cpu")
    out = rays_arrays(scene.rays, "in_")
    out["classes"] = np.array([type(el).__name__ for el in scene.elements] + [type(b).__name__ for b in scene.bundles])
    out["names"] = np.array(scene.element_names + scene.bundle_names)
    for k, v in scene.state_dict().items():
        if k.startswith(("elements.", "bundles.")) and v.dtype.is_floating_point:
            out["param/" + k] = npy(v)
    out["nbounces"] = np.int32(scene.Nbounces)
    with torch.no_grad():
        scene.step()
    out.update(rays_arrays(scene.rays, "step1_"))
    np.savez_compressed(os.path.join(OUT, "project_system5.npz"), **out)
    print("project", list(out["classes"]), len(out["in_pos"]))


def gen_physics(ns):
    """Functors called directly.  RefractFresnel consumes torch.rand_like(R) ([N,1]) once per call; the same
    values are recorded by re-seeding."""
    n = 4096
    for name, (mk, inputs) in scenes.physics_cases(ns).items():
        torch.manual_seed(1)
        hl, d, nrm = inputs(n)
        f = mk()
        torch.manual_seed(0)
        with torch.no_grad():
            nd, im = f(hl, d, nrm)
        torch.manual_seed(0)
        u = torch.rand((n, 1))
        np.savez_compressed(os.path.join(OUT, f"phys_{name}.npz"), hit_local=npy(hl), dir=npy(d), normal=npy(nrm),
                            new_dir=npy(nd), imult=npy(im), u=npy(u))
        print("phys", name)


def gen_profiles(ns):
    """The profile scans of tests/test_spherics.py and tests/test_cylindrics.py: Shape.intersectTest of bare lens
    shapes over a row of parallel rays along x and along y, plus Shape.forward of the first two surfaces."""
    out = {}
    for name, shape, extent in scenes.profile_shapes(ns):
        for axis in (0, 1):
            rays = scenes.profile_rays(ns, extent, axis)
            with torch.no_grad():
                out[f"{name}_t{axis}"] = npy(shape.intersectTest(rays))
                for j in (0, 1):
                    t, hg, ng, hl = shape.forward(rays, j)
                    out[f"{name}_fw{axis}_{j}_t"] = npy(t)
                    out[f"{name}_fw{axis}_{j}_hit"] = npy(hg)
                    out[f"{name}_fw{axis}_{j}_n"] = npy(ng)
                    out[f"{name}_fw{axis}_{j}_local"] = npy(hl)
        print("profile", name, out[f"{name}_t0"].shape, int(np.isfinite(out[f"{name}_t0"]).sum()), int(np.isfinite(out[f"{name}_t1"]).sum()))
    np.savez_compressed(os.path.join(OUT, "shape_profiles.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = ref_loader.load()
    ns = scenes.namespace_from_package(ref)
    if sys.argv[1:] == ["profiles"]:
        return gen_profiles(ns)
    if sys.argv[1:] == ["bundles"]:
        return gen_bundles(ns)
    if sys.argv[1:] == ["naive_grad"]:
        return gen_grad_naive(ns)
    if sys.argv[1:] == ["ray_grad"]:
        return gen_grad_rays(ns)
    if sys.argv[1:] == ["paraxial"]:
        return gen_paraxial(ns)
    if sys.argv[1:] == ["spot_target"]:
        return gen_spot_target(ns)
    if sys.argv[1:] == ["manual_spot"]:
        return gen_manual_spot(ns)
    if sys.argv[1:] == ["grad_fuzz"]:
        return gen_grad_fuzz(ns)
    if sys.argv[1:] == ["grad_fuzz_wide"]:
        return gen_grad_fuzz(ns, seeds=range(10, 40), file_name="grad_fuzz_wide", n=256)
    if sys.argv[1:] == ["plano"]:
        return gen_grad_plano(ns)
    if sys.argv[1:] == ["fuzzy"]:
        return gen_fuzzy(ns)
    if sys.argv[1:] == ["tilt"]:
        return gen_grad_tilt(ns)
    if sys.argv[1:] == ["physics"]:
        return gen_physics(ns)
    if sys.argv[1:] == ["paths"]:
        return gen_paths(ns)
    if sys.argv[1:2] == ["render"]:
        return gen_render(ns, only=sys.argv[2:])
    if sys.argv[1:] == ["sensor"]:
        return gen_sensor_groups(ns)
    if sys.argv[1:] == ["project"]:
        return gen_project(ref)
    if sys.argv[1:2] == ["seq"]:
        return gen_sequential(ns, only=sys.argv[2:])
    gen_primitives(ns)
    gen_sequential(ns)
    gen_naive(ns)
    gen_spot(ns, torch.float32, "f32")
    gen_spot(ns, torch.float64, "f64")
    gen_grad_kats(ns)
    gen_grad_tilt(ns)
    gen_grad_naive(ns)
    gen_grad_rays(ns)
    gen_paraxial(ns)
    gen_spot_target(ns)
    gen_fuzzy(ns)
    gen_manual_spot(ns)
    gen_grad_plano(ns)
    gen_grad_fuzz(ns)
    gen_grad_fuzz(ns, seeds=range(10, 40), file_name="grad_fuzz_wide", n=256)
    gen_bundles(ns)
    gen_physics(ns)
    gen_paths(ns)
    gen_render(ns)
    gen_sensor_groups(ns)
    gen_project(ref)
    gen_profiles(ns)


if __name__ == "__main__":
    main()
