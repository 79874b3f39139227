"""Differentiable trace of the 5-element system (9 surfaces, every parameter trainable): one reverse sweep over the whole
program (full states in LDS, one block per compute unit) against stretches of at most 6 surfaces (compact states, three
blocks per compute unit).  Forward + backward of a projection objective, event time."""
import os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import raytracetorch_amd as rta
import raytracetorch_amd.autograd as AG
import scenes
ns = scenes.namespace_from_package(rta)
dev = torch.device("cuda", 0)
n = int(os.environ.get("RAYS", 4 * 2 ** 20))
b = ns.CollimatedDisk(radius=4.0, ray_id=0, device=dev, transform=ns.RayTransformBundle(translation=[0.0, 0.0, -30.0]).to(dev))
src = b.sample(n, seed=1000)
src._pos
g = torch.Generator(device=dev).manual_seed(3)
wp, wd = (torch.randn(3, n, device=dev, generator=g) for _ in range(2))
for label, limit in (("one sweep", None), ("stretches of <= 6 surfaces", 6), ("stretches of <= 12 surfaces (no cut here)", 12)):
    els = scenes.system5(ns)
    scenes.make_all_trainable(els)
    scene = ns.SequentialScene(els).to(dev)
    real = AG.one_sweep
    if limit is not None:
        AG.SEGMENT_SURFACES = limit
        AG.one_sweep = lambda prog, n_el, real=real, limit=limit: prog.shape[0] <= limit and real(prog, n_el)
    ts, grads = [], None
    for i in range(12):
        for p in scene.parameters():
            p.grad = None
        els[-1].reset()
        rays = ns.Rays.from_planes(src._pos, src._dir, src.intensity, src.id, src.wavelength)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = scene.simulate(rays)
        obj = (out._pos * wp).sum() + (out._dir * wd).sum()
        obj.backward()
        e.record(); torch.cuda.synchronize()
        if i >= 3:
            ts.append(s.elapsed_time(e))
    grads = torch.cat([p.grad.reshape(-1).double().cpu() for p in scene.parameters() if p.grad is not None])
    AG.one_sweep, AG.SEGMENT_SURFACES = real, 12
    print(f"{label:45s}: {statistics.median(ts):8.3f} ms  |grad| {float(grads.norm()):.6e}", flush=True)
