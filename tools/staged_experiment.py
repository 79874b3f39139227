"""Estimate for a staged bounce loop: run the fused loop a few bounces at a time and carry only the rays that still move
into the next stage (torch gathers stand in for a device-side compaction).  Times per schedule against the single launch."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import raytracetorch_amd as rta
from raytracetorch_amd import kernels as K
import scenes
ns = scenes.namespace_from_package(rta)
dev = "cuda:0"
n = 4 << 20
scene = ns.Scene()
for el in scenes.naive32(ns)[:-1]:      # without the sensor: the hit list is not what is being measured
    scene.add_element(el)
b = ns.CollimatedDisk(radius=4.0, ray_id=0, device=dev, transform=ns.RayTransformBundle(translation=[0.0, 0.0, -10.0]).to(dev))
src = b.sample(n, seed=1000, offset=0)
src._pos
prog = scene._program()

def single():
    r = src.clone()
    K.simulate_nearest(r, prog, 100)
    return r

kernel_ms = [0.0]


def staged(schedule, by_z=False):
    cur = src.clone()
    sizes = []
    kernel_ms[0] = 0.0
    evs = []
    for B in schedule:
        if len(cur) == 0:
            break
        sizes.append(len(cur))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        bounces, _ = K.simulate_nearest(cur, prog, B, want_bounces=True)
        e1.record()
        evs.append((e0, e1))
        idx = ((bounces == B) & (cur.intensity > 0)).nonzero().flatten()
        if by_z:   # rays on the same element next to one another (the elements sit along z)
            idx = idx[torch.argsort(cur._pos[2, idx])]
        cur = ns.Rays.from_planes(cur._pos[:, idx].contiguous(), cur._dir[:, idx].contiguous(), cur.intensity[idx].contiguous(),
                                  cur.id[idx].contiguous(), cur.wavelength[idx].contiguous())
    torch.cuda.synchronize()
    kernel_ms[0] = sum(a.elapsed_time(b) for a, b in evs)
    return sizes

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out

t, _ = timed(single)
print(f"single launch, 100 bounces: {t:.2f} ms", flush=True)
for sched in ([100], [5, 3, 3, 4, 5, 10, 20, 50], [2, 2, 2, 2, 3, 4, 5, 10, 20, 50], [1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 3, 5, 10, 20, 50]):
    for by_z in (False, True):
        t, sizes = timed(lambda: staged(sched, by_z))
        print(f"schedule {sched} sorted by z {by_z}: {t:.2f} ms, in the bounce kernels {kernel_ms[0]:.2f} ms   rays per stage {sizes}", flush=True)
