"""Host cost of one Scene.step() of the reference's benchmark scene (singlet + stop + sensor) at small sizes, and of
the 32-surface scene: wall time per call over many calls, no synchronisation in between."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import raytracetorch_amd as rta
import scenes
ns = scenes.namespace_from_package(rta)
dev = "cuda:0"
T = lambda z: ns.RayTransform(translation=[0.0, 0.0, z])
def small():
    sc = ns.Scene()
    for el in (scenes.singlet(ns, transform=T(0.0)), ns.CircularAperture(3.5, transform=T(5.0)), ns.Sensor(ns.Disk(20.0, transform=T(25.0)))):
        sc.add_element(el)
    return sc
def big():
    sc = ns.Scene()
    for el in scenes.naive32(ns):
        sc.add_element(el)
    return sc
b = ns.CollimatedDisk(radius=4.0, ray_id=0, device=dev, transform=ns.RayTransformBundle(translation=[0.0, 0.0, -10.0]).to(dev))
for label, mk, n in (("singlet+stop+sensor", small, 4096), ("singlet+stop+sensor", small, 128000), ("32 surfaces", big, 4096)):
    sc = mk()
    sc.rays = b.sample(n, seed=1, offset=0)
    sc._build_index_maps()
    for _ in range(200):
        sc.step()
    torch.cuda.synchronize()
    reps = 3000
    t0 = time.perf_counter()
    for _ in range(reps):
        sc.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{label:22s} n={n:7d}  host {1e6*(t1-t0)/reps:6.2f} us per step()   with drain {1e6*(t2-t0)/reps:6.2f} us", flush=True)
