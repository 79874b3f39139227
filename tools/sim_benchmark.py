"""The workload of the reference's own benchmark script (benchmarks/sim_benchmark.py:49-151) through this package.

    python tools/sim_benchmark.py                     # this package on cuda:0
    python tools/sim_benchmark.py --impl reference    # the reference itself on the CPU (only where /root/reference exists)

Scene: singlet lens (c1=.05, c2=-.05, d=10, t=3, n=1.5) + CircularAperture(5) + Sensor(Disk(6) at z=19) in a
`Scene`; a CollimatedDisk(4) registered with `add_bundle`; per run `_build_rays()`, `_build_index_maps()` and up to 20
`step()` calls while any ray is alive, wall time between two device synchronisations.  Ray counts, repeats and warm-up
follow the script (BENCH_REPEATS / BENCH_WARMUP are honoured).  `--fused` additionally times `Scene.simulate()` with
`Nbounces = 20`, the same bounce loop as one launch.  Prints one JSON line per ray count.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

RAY_COUNTS = [4096, 16_384, 64_000, 128_000]
NBOUNCES = 20


def build_scene(ns, device):
    scene = ns.Scene()
    scene.add_element(ns.SingletLens(c1=0.05, c2=-0.05, d=10.0, t=3.0, ior_glass=1.5, ior_media=1.0))
    scene.add_element(ns.CircularAperture(radius=5.0, transform=ns.RayTransform(translation=[0.0, 0.0, 0.0])))
    scene.add_element(ns.Sensor(shape=ns.Disk(radius=6.0, transform=ns.RayTransform(translation=[0.0, 0.0, 19.0]))))
    scene.to(device)
    scene._build_index_maps()
    bundle = ns.CollimatedDisk(radius=4.0, ray_id=0, device=device, dtype=torch.float32, transform=None).to(device)
    return scene, bundle


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--impl", default="package", choices=["package", "reference"])
    ap.add_argument("--fused", action="store_true", help="also time Scene.simulate() (one launch for the bounce loop)")
    ap.add_argument("--rays", type=int, nargs="*", default=RAY_COUNTS)
    a = ap.parse_args()
    repeats = int(os.environ.get("BENCH_REPEATS", 20))
    warmup = int(os.environ.get("BENCH_WARMUP", 3))
    import scenes
    if a.impl == "package":
        import raytracetorch_amd as rta
        ns = scenes.namespace_from_package(rta)
        device = torch.device("cuda", 0)
    else:
        import ref_loader
        ns = scenes.namespace_from_package(ref_loader.load())
        device = torch.device("cpu")

    def sync():
        if device.type == "cuda":
            torch.cuda.synchronize()

    def run_steps(scene):
        scene._build_rays()
        sync()
        t0 = time.perf_counter()
        scene._build_index_maps()
        bounces = 0
        for _ in range(NBOUNCES):
            if not (scene.rays.intensity > 0).any():
                break
            scene.step()
            bounces += 1
        sync()
        return time.perf_counter() - t0, bounces

    def run_fused(scene):
        scene._build_rays()
        rays = scene.rays
        sync()
        t0 = time.perf_counter()
        scene.rays = rays
        scene.Nbounces = NBOUNCES
        scene.simulate()
        sync()
        return time.perf_counter() - t0, NBOUNCES

    for n in a.rays:
        for label, run in (("step loop", run_steps),) + ((("fused simulate()", run_fused),) if a.fused else ()):
            scene, bundle = build_scene(ns, device)
            scene.clear_bundles()
            scene.add_bundle(bundle, n)
            with torch.no_grad():
                for _ in range(warmup):
                    run(scene)
                runs = [run(scene) for _ in range(repeats)]
            times = [t for t, _ in runs]
            print(json.dumps({"impl": a.impl, "device": str(device), "mode": label, "n_rays": n, "bounces": runs[-1][1],
                              "mean_ms": statistics.mean(times) * 1e3,
                              "stdev_ms": statistics.stdev(times) * 1e3 if len(times) > 1 else 0.0,
                              "threads": torch.get_num_threads()}), flush=True)


if __name__ == "__main__":
    main()
