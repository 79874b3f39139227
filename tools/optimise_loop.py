"""The spot-size step inside a real optimisation loop (the parameters change every step, so the scene is packed again and
the slot Jacobians are derived again each time) against the benchmark's repeated step on unchanged parameters."""
import os, sys, time, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import raytracetorch_amd as rta
import scenes
ns = scenes.namespace_from_package(rta)
dev = torch.device("cuda", 0)
n = int(os.environ.get("RAYS", 32 * 2 ** 20))
b = ns.CollimatedDisk(radius=4.0, ray_id=0, device=dev, transform=ns.RayTransformBundle(translation=[0.0, 0.0, -30.0]).to(dev))
src = b.sample(n, seed=1000)
src._pos


class Resident(torch.nn.Module):
    def sample(self, _n):
        return ns.Rays.from_planes(src._pos, src._dir, src.intensity, src.id, src.wavelength)


for where in ("device", "host"):
    for update in (False, True):
        lens, sensor = scenes.spot_system(ns, grads=True)
        scene = ns.SequentialScene([lens, sensor])
        if where == "device":
            scene.to(dev)
        loss_fn = ns.SpotSizeLoss(sensor, [Resident()], N_rays=n)
        opt = torch.optim.SGD([p for p in scene.parameters() if p.requires_grad], lr=1e-7)
        ts = []
        for i in range(25):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            opt.zero_grad(set_to_none=True)
            loss = loss_fn(scene)
            loss.backward()
            if update:
                opt.step()
            torch.cuda.synchronize()
            if i >= 5:
                ts.append(time.perf_counter() - t0)
        print(f"parameters on the {where:6s}, {'updated every step' if update else 'unchanged         '}: "
              f"{statistics.median(ts) * 1e3:7.3f} ms per step (min {min(ts) * 1e3:.3f})  loss {float(loss):.6f}", flush=True)
