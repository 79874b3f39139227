"""Where a wavefront of the reverse sweep spends its cycles (profiling build, `make -C raytracetorch_amd/csrc prof`):
    RTT_HIP_LIB=$PWD/raytracetorch_amd/csrc/librtt_hip_prof.so [RTT_BACKWARD=retrace|adjoint] python tools/phase_timers.py [rays]
Runs a few spot-size steps (singlet + sensor, 5 trainable scalars) and prints the shader-clock cycles per wavefront and
tile between the marks of k_trace_seq_adjoint.  Every mark first waits for all memory operations in flight, so a
wait is charged to the phase that issued it; the marks cost some overlap, the proportions are what counts."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import raytracetorch_amd as rta   # noqa: E402
import scenes                      # noqa: E402
from raytracetorch_amd import _native as N   # noqa: E402

NAMES = ["ray loads", "forward sweep / tape staging", "visit: adjoint in, state, walk", "visit: record, dispatch",
         "visit: frames + normal again", "visit: physics adjoint", "visit: normal adjoint", "visit: root adjoint",
         "visit: frames adjoint", "ray adjoint out, loop"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32 * 2 ** 20
steps = 5
ns = scenes.namespace_from_package(rta)
dev = torch.device("cuda:0")
lens, sensor = scenes.spot_system(ns, grads=True)
scene = ns.SequentialScene([lens, sensor]).to(dev)
b = ns.CollimatedDisk(radius=4.0, ray_id=0, device=dev, transform=ns.RayTransformBundle(translation=[0.0, 0.0, -30.0]).to(dev))
src = b.sample(n, seed=1000, offset=0)
src._pos


class Resident(torch.nn.Module):
    def sample(self, _n):
        return ns.Rays.from_planes(src._pos, src._dir, src.intensity, src.id, src.wavelength)


loss_fn = ns.SpotSizeLoss(sensor, [Resident()], N_rays=n)
lib = N.lib()
lib.rtt_debug_phases.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
out = (ctypes.c_ulonglong * 16)()
for it in range(steps + 2):
    if it == 2:
        lib.rtt_debug_phases(out, 16, 1)   # forget the warm-up
    for p in scene.parameters():
        p.grad = None
    loss_fn(scene).backward()
lib.rtt_debug_phases(out, 16, 0)
tiles = steps * (n + 511) // 512      # a tile = 512 rays of a block = 128 rays of each of its 4 wavefronts
waves_tiles = tiles * 4
total = sum(out[:10])
print(f"mode {os.environ.get('RTT_BACKWARD', 'adjoint')}, {n} rays: {total / waves_tiles:.0f} cycles per wavefront and tile")
for k, name in enumerate(NAMES):
    print(f"  {name:34s} {out[k] / waves_tiles:9.0f}  {100.0 * out[k] / total:5.1f} %")
