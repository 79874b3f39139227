"""Where the host time of one small SequentialScene.simulate goes: each piece timed alone over many repetitions."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import raytracetorch_amd as rta  # noqa: E402
from raytracetorch_amd import kernels as K, _native as N, program as P  # noqa: E402
import scenes  # noqa: E402

ns = scenes.namespace_from_package(rta)
dev = torch.device("cuda:0")
els = list(scenes.spot_system(ns, grads=False))
scene = ns.SequentialScene(els)
n = 1024
rays = scenes.disk_bundle(ns).sample(n).to(dev)
out = rays.clone()


def bench(label, fn, reps=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{label:46s} {1e6 * (t1 - t0) / reps:7.2f} us", flush=True)


with torch.no_grad():
    prog = scene._program()
    bench("whole simulate (+ sensor reset)", lambda: (els[1].reset(), scene.simulate(rays, out=out)))
    bench("sensor reset", lambda: els[1].reset())
    bench("scene._program()", scene._program)
    bench("scene._needs_grad(rays)", lambda: scene._needs_grad(rays))
    bench("K._check_rays x2", lambda: (K._check_rays(rays), K._check_rays(out)))
    bench("K._sensor_buffers", lambda: K._sensor_buffers(prog, n, dev, zero_mask=False))
    bench("DeviceProgram.upload (cached)", lambda: N.DeviceProgram.upload(prog, dev))
    bench("N.stream_handle", lambda: N.stream_handle(dev))
    bench("K.trace_sequential", lambda: K.trace_sequential(rays, prog, out=out))
    sh, sw, sm = K.trace_sequential(rays, prog, out=out)
    bench("deliver records (+ reset)", lambda: (els[1].reset(), scene._deliver_sensor_records(prog, rays, sh, sw, sm, dst=out)))
    bench("torch.empty (one)", lambda: torch.empty((3, n), device=dev))
    lib = N.lib()
    dp = N.DeviceProgram.upload(prog, dev)
    args = (rays._pos.data_ptr(), rays._dir.data_ptr(), rays.intensity.data_ptr(), n, out._pos.data_ptr(), out._dir.data_ptr(),
            out.intensity.data_ptr(), n, n, dp.dev.data_ptr(), dp.host_ptr, dp.n, N.ptr(sh), N.ptr(sw), N.ptr(sm), n, N.stream_handle(dev))
    bench("the foreign call alone (launch)", lambda: lib.rtt_trace_seq_f32(*args))
    bench("7 data_ptr()", lambda: (rays._pos.data_ptr(), rays._dir.data_ptr(), rays.intensity.data_ptr(), out._pos.data_ptr(),
                                   out._dir.data_ptr(), out.intensity.data_ptr(), dp.dev.data_ptr()))
