import sys, torch
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/oracle"]
import raytracetorch_amd as rta, scenes
from raytracetorch_amd import kernels as K, program as P
ns = scenes.namespace_from_package(rta)
torch.manual_seed(3)
lens, sensor = scenes.spot_system(ns, grads=False)
prog = P.compile_elements([lens, sensor])
for n in (20001, 20000, 512, 1024*37):
    src = scenes.disk_bundle(ns).sample(n).to("cuda:0")
    a = src.clone(); K.trace_sequential(a, prog)
    b = src.clone(); o = src.clone(); o._pos.zero_(); o._dir.zero_(); o.intensity.zero_()
    h, w, m, tape = K.trace_sequential(b, prog, out=o, tape=True)
    torch.cuda.synchronize()
    dp = (a._pos != o._pos); dd = (a._dir != o._dir)
    print(n, "pos diff", int(dp.sum()), "dir diff", int(dd.sum()), "nan", int(torch.isnan(a._pos).sum()), int(torch.isnan(o._pos).sum()),
          "max", float((a._pos - o._pos).abs().nan_to_num().max()), "where", dp.nonzero()[:5].tolist())
    print("   input untouched", torch.equal(b._pos, src._pos))
