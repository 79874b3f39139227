"""Diagnose one fuzz case of the nearest-hit step against the oracle: python tools/debug_fuzz.py <gen> <seed>"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import raytracetorch_amd as rta  # noqa: E402
import raytracetorch_amd.program as P  # noqa: E402
import rtt_oracle as O  # noqa: E402
import scenes  # noqa: E402
from raytracetorch_amd import _native, kernels as K  # noqa: E402

gen, seed = sys.argv[1], int(sys.argv[2])
ns = scenes.namespace_from_package(rta)
els = getattr(scenes, gen)(ns, np.random.default_rng(seed))
print([type(e).__name__ + ":" + type(e.shape).__name__ for e in els])
torch.manual_seed(seed)
rays = scenes.disk_bundle(ns, radius=5.0, z=-15.0, rot=[0.03 * math.sin(seed), 0.02, 0.0]).sample(3001)
p0, d0, I0 = rays.pos.clone(), rays.dir.clone(), rays.intensity.clone()
prog = P.compile_elements(els)
to, wo, tm = O.ray_cast(prog, p0, d0)
o = O.step_nearest(prog, p0, d0, I0)
if torch.cuda.is_available():
    for cull in (0, 1):
        _native.check(_native.lib().rtt_tune(3, cull), "tune")
        dev = rays.to("cuda:0")
        w, t = K.ray_cast(dev, prog)
        st = dev.clone()
        K.step_nearest(st, prog)
        w, t = w.cpu(), t.cpu()
        bad = (w != wo.to(torch.int32))
        err = (st.pos.cpu() - o["pos"]).abs().max(1).values
        print("cull", cull, "winner mismatches", int(bad.sum()), "pos err max", float(err.max()), "n>2e-5", int((err > 2e-5).sum()))
        for i in torch.nonzero(err > 2e-5).flatten()[:5].tolist():
            srt = np.sort(tm[i].numpy())
            print("  ray", i, "win gpu", int(w[i]), "oracle", int(wo[i]), "t gpu", float(t[i]), "t oracle", float(to[i]), "two smallest", srt[:2],
                  "kinds", int(prog[int(wo[i])]["surf_kind"]), int(prog[int(wo[i])]["phys_kind"]), "err", float(err[i]))
