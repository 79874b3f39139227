"""GPU: time of one second-moment pass (rtt_spot_moments_f32, pass 2) over 32 Mi dense records for block counts and
block sizes (rtt_tune keys 4 and 8)."""
import sys, torch
sys.path.insert(0, '.')
import raytracetorch_amd as r
from raytracetorch_amd import kernels as K, _native as N
n = 32 * 2**20
dev = 'cuda:0'
x = torch.randn(n, device=dev); y = torch.randn(n, device=dev); w = torch.ones(n, device=dev); m = torch.ones(n, dtype=torch.uint8, device=dev)
out = torch.zeros(6, dtype=torch.float64, device=dev)
for threads in (256, 1024):
    N.lib().rtt_tune(8, threads)
    for blocks in (128, 256, 384, 512, 768, 1024, 2048):
        N.lib().rtt_tune(4, blocks)
        p = 2
        for _ in range(3): K.spot_moments(x, y, w, m, p, 0.0, 0.0, float(n), out)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): K.spot_moments(x, y, w, m, p, 0.0, 0.0, float(n), out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"threads {threads:5d} blocks {blocks:5d}  {us:7.1f} us  {n * 13 / us / 1e6:5.2f} TB/s", flush=True)
