import sys, torch, time
sys.path.insert(0, '.')
import raytracetorch_amd as r
from raytracetorch_amd import kernels as K, _native as N
n = 32 * 2**20
dev = 'cuda:0'
x = torch.randn(n, device=dev); y = torch.randn(n, device=dev); w = torch.ones(n, device=dev); m = torch.ones(n, dtype=torch.uint8, device=dev)
out = torch.zeros(4, dtype=torch.float64, device=dev)
for blocks in (256, 512, 768, 1024, 1536, 2048):
    N.lib().rtt_tune(4, blocks)
    for p in (0, 1):
        for _ in range(3): K.spot_moments(x, y, w, m, p, 0.0, 0.0, float(n), out)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): K.spot_moments(x, y, w, m, p, 0.0, 0.0, float(n), out)
        e1.record(); torch.cuda.synchronize()
        print(blocks, p, e0.elapsed_time(e1) / 20 * 1e3, 'us', n * 13 / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e12, 'TB/s', flush=True)
