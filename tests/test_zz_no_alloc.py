"""GPU: rtt_trace_seq_adjoint_static_f32, the entry point that allocates nothing (its block-sum scratch is a static
device array of the library), gives what rtt_trace_seq_adjoint_f32 gives with a caller's workspace — on two calls in a
row, the second with a smaller bundle, so that what the first call left in the shared scratch cannot leak into it."""
import numpy as np
import pytest
import torch

import scenes

import raytracetorch_amd.program as P
from raytracetorch_amd import _native as N

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _call(ns, static, rays, prog, masks, adj_pos, adj_dir, adj_sens, n_acc):
    lib = N.lib()
    n = len(rays)
    dp = N.DeviceProgram.upload(prog, DEV)
    slot_adj = torch.full((n_acc,), float("nan"), dtype=torch.float64, device=DEV)
    ray_adj = torch.full((6, n), float("nan"), device=DEV)
    head = (N.ptr(rays._pos), N.ptr(rays._dir), N.ptr(rays.intensity), n, n, N.ptr(dp.dev), dp.host_ptr, dp.n, masks.ctypes.data,
            N.ptr(adj_pos), N.ptr(adj_dir), n, N.ptr(adj_sens), n, -1, 0, 0.0, 0.0, 1.0, 0, N.ptr(ray_adj), n, N.ptr(slot_adj))
    if static:
        code = lib.rtt_trace_seq_adjoint_static_f32(*head, N.stream_handle(DEV))
    else:
        ws = torch.empty(int(lib.rtt_trace_seq_adjoint_workspace_bytes(n_acc)), dtype=torch.uint8, device=DEV)
        code = lib.rtt_trace_seq_adjoint_f32(*head, N.ptr(ws), ws.numel(), N.stream_handle(DEV))
    N.check(code, "adjoint entry point")
    torch.cuda.synchronize()
    return slot_adj.cpu(), ray_adj.cpu()


def test_static_scratch_entry_point_equals_the_workspace_one(ns):
    lens, sensor = scenes.spot_system(ns, grads=False)
    prog = P.compile_elements([lens, sensor])
    pairs = [(0, 23), (0, 24), (0, 26), (0, 30), (1, 23), (1, 24), (1, 31), (3, 21), (3, 23)]
    masks = np.zeros(4, dtype=np.uint32)
    for s, j in pairs:
        masks[s] |= np.uint32(1 << j)
    for n in (700001, 30001):   # 1368 blocks of two-ray lanes, then 59: the second call reuses a corner of the scratch
        g = torch.Generator().manual_seed(n)
        rays = scenes.disk_bundle(ns, radius=4.0, z=-30.0, rot=[0.01, 0.02, 0.0]).sample(n).to(DEV)
        adj_pos, adj_dir = (torch.randn(3, n, generator=g).to(DEV) for _ in range(2))
        adj_sens = torch.randn(1, 3, n, generator=g).to(DEV)
        want = _call(ns, False, rays, prog, masks, adj_pos, adj_dir, adj_sens, len(pairs))
        got = _call(ns, True, rays, prog, masks, adj_pos, adj_dir, adj_sens, len(pairs))
        assert torch.isfinite(got[0]).all() and torch.isfinite(got[1]).all()
        # the same kernels add the same block sums in the same order: identical bits, well inside the 1e-4 gradient bar
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    # more slot sums than the scratch has columns for: refused, nothing launched
    lib = N.lib()
    prog9 = P.compile_elements(scenes.system5(ns))
    many = np.full(9, 0xFFFFFFFF, dtype=np.uint32)   # 288 slot sums asked for, 128 columns exist
    dp = N.DeviceProgram.upload(prog9, DEV)
    code = lib.rtt_trace_seq_adjoint_static_f32(N.ptr(rays._pos), N.ptr(rays._dir), N.ptr(rays.intensity), n, n, N.ptr(dp.dev), dp.host_ptr, dp.n, many.ctypes.data,
              0, 0, n, 0, n, -1, 0, 0.0, 0.0, 1.0, 0, 0, 0, N.ptr(rays._pos), N.stream_handle(DEV))
    assert code == -2 and b"size" in lib.rtt_strerror(code)


def test_static_entry_point_allocates_nothing_and_replays_in_a_graph(ns):
    """After one preparing call per device (a call without rays does it: symbol lookup, LDS grant, code object load) the
    entry point leaves the allocator's counters untouched, and a capture of it replays to the same bits."""
    lib = N.lib()
    lens, sensor = scenes.spot_system(ns, grads=False)
    prog = P.compile_elements([lens, sensor])
    pairs = [(0, 23), (0, 24), (1, 24), (1, 31), (3, 23)]
    masks = np.zeros(4, dtype=np.uint32)
    for s, j in pairs:
        masks[s] |= np.uint32(1 << j)
    n = 50001
    rays = scenes.disk_bundle(ns, radius=4.0, z=-30.0).sample(n).to(DEV)
    g = torch.Generator().manual_seed(1)
    adj_pos = torch.randn(3, n, generator=g).to(DEV)
    dp = N.DeviceProgram.upload(prog, DEV)
    slot_adj = torch.zeros(len(pairs), dtype=torch.float64, device=DEV)

    def call(count, stream):
        return lib.rtt_trace_seq_adjoint_static_f32(N.ptr(rays._pos), N.ptr(rays._dir), N.ptr(rays.intensity), n, count, N.ptr(dp.dev),
                                                    dp.host_ptr, dp.n, masks.ctypes.data, N.ptr(adj_pos), 0, n, 0, n, -1, 0, 0.0, 0.0,
                                                    1.0, 0, 0, 0, N.ptr(slot_adj), stream)
    assert call(0, N.stream_handle(DEV)) == 0          # the preparing call: no rays, zeros in slot_adj
    torch.cuda.synchronize()
    assert float(slot_adj.abs().max()) == 0.0
    before = (torch.cuda.memory_allocated(DEV), torch.cuda.memory_reserved(DEV))
    assert call(n, N.stream_handle(DEV)) == 0
    torch.cuda.synchronize()
    assert (torch.cuda.memory_allocated(DEV), torch.cuda.memory_reserved(DEV)) == before
    eager = slot_adj.clone()
    assert float(eager.abs().max()) > 0.0
    slot_adj.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        assert call(n, N.stream_handle(DEV)) == 0     # the capture stream is torch's current stream inside the context
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(slot_adj, eager)
