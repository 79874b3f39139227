"""GPU: the reverse-mode backward of the sequential trace (rtt_trace_seq_adjoint_f32) against the forward-mode
kernels (rtt_trace_seq_tangent_f32 / _ray_adjoint_f32), which differentiate the same formulas independently.  The
goldens from the reference's autograd are checked in tests/test_gpu_autograd.py through whichever mode is the default;
here the two modes are held against each other, the entry point is called directly through the C ABI, and the
LDS-budget edges (slot masks split over several launches, programs too long for the kernel) are exercised."""
import ctypes
import math

import numpy as np
import pytest
import torch

import scenes
from util import golden

import raytracetorch_amd.autograd as AG
import raytracetorch_amd.program as P
from raytracetorch_amd import _native as N

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _grads(ns, build, objective, mode, monkeypatch, calls=None):
    monkeypatch.setenv("RTT_BACKWARD", mode)
    torch.manual_seed(0)
    els, rays = build()
    params = {k: p for k, p in torch.nn.ModuleList(els).named_parameters() if p.requires_grad}
    if rays._pos.is_leaf:
        rays._pos.requires_grad_(True)
        rays._dir.requires_grad_(True)
    if calls is not None:   # the reverse sweeps, with their own trace inside or from a tape
        lib = N.lib()
        for name in ("rtt_trace_seq_adjoint_f32", "rtt_trace_seq_adjoint_tape_f32"):
            def counted(*a, real=getattr(lib, name), name=name):
                calls.append(name)
                return real(*a)
            monkeypatch.setattr(lib, name, counted, raising=False)
    obj = objective(els, rays)
    obj.backward()
    out = {k: p.grad.detach().double().cpu().reshape(-1) for k, p in params.items() if p.grad is not None}
    out["rays.pos"] = rays._pos.grad.detach().double().cpu().reshape(-1)
    out["rays.dir"] = rays._dir.grad.detach().double().cpu().reshape(-1)
    return float(obj.detach()), out


def _projection(ns, seed):
    def objective(els, rays):
        g = torch.Generator().manual_seed(seed)
        n = len(rays)
        wp, wd, ws = (torch.randn(n, 3, generator=g).to(DEV) for _ in range(3))
        src_pos, src_dir = rays._pos, rays._dir
        r = ns.SequentialScene(els).simulate(ns.Rays.from_planes(src_pos, src_dir, rays.intensity, rays.id, rays.wavelength))
        obj = (r.pos * wp).sum() + (r.dir * wd).sum()
        for el in els:
            if hasattr(el, "getHitsTensors") and el.hitLocs:
                locs, w, _ = el.getHitsTensors()
                obj = obj + (locs * ws[: locs.shape[0]] * w[:, None]).sum()
        return obj
    return objective


def _compare(a, b, tol=3e-4, system_scale=False):
    """Per gradient, relative to its own magnitude (at least 1), or, `system_scale`, to the largest gradient of the system:
    two float32 evaluations of one derivative differ by rounding that scales with the terms summed, not with a
    gradient that is small because they cancel."""
    assert set(a) == set(b)
    big = max(float(v.abs().max()) for v in b.values())
    for k in a:
        scale = max(1.0, big if system_scale else float(b[k].abs().max()))
        assert float((a[k] - b[k]).abs().max()) <= tol * scale, (k, a[k], b[k])


CASES = {
    "spot_system": lambda ns: list(scenes.spot_system(ns, grads=True)),
    "tilted": lambda ns: list(scenes.tilt_system(ns, [0.03, -0.02, 0.1])),
    "identity_rotation": lambda ns: list(scenes.tilt_system(ns, [0.0, 0.0, 0.0])),
    "system5": lambda ns: scenes.system5(ns),
}


@pytest.mark.parametrize("name", list(CASES))
def test_reverse_and_forward_mode_agree(ns, name, monkeypatch):
    def build():
        els = CASES[name](ns)
        scenes.make_all_trainable(els)
        rays = scenes.disk_bundle(ns, radius=4.5, z=-30.0, rot=[0.01, -0.02, 0.0]).sample(5001).to(DEV)
        return els, rays
    calls = []
    obj_a, ga = _grads(ns, build, _projection(ns, 3), "adjoint", monkeypatch, calls)
    assert calls and set(calls) == {"rtt_trace_seq_adjoint_f32"}, "the reverse-mode entry point was not used"
    obj_t, gt = _grads(ns, build, _projection(ns, 3), "tangent", monkeypatch)
    assert obj_a == obj_t
    assert len(ga) > 4
    _compare(ga, gt)
    # and the sweep that reads a tape of the forward trace instead of tracing again gives the same gradients; programs
    # too long for it fall back to the sweep with its own trace
    calls = []
    obj_r, gr = _grads(ns, build, _projection(ns, 3), "tape", monkeypatch, calls)
    short = len(P.compile_elements(CASES[name](ns))) <= 6
    assert set(calls) == {"rtt_trace_seq_adjoint_tape_f32" if short else "rtt_trace_seq_adjoint_f32"}
    assert abs(obj_r - obj_a) <= 1e-6 * abs(obj_a)    # (the trace that writes a tape is another instantiation of the kernel)
    _compare(gr, ga, tol=2e-5, system_scale=True)


@pytest.mark.parametrize("gen", ["random_system", "random_system_extra"])
@pytest.mark.parametrize("seed", range(12))
def test_random_systems_in_both_modes(ns, gen, seed, monkeypatch):
    """Every element kind of the random generators (doublets, cylindrical lenses, boxes, cones, ideal elements,
    mirrors, stops) with every floating-point parameter trainable, and gradients with respect to the rays."""
    def build():
        els = getattr(scenes, gen)(ns, np.random.default_rng(seed))
        scenes.make_all_trainable(els)
        rays = scenes.disk_bundle(ns, radius=5.0, z=-15.0, rot=[0.03 * math.sin(seed), 0.02, 0.0]).sample(1501).to(DEV)
        return els, rays
    obj_a, ga = _grads(ns, build, _projection(ns, seed), "adjoint", monkeypatch)
    obj_t, gt = _grads(ns, build, _projection(ns, seed), "tangent", monkeypatch)
    if not np.isfinite(obj_t):
        pytest.skip("objective not finite for this system (NaN roots of unbounded surfaces)")
    assert obj_a == obj_t
    # both are float32 evaluations of the same derivative; ill-conditioned systems (paraboloids, long free paths)
    # amplify the rounding of either, so the bar is relative to the largest gradient of the system
    big = max(float(v.abs().max()) for v in gt.values())
    for k in gt:
        assert float((ga[k] - gt[k]).abs().max()) <= 2e-3 * max(1.0, big), (k, ga[k], gt[k])


def _many_singlets(ns, count):
    els = [scenes.singlet(ns, c1=0.02, c2=-0.02, d=12.0, t=1.0, c1_grad=True, c2_grad=True, t_grad=True, ior_glass_grad=True,
                          transform=ns.RayTransform(translation=[0.0, 0.0, 3.0 * k])) for k in range(count)]
    return els + [ns.Sensor(ns.Disk(20.0, transform=ns.RayTransform(translation=[0.0, 0.0, 3.0 * count + 2.0])))]


@pytest.mark.parametrize("count,launches", [(6, "several"), (8, None)])
def test_slot_budget_edges(ns, count, launches, monkeypatch):
    """19 surfaces leave room for 27 slot accumulators per launch, fewer than the wanted slots of six trainable
    singlets, which therefore take several launches; 25 surfaces leave none for one sweep, and the program is swept in
    stretches of whole elements (a single element that long is left to the forward-mode kernels)."""
    def build():
        return _many_singlets(ns, count), scenes.disk_bundle(ns, radius=3.0, z=-10.0).sample(3001).to(DEV)
    n_surf = 3 * count + 1
    budget = N.lib().rtt_trace_seq_adjoint_max_slots(n_surf)
    assert budget == max(0, 160 - 7 * n_surf)
    calls = []
    obj_a, ga = _grads(ns, build, _projection(ns, 1), "adjoint", monkeypatch, calls)
    obj_t, gt = _grads(ns, build, _projection(ns, 1), "tangent", monkeypatch)
    assert abs(obj_a - obj_t) <= 1e-6 * abs(obj_t)
    if launches is None:
        assert budget == 0 and len(calls) >= 2
    else:
        assert len(calls) >= 2, (len(calls), budget)
    _compare(ga, gt)


def test_entry_point_directly(ns):
    """Through the C ABI: the adjoint of one record slot, summed over the rays, equals the forward-mode contraction
    with the unit tangent of that slot; the input-ray adjoint equals the per-ray forward-mode one; two calls give the
    same bits (the block sums are added in a fixed order)."""
    lib = N.lib()
    lens, sensor = scenes.spot_system(ns, grads=False)
    prog = P.compile_elements([lens, sensor])
    n = 40001
    rays = scenes.disk_bundle(ns, radius=4.0, z=-30.0, rot=[0.01, 0.02, 0.0]).sample(n).to(DEV)
    dp = N.DeviceProgram.upload(prog, DEV)
    g = torch.Generator().manual_seed(5)
    adj_pos, adj_dir = (torch.randn(3, n, generator=g).to(DEV) for _ in range(2))
    adj_sens = torch.randn(1, 3, n, generator=g).to(DEV)
    pairs = [(0, 23), (0, 24), (0, 26), (0, 30), (1, 23), (1, 24), (1, 31), (3, 21), (3, 23)]
    masks = np.zeros(4, dtype=np.uint32)
    for s, j in pairs:
        masks[s] |= np.uint32(1 << j)
    order = sorted(pairs)
    outs = []
    for _ in range(2):
        slot_adj = torch.empty(len(pairs), dtype=torch.float64, device=DEV)
        ray_adj = torch.empty((6, n), device=DEV)
        ws = torch.empty(int(lib.rtt_trace_seq_adjoint_workspace_bytes(len(pairs))), dtype=torch.uint8, device=DEV)
        code = lib.rtt_trace_seq_adjoint_f32(N.ptr(rays._pos), N.ptr(rays._dir), N.ptr(rays.intensity), n, n, N.ptr(dp.dev),
                                             dp.host_ptr, dp.n, masks.ctypes.data, N.ptr(adj_pos), N.ptr(adj_dir), n, N.ptr(adj_sens), n,
                                             -1, 0, 0.0, 0.0, 1.0, 0, N.ptr(ray_adj), n, N.ptr(slot_adj), N.ptr(ws), ws.numel(),
                                             N.stream_handle(DEV))
        N.check(code, "rtt_trace_seq_adjoint_f32")
        outs.append((slot_adj.cpu(), ray_adj.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    table = np.zeros((4, 32, len(order)), dtype=np.float32)
    for k, (s, j) in enumerate(order):
        table[s, j, k] = 1.0
    want = AG._tangent_contract(rays, prog, table, adj_pos, adj_dir, adj_sens).cpu()
    scale = float(want.abs().max())
    assert float((outs[0][0] - want).abs().max()) <= 2e-4 * scale, (outs[0][0], want)
    want_rays = AG._ray_adjoint(rays, prog, adj_pos, adj_dir, adj_sens).cpu()
    assert float((outs[0][1] - want_rays).abs().max()) <= 2e-4 * float(want_rays.abs().max())
    # one ray per lane and two rays per lane run the same arithmetic per ray
    assert lib.rtt_tune(6, 1) == 0
    try:
        slot_adj = torch.empty(len(pairs), dtype=torch.float64, device=DEV)
        code = lib.rtt_trace_seq_adjoint_f32(N.ptr(rays._pos), N.ptr(rays._dir), N.ptr(rays.intensity), n, n, N.ptr(dp.dev),
                                             dp.host_ptr, dp.n, masks.ctypes.data, N.ptr(adj_pos), N.ptr(adj_dir), n, N.ptr(adj_sens), n,
                                             -1, 0, 0.0, 0.0, 1.0, 0, 0, 0, N.ptr(slot_adj), N.ptr(ws), ws.numel(), N.stream_handle(DEV))
        N.check(code, "rtt_trace_seq_adjoint_f32")
        assert float((slot_adj.cpu() - outs[0][0]).abs().max()) <= 1e-5 * scale
    finally:
        lib.rtt_tune(6, 0)


def test_forward_mode_still_reproduces_the_reference(ns, monkeypatch):
    """The spot-size gradients of the reference's autograd through the forward-mode kernels (the default path is
    checked against the same golden in tests/test_gpu_autograd.py)."""
    monkeypatch.setenv("RTT_BACKWARD", "tangent")
    g = golden("spot_f32")
    torch.manual_seed(0)
    lens, sensor = scenes.spot_system(ns, grads=True)
    loss = ns.SpotSizeLoss(sensor, [scenes.disk_bundle(ns)], N_rays=65536)(ns.SequentialScene([lens, sensor]))
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-5 * float(g["loss"])
    s0, s1 = lens.shape.surfaces[0], lens.shape.surfaces[1]
    got = dict(c1=s0.c.grad, c2=s1.c.grad, ior_glass=lens.ior_glass.grad)
    for k, v in got.items():
        assert abs(float(v) - float(g["g_" + k])) <= 2e-4 * abs(float(g["g_" + k])), k


@pytest.mark.parametrize("n", [1, 511, 70001, 2 ** 20 + 3])
def test_trace_with_moments_equals_trace_then_moment_pass(ns, n):
    """rtt_trace_seq_moments_f32: rays and sensor record identical to the plain trace, and the three first moments
    equal to the float64 moment pass over that record (they are summed in float32 within a wavefront)."""
    import raytracetorch_amd.kernels as K
    elements = scenes.system5(ns)
    prog = P.compile_elements(elements)
    rays = scenes.disk_bundle(ns, radius=4.0, z=-30.0, rot=[0.004, -0.003, 0.0]).sample(n).to(DEV)
    rays.intensity = (torch.rand(n, device=DEV) > 0.1).float() * (0.5 + torch.rand(n, device=DEV))   # some rays dark
    a, b = rays.clone(), rays.clone()
    hit0, w0, m0 = K.trace_sequential(a, prog)
    mom = torch.full((6,), 7.0, dtype=torch.float64, device=DEV)
    hit1, w1, m1 = K.trace_sequential(b, prog, moments=(0, mom))
    assert torch.equal(a._pos, b._pos) and torch.equal(a._dir, b._dir) and torch.equal(a.intensity, b.intensity)
    assert torch.equal(m0, m1)
    on = m0[0].bool()
    assert torch.equal(hit0[0][:, on], hit1[0][:, on]) and torch.equal(w0[0][on], w1[0][on])
    want = torch.zeros(4, dtype=torch.float64, device=DEV)
    K.spot_moments(hit0[0][0].contiguous(), hit0[0][1].contiguous(), w0[0], m0[0], 0, 0.0, 0.0, 1.0, want)
    got, want = mom[:3].cpu(), want[:3].cpu()
    assert torch.equal(mom[3:].cpu(), torch.full((3,), 7.0, dtype=torch.float64))    # only the first three are written
    scale = float(want[0]) * 4.0 + 1e-30      # |x|, |y| <= 4 on this sensor
    assert float((got - want).abs().max()) <= 2e-6 * scale, (got, want)
    lib = N.lib()
    dp = N.DeviceProgram.upload(prog, DEV)
    ws = torch.empty(int(lib.rtt_trace_seq_moments_workspace_bytes(n)), dtype=torch.uint8, device=DEV)
    args = lambda slot, ws_bytes: (N.ptr(a._pos), N.ptr(a._dir), N.ptr(a.intensity), n, N.ptr(b._pos), N.ptr(b._dir), N.ptr(b.intensity),
                                   n, n, N.ptr(dp.dev), dp.host_ptr, dp.n, N.ptr(hit1), N.ptr(w1), N.ptr(m1), n, slot, N.ptr(mom),
                                   N.ptr(ws), ws_bytes, N.stream_handle(DEV))
    assert lib.rtt_trace_seq_moments_f32(*args(1, ws.numel())) == -2     # the program has one sensor slot
    assert lib.rtt_trace_seq_moments_f32(*args(0, ws.numel() - 8)) == -2  # workspace too small


def test_fused_spot_loss_leaves_the_traced_rays_to_be_made_on_demand(ns):
    """The fused spot-size loss reads the sensor only: its trace does not write the traced bundle
    (rtt_trace_seq_moments_f32 with null ray outputs), and `scene.rays` is traced when somebody reads it.  Loss and
    gradients are those of the trace that writes everything; the rays, once read, are those of `simulate`."""
    from raytracetorch_amd.rays.ray import Rays, SampledRays
    from raytracetorch_amd import kernels as K
    torch.manual_seed(3)
    lens, sensor = scenes.spot_system(ns, grads=True)
    scene = ns.SequentialScene([lens, sensor]).to(DEV)
    src = scenes.disk_bundle(ns).sample(20001).to(DEV)

    class Resident(torch.nn.Module):
        def sample(self, _n):
            return src.clone()
    loss = ns.SpotSizeLoss(sensor, [Resident()], N_rays=len(src))(scene)
    loss.backward()
    assert type(scene.rays) is SampledRays and len(scene.rays) == len(src)
    grads = [p.grad.clone() for p in scene.parameters() if p.grad is not None]
    locs, w, _ = sensor.getHitsTensors()
    # the same through the plain differentiable trace
    ref_scene = ns.SequentialScene([lens, sensor])
    for p in scene.parameters():
        p.grad = None
    sensor.reset()
    traced = ref_scene.simulate(src.clone())
    # (the trace that keeps a tape is another instantiation of the kernel than the plain trace that makes `scene.rays`:
    # the same formulas, but the compiler may contract a product-sum of a rarely taken branch differently, which shows
    # as one unit in the last place on a few rays in ten thousand)
    def same(a, b):
        return float((a != b).float().mean()) <= 1e-3 and float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
    assert same(scene.rays._pos, traced._pos.detach()) and type(scene.rays) is Rays
    assert same(scene.rays._dir, traced._dir.detach()) and torch.equal(scene.rays.intensity, traced.intensity.detach())
    locs2, w2, _ = sensor.getHitsTensors()
    assert same(locs, locs2.detach()) and torch.equal(w, w2.detach())
    # moments and record of the launch that leaves the rays unwritten equal those of the launch that writes them
    prog = scene._program()
    outs = []
    for store in (True, False):
        mom = torch.zeros(6, device=DEV, dtype=torch.float64)
        rays = src.clone()
        hit, ww, mask = K.trace_sequential(rays, prog, moments=(0, mom), store=store)
        outs.append((mom.clone(), hit, ww, mask, rays))
    (ma, ha, wa, ka, ra), (mb, hb, wb, kb, rb) = outs
    assert torch.equal(ma, mb) and torch.equal(ka, kb)
    m = ka[0].bool()
    assert torch.equal(ha[0][:, m], hb[0][:, m]) and torch.equal(wa[0][m], wb[0][m])
    assert torch.equal(rb._pos, src._pos) and not torch.equal(ra._pos, src._pos)
    assert len(grads) >= 4


def test_second_device_in_one_process(ns):
    """The grant of dynamic LDS above 64 KB is a property of a device's copy of the kernel: a process that sweeps on
    cuda:0 and then on cuda:1 must get it on both (skipped on a one-GPU box)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two devices")
    lib = N.lib()
    lens, sensor = scenes.spot_system(ns, grads=False)
    prog = P.compile_elements([lens, sensor])
    masks = np.zeros(4, dtype=np.uint32)
    masks[0] = masks[1] = np.uint32((1 << 23) | (1 << 24))
    outs = []
    assert lib.rtt_tune(7, 1) == 0     # full states in LDS: about 100 KB for this program
    try:
        for dev in ("cuda:0", "cuda:1"):
            n = 20001
            rays = scenes.disk_bundle(ns, radius=4.0, z=-30.0).sample(n).to(dev)
            adj = torch.ones(3, n, device=dev)
            dp = N.DeviceProgram.upload(prog, dev)
            slot_adj = torch.empty(4, dtype=torch.float64, device=dev)
            ws = torch.empty(int(lib.rtt_trace_seq_adjoint_workspace_bytes(4)), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                code = lib.rtt_trace_seq_adjoint_f32(N.ptr(rays._pos), N.ptr(rays._dir), N.ptr(rays.intensity), n, n, N.ptr(dp.dev),
                                                     dp.host_ptr, dp.n, masks.ctypes.data, N.ptr(adj), 0, n, 0, n, -1, 0, 0.0, 0.0,
                                                     1.0, 0, 0, 0, N.ptr(slot_adj), N.ptr(ws), ws.numel(), N.stream_handle(dev))
            N.check(code, "rtt_trace_seq_adjoint_f32 on " + dev)
            torch.cuda.synchronize(dev)
            outs.append(slot_adj.cpu())
    finally:
        lib.rtt_tune(7, 0)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("state", [0])
def test_state_placements_agree(ns, state):
    """Compact states in LDS (the default for short programs) and full states in LDS sweep the same formulas per ray.  Each placement is its own instantiation of the kernel, in which the compiler may contract
    product-sums differently, so agreement is required to float32 rounding, not bit for bit."""
    lib = N.lib()
    lens, sensor = scenes.spot_system(ns, grads=False)
    prog = P.compile_elements([lens, sensor])
    n = 300007
    rays = scenes.disk_bundle(ns, radius=5.2, z=-30.0, rot=[0.01, 0.02, 0.0]).sample(n).to(DEV)   # some rays miss the lens
    dp = N.DeviceProgram.upload(prog, DEV)
    g = torch.Generator().manual_seed(9)
    adj_pos, adj_dir = (torch.randn(3, n, generator=g).to(DEV) for _ in range(2))
    adj_sens = torch.randn(1, 3, n, generator=g).to(DEV)
    pairs = [(0, 23), (0, 24), (0, 26), (0, 30), (1, 23), (1, 24), (1, 31), (2, 24), (3, 21), (3, 23)]
    masks = np.zeros(4, dtype=np.uint32)
    for s, j in pairs:
        masks[s] |= np.uint32(1 << j)

    def run(which, lens_set=1):
        assert lib.rtt_tune(7, which) == 0 and lib.rtt_tune(11, lens_set) == 0
        try:
            slot_adj = torch.empty(len(pairs), dtype=torch.float64, device=DEV)
            ray_adj = torch.empty((6, n), device=DEV)
            ws = torch.empty(int(lib.rtt_trace_seq_adjoint_workspace_bytes(len(pairs))), dtype=torch.uint8, device=DEV)
            code = lib.rtt_trace_seq_adjoint_f32(N.ptr(rays._pos), N.ptr(rays._dir), N.ptr(rays.intensity), n, n, N.ptr(dp.dev),
                                                 dp.host_ptr, dp.n, masks.ctypes.data, N.ptr(adj_pos), N.ptr(adj_dir), n, N.ptr(adj_sens), n,
                                                 -1, 0, 0.0, 0.0, 1.0, 0, N.ptr(ray_adj), n, N.ptr(slot_adj), N.ptr(ws), ws.numel(),
                                                 N.stream_handle(DEV))
            N.check(code, "rtt_trace_seq_adjoint_f32")
            torch.cuda.synchronize()
            return slot_adj.cpu(), ray_adj.cpu()
        finally:
            lib.rtt_tune(7, 0)
            lib.rtt_tune(11, 1)
    want, got = run(1), run(state)
    worst = float((got[1] - want[1]).abs().max()) / float(want[1].abs().max())
    assert worst <= 1e-5, worst
    assert float((got[0] - want[0]).abs().max()) <= 1e-5 * float(want[0].abs().max())
    # the instantiations that hold only the visits of a lens system (this program is one) against the general ones
    for which in (0, 1):
        lean, general = run(which, 1), run(which, 0)
        assert float((lean[1] - general[1]).abs().max()) <= 1e-5 * float(general[1].abs().max())
        assert float((lean[0] - general[0]).abs().max()) <= 1e-5 * float(general[0].abs().max())
