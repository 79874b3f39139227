"""Benchmark of the sequential singlet-lens trace (the headline workload) on N GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one `SequentialScene.simulate` of the singlet scene over one resident bundle of `--rays`
rays per GPU (default 64 * 2^20): one kernel launch that tests 3 surfaces per ray.  Rays are
sharded across ranks (each rank owns its own bundle, seeded by rank); the trace has no exchange
step, so there is no data-path collective and scaling is weak.  Rank 0 prints ONE JSON line.

Started without a launcher (`WORLD_SIZE` unset) and with `--gpus N` > 1, this process only starts N
fresh rank processes (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), waits for them and
exits with the first non-zero status; it never touches the GPU itself.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

LIB_PATH = os.environ.get("RTT_HIP_LIB") or os.path.join(ROOT, "raytracetorch_amd", "csrc", "librtt_hip.so")
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
BYTES_PER_RAY = 56             # read pos+dir+intensity (28 B) + write the same (28 B), surface-count independent
SURFACES_TESTED = 3            # front cap, back cap, edge cylinder


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=64 * 2 ** 20, help="rays per GPU")
    ap.add_argument("--cpu-rays", type=int, default=2 ** 24, help="rays of the CPU-baseline sample (about 10 s of host work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="singlet", choices=["singlet", "system5", "naive32", "naive_sim", "spot", "sample_trace"],
                    help="singlet = the headline line; the others time the remaining configurations (informational)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo lets two "
                                                      "ranks rehearse the multi-rank path on one GPU)")
    ap.add_argument("--dist-single", action="store_true",
                    help="with --gpus 1: still form a process group of one rank over --backend, so that the collective "
                         "calls of the N > 1 path (RCCL barrier, all-reduce) execute on a one-GPU box")
    ap.add_argument("--step-loop", action="store_true",
                    help="naive_sim workload: one launch per bounce (the reference's loop) instead of the fused bounce loop")
    ap.add_argument("--two-launches", action="store_true",
                    help="sample_trace workload: generate the bundle with the sampling kernel, then trace it (two launches) "
                         "instead of the launch that generates and traces")
    ap.add_argument("--graph", action="store_true",
                    help="singlet workload: capture one simulate() in a HIP graph and replay it per step (launch-bound sizes)")
    ap.add_argument("--tune-rpt", type=int, default=0, help="override rays per lane (1, 2, 4) for experiments")
    ap.add_argument("--tune-lds", type=int, default=-1, help="1: surface program staged in LDS (experiment)")
    ap.add_argument("--tune-blocks", type=int, default=0, help="override the grid cap for experiments")
    ap.add_argument("--tune-reduce-blocks", type=int, default=0, help="override the grid of the reducing kernels (experiment)")
    ap.add_argument("--tune-cull", type=int, default=-1, help="0: nearest-hit kernels test every surface (experiment)")
    ap.add_argument("--tune-adjoint-lane", type=int, default=0, help="rays per lane of the reverse-mode backward: 1, 2, 0 = automatic")
    ap.add_argument("--tune-adjoint-rounds", type=int, default=0, help="rounds of resident blocks in the grid of the reverse-mode backward (experiment)")
    ap.add_argument("--tune-lens-set", type=int, default=-1, help="0: the reverse sweep of a lens-only program runs the instantiation with every visit (experiment)")
    ap.add_argument("--tune-lean-bounces", type=int, default=-1, help="0: the bounce loops always run the instantiation with the generic visit (experiment)")
    ap.add_argument("--tune-moments-blocks", type=int, default=0, help="override the grid of the spot-moment passes (experiment)")
    ap.add_argument("--tune-late-records", type=int, default=-1, help="0: every stage of the staged bounce loop writes the output planes itself (experiment)")
    ap.add_argument("--tune-adjoint-state", type=int, default=-1, help="1: the reverse-mode backward keeps its states in LDS also for short programs")
    return ap.parse_args()


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def child_envs(n_ranks, port, base=None):
    """Environment of each rank process of a one-node run: what `torch.distributed.run` would set."""
    base = dict(os.environ if base is None else base)
    envs = []
    for r in range(n_ranks):
        e = dict(base)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        envs.append(e)
    return envs


def launch_ranks(n_ranks, argv, timeout=None):
    """Start one fresh process per rank and wait for all of them.  The caller has not initialised the GPU and this
    function does not either: the ranks are children (never a replacement of this process), rank 0 inherits stdout
    and prints the JSON line.  Returns the first non-zero exit status (0 when every rank succeeded); when one rank
    fails the others are ended, by their own process ids."""
    import subprocess
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e)
             for e in child_envs(n_ranks, free_port())]
    deadline = None if timeout is None else time.monotonic() + timeout
    status, pending = 0, list(procs)
    while pending:
        time.sleep(0.05)
        for p in list(pending):
            rc = p.poll()
            if rc is None:
                continue
            pending.remove(p)
            if rc != 0 and status == 0:
                status = rc
        if pending and (status != 0 or (deadline is not None and time.monotonic() > deadline)):
            if status == 0:
                status = 124
            for p in pending:
                p.terminate()
            for p in pending:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            pending = []
    return status


def max_over_ranks(value, dist, dev):
    """MAX of a host scalar over all ranks (the timed region ends when the slowest rank does).  Under gloo the
    value travels as a host tensor: that backend is the rehearsal one (ranks may share a GPU) and needs no device."""
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def lib_sha16():
    """First 16 hex digits of the SHA-256 of the loaded kernel library: ties a committed counter profile to a build."""
    import hashlib
    try:
        return hashlib.sha256(open(LIB_PATH, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def device_bundle(n, device, seed, offset=0, ns=None):
    """CollimatedDisk(radius=4) at z=-30 along +z as [3,N] planes.  On a HIP device the rays come from the
    product's device sampler (Philox stream `seed`, global ray indices offset .. offset+n-1, so the shards of
    all ranks form one bundle); the CPU sample for the baseline uses torch ops."""
    import math
    if ns is not None and torch.device(device).type == "cuda":
        b = ns.CollimatedDisk(radius=4.0, ray_id=0, device=device,
                              transform=ns.RayTransformBundle(translation=[0.0, 0.0, -30.0]).to(device))
        r = b.sample(n, offset=offset, seed=seed)
        return r._pos, r._dir
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    theta = torch.rand(n, device=device, generator=g) * (2 * math.pi)
    r = torch.sqrt(torch.rand(n, device=device, generator=g) * 16.0)
    pos = torch.empty((3, n), device=device)
    torch.mul(r, torch.cos(theta), out=pos[0])
    torch.mul(r, torch.sin(theta), out=pos[1])
    pos[2].fill_(-30.0)
    del theta, r
    d = torch.zeros((3, n), device=device)
    d[2].fill_(1.0)
    return pos, d


def profile_mode(workload):
    """What, beside workload, ray count and library, decides which kernels a step launches: the backward mode of the
    spot workload (reverse-mode or forward-mode kernels).  A counter profile describes one mode only."""
    return os.environ.get("RTT_BACKWARD", "adjoint") if workload == "spot" else None


def matching_profiles(workload, n):
    """Committed traffic files of this workload, newest round first, each with the reason it does not describe the run
    that is being timed (None when it does): other ray count, other build of the library, other backward mode."""
    import glob
    sha, mode = lib_sha16(), profile_mode(workload)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", f"traffic_{workload}.json")), reverse=True):
        d = json.load(open(f))
        name = os.path.relpath(f, ROOT)
        why = None
        if d.get("rays_per_gpu") != n:
            why = f"{name} was taken at {d.get('rays_per_gpu')} rays"
        elif d.get("lib_sha16") != sha:
            why = f"{name} was taken with another build of the library ({d.get('lib_sha16')} != {sha})"
        elif (d.get("backward_mode") or ("adjoint" if workload == "spot" else None)) != mode:
            why = f"{name} was taken with RTT_BACKWARD={d.get('backward_mode') or 'adjoint'}, this run uses {mode}"
        yield name, d, why


def measured_traffic(workload, n):
    """(HBM bytes per step, note) from the committed rocprofv3 PMC passes of this same command.  The figure is
    attached only when the profile was taken with the very library that is loaded now (SHA-256 recorded in the
    file), with the same ray count and in the same backward mode; otherwise (None, reason)."""
    reason = f"no committed profile of workload {workload}"
    for name, d, why in matching_profiles(workload, n):
        if why is None:
            return d["traffic_bytes_per_launch"], name
        reason = why
    return None, reason


def measured_issue(workload, n):
    """Issue-side counters of the step's longest kernel from the same committed profile (same validity rule as the
    traffic): share of the kernel's cycles in which the vector / scalar unit of a SIMD issues, and vector instructions
    per ray.  For the kernels that are bound by instruction issue and latency rather than by HBM this, not the HBM
    fraction, says how close they run to their bound.  None when there is no profile of the loaded build."""
    for source, d, why in matching_profiles(workload, n):
        if why is not None or not d.get("kernels"):
            continue
        name, k = max(d["kernels"].items(), key=lambda kv: kv[1].get("avg_us", 0.0) * kv[1].get("launches_per_step", 1.0))
        if "valu_busy" not in k:
            return None
        return {"kernel": name[:48], "kernel_us": k.get("avg_us"), "valu_busy": k.get("valu_busy"), "salu_busy": k.get("salu_busy"),
                "valu_insts_per_ray": k.get("valu_insts_per_ray"), "source": source}
    return None


def timing_stats(ms):
    return {"kernel_ms": statistics.mean(ms), "kernel_ms_min": min(ms), "kernel_ms_median": statistics.median(ms),
            "kernel_ms_max": max(ms)}


def cpu_baseline(n, scenes, ns):
    """The CPU oracle (a torch-CPU port of the reference's per-ray arithmetic) on a bounded sample."""
    import rtt_oracle as O
    import raytracetorch_amd.program as P
    prog = P.compile_elements([scenes.singlet(ns)])
    pos, d = device_bundle(n, "cpu", 0)
    pos, d = pos.t().contiguous(), d.t().contiguous()
    inten = torch.ones(n)
    times = []
    chunk = 2 ** 20   # cache-sized pieces: the torch-CPU port is several times faster than on one 16 Mi-ray batch
    with torch.no_grad():
        O.trace_sequential(prog, pos[:65536], d[:65536], inten[:65536])
        for _ in range(5):
            t0 = time.perf_counter()
            for a in range(0, n, chunk):
                O.trace_sequential(prog, pos[a:a + chunk], d[a:a + chunk], inten[a:a + chunk])
            times.append(time.perf_counter() - t0)
    t = statistics.median(times)
    return {"value": n * SURFACES_TESTED / t, "unit": "ray-surface intersections/s",
            "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/rtt_oracle.py trace_sequential, singlet (3 surfaces), {n} rays in 2^20-ray chunks, median of 5, "
                      f"{t * 1e3:.0f} ms per trace"}


def cpu_baseline_other(workload, scenes, ns):
    """The CPU oracle on a bounded sample of the 5-element system (2^20 rays) or the 32-surface nearest-hit step
    (2^18 rays); None for workloads the oracle has no single call for."""
    import rtt_oracle as O
    import raytracetorch_amd.program as P
    if workload == "system5":
        n, surfaces, prog, z0 = 2 ** 20, 9, P.compile_elements(scenes.system5(ns)), -30.0
        run = lambda p, q, w: O.trace_sequential(prog, p, q, w)
        what = "trace_sequential, 5-element system (9 surfaces)"
    elif workload == "naive32":
        n, surfaces, prog, z0 = 2 ** 18, 32, P.compile_elements(scenes.naive32(ns)), -10.0
        run = lambda p, q, w: O.step_nearest(prog, p, q, w)
        what = "step_nearest, 32-surface scene (every ray tests every surface)"
    else:
        return None
    pos, d = device_bundle(n, "cpu", 0)
    pos[2].fill_(z0)
    pos, d, inten = pos.t().contiguous(), d.t().contiguous(), torch.ones(n)
    times = []
    with torch.no_grad():
        run(pos[:65536], d[:65536], inten[:65536])
        for _ in range(5):
            t0 = time.perf_counter()
            run(pos, d, inten)
            times.append(time.perf_counter() - t0)
    t = statistics.median(times)
    return {"value": n * surfaces / t, "unit": "ray-surface intersections/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"oracle/rtt_oracle.py {what}, {n} rays, median of 5, {t * 1e3:.0f} ms per pass"}


def other_workload(a, ns, scenes, dev, rank, world, dist):
    """Configurations 3-5: 5-element system (16 Mi rays), 32-surface nearest-hit step (4 Mi rays), spot-size
    loss forward+backward (singlet + sensor).  Same timing discipline as the headline line."""
    defaults = {"system5": 16 * 2 ** 20, "naive32": 4 * 2 ** 20, "spot": 32 * 2 ** 20, "sample_trace": 64 * 2 ** 20,
                "naive_sim": 4 * 2 ** 20}
    n = a.rays if a.rays != 64 * 2 ** 20 else defaults[a.workload]
    pos, d = device_bundle(n, dev, 1000, offset=rank * n, ns=ns)
    mk = lambda p, q: ns.Rays.from_planes(p, q, torch.ones(n, device=dev), torch.zeros(n, device=dev, dtype=torch.int8),
                                          torch.zeros(n, device=dev))
    rays = mk(pos, d)
    kernel = "k_trace_seq"
    pre = None   # untimed preparation of a step, where the step consumes its input
    extra_bytes = None   # algorithmic bytes of a step beyond bytes_per_ray * n, known once a step has run
    if a.workload == "system5":
        elements = scenes.system5(ns)
        scene = ns.SequentialScene(elements)
        out = mk(torch.empty_like(pos), torch.empty_like(d))
        surfaces = 9

        def step():
            elements[-1].reset()
            scene.simulate(rays, out=out)
        bytes_per_ray = 56 + 1 + 17  # state in/out + sensor mask + (upper bound) one recorded hit
    elif a.workload == "sample_trace":
        # end to end: CollimatedDisk.sample(n) on the device (write-only kernel) + in-place singlet trace
        scene = ns.SequentialScene([scenes.singlet(ns)])
        bundle = ns.CollimatedDisk(radius=4.0, ray_id=0, device=dev,
                                   transform=ns.RayTransformBundle(translation=[0.0, 0.0, -30.0]).to(dev))
        surfaces = 3
        kernel = "k_sample_bundle + k_trace_seq" if a.two_launches else "k_trace_seq<generate>"
        del rays, pos, d

        def step():
            fresh = bundle.sample(n, offset=rank * n, seed=7)
            if a.two_launches:
                fresh._force()   # (reading the sample generates it)
            scene.simulate(fresh)
        # written: traced planes, intensity, id, wavelength (33 B); two launches also write and re-read the sample
        bytes_per_ray = 33 + 56 if a.two_launches else 33
    elif a.workload == "naive_sim":
        # Scene.simulate() of the 32-surface scene with the reference's default of 100 bounces
        scene = ns.Scene()
        for el in scenes.naive32(ns):
            scene.add_element(el)
        rays._pos[2].fill_(-10.0)
        surfaces = 32
        kernel = "k_simulate_nearest (launch per bounce)" if a.step_loop else "k_simulate_nearest (fused bounce loop)"

        def step():
            scene.elements[-1].reset()
            scene.rays = rays.clone()
            if a.step_loop:
                scene._build_index_maps()
                for _ in range(scene.Nbounces):
                    if not (scene.rays.intensity > 0).any():
                        break
                    scene.step()
            else:
                scene.simulate()
        bytes_per_ray = 56

        def extra_bytes():   # every crossing of the sensor is appended to its hit list: 16 B of hit + 8 B of key
            return 24 * sum(int(x.shape[0]) for x in scene.elements[-1].hitLocs)
    elif a.workload == "naive32":
        scene = ns.Scene()
        for el in scenes.naive32(ns):
            scene.add_element(el)
        rays._pos[2].fill_(-10.0)
        work = rays.clone()
        surfaces = 32
        kernel = "k_simulate_nearest (one bounce)"

        def pre():   # (the bounce works in place: start every step from the same rays; not part of the bounce)
            work._pos.copy_(rays._pos); work._dir.copy_(rays._dir); work.intensity.copy_(rays.intensity)
            scene.elements[-1].reset()
            scene.rays = work

        def step():
            scene.step()
        bytes_per_ray = 56 + 1
    else:
        lens, sensor = scenes.spot_system(ns, grads=True)
        scene = ns.SequentialScene([lens, sensor])
        scene.to(dev)   # trainable parameters resident on the GPU (what project.load_scene leaves): the step never waits for the host
        if dist is not None:
            scene.grad_group = dist.group.WORLD
        surfaces = 4
        kernel = {"tape": "k_trace_seq<moments, tape> + k_sum_partials + k_spot_moments + k_trace_seq_adjoint<tape> + k_sum_partials",
                  "adjoint": "k_trace_seq<moments> + k_sum_partials + k_spot_moments + k_trace_seq_adjoint + k_sum_partials",
                  "tangent": "k_trace_seq + k_spot_moments x2 + k_trace_seq_tangent<6, spot>"}[os.environ.get("RTT_BACKWARD", "adjoint")]

        class _Resident(torch.nn.Module):
            """Bundle stand-in for SpotSizeLoss: hands out the resident synthetic rays."""
            def sample(self, _n):
                return ns.Rays.from_planes(rays._pos, rays._dir, rays.intensity, rays.id, rays.wavelength)
        loss_fn = ns.SpotSizeLoss(sensor, [_Resident()], N_rays=n)

        def step():
            for p in scene.parameters():
                p.grad = None
            loss_fn(scene).backward()
        bytes_per_ray = 108

    def barrier():
        if dist is not None:
            dist.barrier()

    prepare = pre
    for _ in range(a.warmup):
        if prepare:
            prepare()
        step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    import raytracetorch_amd.sharding as sharding
    collectives = sharding.collective_times = [] if (a.workload == "spot" and dist is not None) else None
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    idle = 0.0
    for s, e in ev:
        if prepare:   # untimed: the wall clock is stopped while the input of the next step is restored
            torch.cuda.synchronize(); t1 = time.perf_counter()
            prepare()
            torch.cuda.synchronize(); idle += time.perf_counter() - t1
        s.record(); step(); e.record()
    torch.cuda.synchronize(); barrier()
    elapsed = time.perf_counter() - t0 - idle
    elapsed = max_over_ranks(elapsed, dist, dev)
    sharding.collective_times = None
    stats = timing_stats([s.elapsed_time(e) for s, e in ev])
    gpu_ms = stats["kernel_ms"]
    if rank == 0:
        algorithmic = bytes_per_ray * n + (extra_bytes() if extra_bytes else 0)
        achieved = algorithmic / (gpu_ms * 1e-3) / 1e9
        traffic, traffic_note = measured_traffic(a.workload, n)
        issue = measured_issue(a.workload, n)
        other_form = "--step-loop" if (a.workload == "naive_sim" and a.step_loop) else (
            "--two-launches" if (a.workload == "sample_trace" and a.two_launches) else None)
        if other_form:   # the committed counters describe the default form of the workload, not this one
            traffic, traffic_note, issue = None, f"the profile of workload {a.workload} was taken without {other_form}", None
        line = {
            "metric": "ray-surface intersections/sec", "value": world * n * surfaces / (elapsed / a.steps),
            "unit": "ray-surface intersections/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed * 1e3 / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": a.workload, "rays_per_gpu": n, "rays_total": world * n, "surfaces": surfaces,
                       "ranks": (dist.get_world_size() if dist is not None else 1), "backend": (dist.get_backend() if dist is not None else None),
                       "parallelism": (f"rays sharded over {world} GPU(s); "
                                       + ("sums of a few float64 (spot moments, loss, parameter gradients) all-reduced"
                                          if a.workload == "spot" else "no data-path collective")),
                       # rank 0's event time inside the all-reduces of one step (it includes waiting for the slowest rank)
                       **({"collective_ms": sharding.collective_ms(collectives) / a.steps,
                           "collectives_per_step": len(collectives) / a.steps} if collectives is not None else {})},
            "rays_per_s": world * n / (elapsed / a.steps),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                         "kernel": kernel, **stats,
                         "algorithmic_bytes_per_launch": algorithmic,
                         "issue": issue,
                         "note": "event time spans every kernel of the step, including torch glue"}}
        base = None if (world > 1 or a.no_cpu_baseline) else cpu_baseline_other(a.workload, scenes, ns)
        if base is not None:
            line["cpu_baseline"] = base
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    a = parse()
    if os.environ.get("RTT_BENCH_WATCHDOG"):   # dump every thread's stack and exit if the run takes longer
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["RTT_BENCH_WATCHDOG"]), exit=True)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # no launcher above us: be the launcher.  Nothing here has touched the GPU (device_count() does not)
        if a.backend == "nccl" and torch.cuda.device_count() < a.gpus:
            raise SystemExit(f"--gpus {a.gpus} over RCCL needs {a.gpus} GPUs, this node shows {torch.cuda.device_count()} "
                             f"(two ranks may share a GPU only with --backend gloo)")
        raise SystemExit(launch_ranks(a.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} does not match WORLD_SIZE={world}: the launcher and the flag must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback for the product path")
    device_index = local_rank % torch.cuda.device_count()   # one rank per GPU on a full node
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    dist = None
    if world > 1 or a.dist_single:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:   # (a single rank started by hand)
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", WORLD_SIZE="1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)

    import raytracetorch_amd as rta
    import scenes
    ns = scenes.namespace_from_package(rta)

    from raytracetorch_amd import _native
    if a.tune_rpt:
        _native.check(_native.lib().rtt_tune(0, a.tune_rpt), "rtt_tune")
    if a.tune_lds >= 0:
        _native.check(_native.lib().rtt_tune(2, a.tune_lds), "rtt_tune")
    if a.tune_blocks:
        _native.check(_native.lib().rtt_tune(1, a.tune_blocks), "rtt_tune")
    if a.tune_reduce_blocks:
        _native.check(_native.lib().rtt_tune(5, a.tune_reduce_blocks), "rtt_tune")
    if a.tune_cull >= 0:
        _native.check(_native.lib().rtt_tune(3, a.tune_cull), "rtt_tune")
    if a.tune_adjoint_lane:
        _native.check(_native.lib().rtt_tune(6, a.tune_adjoint_lane), "rtt_tune")
    if a.tune_lens_set >= 0:
        _native.check(_native.lib().rtt_tune(11, a.tune_lens_set), "rtt_tune")
    if a.tune_lean_bounces >= 0:
        _native.check(_native.lib().rtt_tune(12, a.tune_lean_bounces), "rtt_tune")
    if a.tune_moments_blocks:
        _native.check(_native.lib().rtt_tune(4, a.tune_moments_blocks), "rtt_tune")
    if a.tune_late_records >= 0:
        _native.check(_native.lib().rtt_tune(10, a.tune_late_records), "rtt_tune")
    if a.tune_adjoint_rounds:
        _native.check(_native.lib().rtt_tune(9, a.tune_adjoint_rounds), "rtt_tune")
    if a.tune_adjoint_state >= 0:
        _native.check(_native.lib().rtt_tune(7, a.tune_adjoint_state), "rtt_tune")

    if a.workload != "singlet":
        return other_workload(a, ns, scenes, dev, rank, world, dist)

    n = a.rays
    scene = ns.SequentialScene([scenes.singlet(ns)])
    pos, d = device_bundle(n, dev, 1000, offset=rank * n, ns=ns)
    mk = lambda p, q: ns.Rays.from_planes(p, q, torch.ones(n, device=dev), torch.zeros(n, device=dev, dtype=torch.int8),
                                          torch.zeros(n, device=dev))
    rays = mk(pos, d)
    out = mk(torch.empty_like(pos), torch.empty_like(d))

    def barrier():
        if dist is not None:
            dist.barrier()

    step = lambda: scene.simulate(rays, out=out)
    if a.graph:
        # the launch is capturable: the C side neither allocates nor synchronises and the surface program is
        # already resident after the first call
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        step = graph.replay
    for _ in range(a.warmup - 1):
        step()
    # Launch-bound sizes (a step of a few tens of microseconds): two event records per step would cost as much as the
    # step, so one pair brackets all K steps and only the mean is reported.  The last warm-up step is the one that is
    # clocked for this decision, so the command dispatches exactly warmup + steps launches (with --warmup 0, one more).
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    per_step_events = (time.perf_counter() - t0) > 2e-4
    pairs = a.steps if per_step_events else 1
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(pairs)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if per_step_events:
        for s, e in ev:
            s.record()
            step()
            e.record()
    else:
        ev[0][0].record()
        for _ in range(a.steps):
            step()
        ev[0][1].record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, dist, dev)
    if per_step_events:
        stats = timing_stats([s.elapsed_time(e) for s, e in ev])
    else:
        stats = {"kernel_ms": ev[0][0].elapsed_time(ev[0][1]) / a.steps, "kernel_ms_min": None, "kernel_ms_median": None,
                 "kernel_ms_max": None, "timing_note": "one HIP-event pair around all steps (launch-bound size)"}
    kernel_ms = stats["kernel_ms"]
    alive = int((out.intensity > 0).sum().item())

    if rank == 0:
        ms_per_step = elapsed * 1e3 / a.steps
        value = world * n * SURFACES_TESTED / (elapsed / a.steps)
        achieved = BYTES_PER_RAY * n / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_note = measured_traffic("singlet", n)
        line = {
            "metric": "ray-surface intersections/sec", "value": value, "unit": "ray-surface intersections/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "sequential singlet-lens trace (SingletLens c1=.05 c2=-.05 d=10 t=3 n=1.5 inked; "
                                   "CollimatedDisk r=4 at z=-30), 3 surfaces tested per ray",
                       "rays_per_gpu": n, "rays_total": world * n, "surfaces": SURFACES_TESTED,
                       "parallelism": f"rays sharded over {world} GPU(s), no data-path collective",
                       "ranks": (dist.get_world_size() if dist is not None else 1), "backend": (dist.get_backend() if dist is not None else None),
                       "launch": "hip graph replay" if a.graph else "python launcher",
                       "alive_after_trace_rank0": alive},
            "rays_per_s": world * n / (elapsed / a.steps),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                         "kernel": "k_trace_seq", **stats, "algorithmic_bytes_per_launch": BYTES_PER_RAY * n,
                         "issue": measured_issue("singlet", n)},
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.cpu_rays, scenes, ns)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
