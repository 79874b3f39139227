"""Sharding of a ray bundle over the GPUs of one node (one process per GPU).

Rays are independent: every rank traces a contiguous slice of the bundle with a replica of the
surface program and no exchange.  The only collectives of the whole path are sums of a handful of
float64 scalars — the spot-size moments [W, sum w x, sum w y], the loss terms and, under autograd, the
parameter-gradient vector — over RCCL (backend "nccl" on ROCm) or gloo in CPU tests.
"""
import torch


def shard_bounds(n: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of `n` rays owned by `rank`; sizes differ by at most one ray."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


# While somebody measures (the benchmark), a list that receives one entry per all-reduce: a pair of CUDA events around
# a collective on device tensors over RCCL, or the host seconds of one that went through the host.
collective_times = None


def allreduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over the ranks of `group`; identity when `group` is None."""
    if group is None:
        return t
    import torch.distributed as dist
    log = collective_times
    if t.is_cuda and dist.get_backend(group) == "gloo":
        # gloo is the rehearsal backend (two ranks on one GPU, CPU tests); its own staging of device tensors has
        # been seen to hang when two processes share a GPU, and these sums are a few doubles: reduce them on the host
        import time
        t0 = time.perf_counter()
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
        if log is not None:
            log.append(time.perf_counter() - t0)
        return t
    if log is not None and t.is_cuda:
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        end.record()
        log.append((start, end))
        return t
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def collective_ms(log) -> float:
    """Sum of the entries `allreduce_sum_` left in `log`, in milliseconds (the device must have finished them)."""
    return sum(e[0].elapsed_time(e[1]) if isinstance(e, tuple) else e * 1e3 for e in log)


def gather_sensor_hits(sensor, group=None, ray_id=None):
    """`Sensor.getHitsTensors` over all ranks of `group` (elements/sensor.py:46-65): every rank gets the hit lists
    of all ranks, concatenated in rank order, i.e. in the order of the contiguous ray ranges the ranks own.  The
    lists have different lengths per rank, so sizes are exchanged first and the rows travel padded to the longest;
    meant for inspection at small sizes, the optimisation path exchanges only sums."""
    locs, inten, ids = sensor.getHitsTensors(ray_id) if sensor.hitLocs else (
        torch.zeros((0, 3)), torch.zeros((0,)), torch.zeros((0,), dtype=torch.int8))
    if group is None:
        return locs, inten, ids
    import torch.distributed as dist
    world = dist.get_world_size(group)
    on_host = dist.get_backend(group) == "gloo"
    dev = torch.device("cpu") if on_host else locs.device
    count = torch.tensor([locs.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    counts = [int(c) for c in counts]
    longest = max(counts)
    # one padded [longest, 5] float32 block per rank: x, y, z, intensity, id
    mine = torch.zeros((longest, 5), dtype=torch.float32, device=dev)
    if locs.shape[0]:
        mine[:locs.shape[0], :3] = locs.detach().to(dev, torch.float32)
        mine[:locs.shape[0], 3] = inten.detach().to(dev, torch.float32)
        mine[:locs.shape[0], 4] = ids.to(dev, torch.float32)
    blocks = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(blocks, mine, group=group)
    rows = torch.cat([b[:c] for b, c in zip(blocks, counts)]).to(locs.device)
    return rows[:, :3].contiguous(), rows[:, 3].contiguous(), rows[:, 4].to(torch.int8)
