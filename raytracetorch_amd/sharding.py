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


def allreduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over the ranks of `group`; identity when `group` is None."""
    if group is None:
        return t
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend(group) == "gloo":
        # gloo is the rehearsal backend (two ranks on one GPU, CPU tests); its own staging of device tensors has
        # been seen to hang when two processes share a GPU, and these sums are a few doubles: reduce them on the host
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
        return t
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
