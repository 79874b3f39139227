"""GPU, two processes on one card (gloo for the exchange): the fused spot-size loss of a bundle sharded over two ranks
— trace, moments, tangent trace per rank, sums of a few doubles between them — equals the single-process loss and
gradients of the whole bundle.  Also with a bundle of one ray, which leaves the second rank an empty shard."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import scenes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _loss_and_grads(ns, rays, group):
    lens, sensor = scenes.spot_system(ns, grads=True)
    scene = ns.SequentialScene([lens, sensor])
    scene.to(DEV)
    if group is not None:
        scene.grad_group = group

    class Fixed(torch.nn.Module):
        def sample(self, _n):
            return rays

    loss = ns.SpotSizeLoss(sensor, [Fixed()], N_rays=len(rays))(scene)
    loss.backward()
    grads = {k: p.grad.detach().cpu().double().reshape(-1) for k, p in scene.named_parameters() if p.grad is not None}
    return float(loss.detach()), grads


def _bundle(ns, n):
    torch.manual_seed(5)
    return scenes.disk_bundle(ns, radius=4.0, z=-30.0, rot=[0.01, -0.02, 0.0]).sample(n)


def _shard(ns, rays, lo, hi):
    return ns.Rays.from_planes(rays._pos[:, lo:hi].contiguous(), rays._dir[:, lo:hi].contiguous(),
                               rays.intensity[lo:hi].contiguous(), rays.id[lo:hi].contiguous(),
                               rays.wavelength[lo:hi].contiguous()).to(DEV)


def _worker(rank, world, port, n, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import raytracetorch_amd as rta
        from raytracetorch_amd import sharding
        ns = scenes.namespace_from_package(rta)
        lo, hi = sharding.shard_bounds(n, rank, world)
        loss, grads = _loss_and_grads(ns, _shard(ns, _bundle(ns, n), lo, hi), dist.group.WORLD)
        ret[rank] = (loss, {k: v.tolist() for k, v in grads.items()})
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("n", [20001, 1])
def test_two_ranks_give_the_single_process_loss_and_gradients(ns, n):
    want_loss, want = _loss_and_grads(ns, _shard(ns, _bundle(ns, n), 0, n), None)
    assert want and (n == 1 or want_loss > 0)
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, _free_port(), n, ret), nprocs=2, join=True)
    assert len(ret) == 2
    for rank in range(2):
        loss, grads = ret[rank]
        assert abs(loss - want_loss) <= 1e-5 * max(abs(want_loss), 1e-6), (rank, loss, want_loss)
        assert set(grads) == set(want)
        for k, g in grads.items():
            g = torch.tensor(g, dtype=torch.float64)
            scale = max(float(want[k].abs().max()), 1e-9)
            assert float((g - want[k]).abs().max()) <= 2e-4 * scale, (rank, k, g, want[k])
