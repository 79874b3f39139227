"""Ray sources (rays/bundle.py of the reference).

CPU: the oracle's sampling formulas reproduce ``Bundle.sample`` of the unmodified reference when fed
with the torch.rand values the reference consumed (goldens ``bundle_*.npz``); the oracle's Philox4x32-10
matches the Random123 known answers.  GPU: the device sampler (``rtt_sample_bundle_f32`` through
``Bundle.sample``) matches the oracle fed with the same Philox uniforms within 1e-5 (relative to the
vector length), its values do not depend on how a bundle is split into calls, and its distribution
agrees with the reference's sample.
"""
import numpy as np
import pytest
import torch

import rtt_oracle as O
import scenes
from util import golden, rel_err

CASES = ["pencil", "disk", "disk_unposed", "line", "fan", "point"]


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds, counter words (c0, c1, 0, 0)
    got = O.philox4x32_10(np.array([0], dtype=np.uint64), 0)[0]
    assert [int(v) for v in got] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    ua, ub = O.philox_uniforms(seed=7, offset=3, n=6)
    ua2, ub2 = O.philox_uniforms(seed=7, offset=0, n=9)
    assert np.array_equal(ua, ua2[3:]) and np.array_equal(ub, ub2[3:])
    assert ua.dtype == np.float32 and (ua >= 0).all() and (ua < 1).all()


@pytest.mark.parametrize("name", CASES)
def test_oracle_sampler_matches_reference(name):
    g = golden("bundle_" + name)
    pos, d = O.sample_bundle(int(g["kind"]), g["par"], g["R"], g["t"], g["ua"], g["ub"])
    assert rel_err(pos.numpy(), g["pos"]) < 2e-6
    assert rel_err(d.numpy(), g["dir"]) < 2e-6
    assert (g["intensity"] == 1).all() and (g["wavelength"] == 0).all() and (g["id"] == g["ray_id"]).all()


@pytest.mark.parametrize("name", CASES)
def test_package_cpu_sampler_has_reference_distribution(ns, name):
    """The package's CPU path (torch ops, like the reference) consumes torch.rand in the reference's order."""
    g = golden("bundle_" + name)
    b = scenes.bundle_cases(ns)[name][0]()
    torch.manual_seed(0)
    rays = b.sample(4096)
    assert rel_err(rays.pos.numpy(), g["pos"]) < 2e-6
    assert rel_err(rays.dir.numpy(), g["dir"]) < 2e-6
    assert rays.id.dtype == torch.int8 and (rays.id.numpy() == g["ray_id"]).all()


def _device_bundle(ns, name):
    mk, kind, par_of, draws = scenes.bundle_cases(ns)[name]
    b = mk()
    b.device = "cuda:0"
    b.transform.to("cuda:0")
    return b, kind, par_of(b)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("n,offset", [(4096, 0), (4097, 0), (1000, 7), (1, 1), (0, 0)])
def test_device_sampler_matches_oracle(ns, name, n, offset):
    b, kind, par = _device_bundle(ns, name)
    rays = b.sample(n, offset=offset, seed=1234)
    assert rays._pos.is_cuda and tuple(rays._pos.shape) == (3, n) and rays.id.dtype == torch.int8
    ua, ub = O.philox_uniforms(1234, offset, n)
    pos, d = O.sample_bundle(kind, par, b.transform.rot.cpu(), b.transform.trans.detach().cpu(), ua, ub)
    assert rel_err(rays.pos.cpu().numpy(), pos.numpy()) < 1e-5   # tolerance of the parity contract
    assert rel_err(rays.dir.cpu().numpy(), d.numpy()) < 1e-5
    assert (rays.intensity == 1).all() and (rays.wavelength == 0).all() and (rays.id == b.ray_id).all()


@pytest.mark.gpu
def test_device_sampler_is_split_invariant_and_seeded(ns):
    b, _, _ = _device_bundle(ns, "disk")
    whole = b.sample(10001, offset=0, seed=5)
    parts = [b.sample(m, offset=o, seed=5) for o, m in ((0, 3333), (3333, 3334), (6667, 3334))]
    assert torch.equal(torch.cat([p._pos for p in parts], dim=1), whole._pos)
    assert torch.equal(torch.cat([p._dir for p in parts], dim=1), whole._dir)
    torch.manual_seed(11)
    a1, a2 = b.sample(512), b.sample(512)
    torch.manual_seed(11)
    b1 = b.sample(512)
    assert torch.equal(a1._pos, b1._pos) and not torch.equal(a1._pos, a2._pos)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["disk", "point", "fan", "line"])
def test_device_sampler_distribution_matches_reference(ns, name):
    """Statistical parity with the reference's own sample (different generators): first and second
    moments of positions and directions agree within sampling error."""
    g = golden("bundle_" + name)
    b, _, _ = _device_bundle(ns, name)
    rays = b.sample(1 << 20, offset=0, seed=99)
    for got, ref in ((rays.pos.cpu().numpy(), g["pos"]), (rays.dir.cpu().numpy(), g["dir"])):
        got, ref = got.astype(np.float64), ref.astype(np.float64)
        sd = ref.std(0) + 1e-6
        assert (np.abs(got.mean(0) - ref.mean(0)) < 6 * sd / np.sqrt(len(ref)) + 1e-5).all()
        assert (np.abs(got.std(0) - ref.std(0)) < 0.06 * sd + 1e-5).all()


@pytest.mark.gpu
def test_device_sampler_feeds_the_trace(ns):
    lens = scenes.singlet(ns)
    scene = ns.SequentialScene([lens])
    b, kind, par = _device_bundle(ns, "disk_unposed")
    b.transform.trans.data[2] = -30.0
    rays = b.sample(1 << 16, offset=0, seed=3)
    src = rays.clone()
    out = scene.simulate(rays)
    ref = O.trace_sequential(scene._program(), src.pos.cpu(), src.dir.cpu(), src.intensity.cpu())
    assert rel_err(out.pos.cpu().numpy(), ref["pos"].numpy()) < 1e-5
    assert rel_err(out.dir.cpu().numpy(), ref["dir"].numpy()) < 1e-5
