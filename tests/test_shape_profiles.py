"""Bare lens shapes scanned by a row of parallel rays — the constructions of the reference's profile scripts
(tests/test_spherics.py:108-146, tests/test_cylindrics.py:106-122): tensor-valued constructor arguments, a singlet
rotated by 90 degrees, a box rotated by 45 degrees, cemented stacks, cylindrical lenses.  The golden holds the
reference's `Shape.intersectTest` matrix for a scan along x and one along y, and `Shape.forward` of the first two
surfaces.  CPU: the oracle on the records this package packs for the shape.  GPU: the package's own calls."""
import numpy as np
import pytest
import torch

import rtt_oracle as O
import scenes
from util import assert_parity, golden, pattern_mismatch

NAMES = ["biconvex", "biconvex_flipped", "meniscus", "plano", "half_sphere", "doublet", "triplet", "box45",
         "cyl_biconvex", "cyl_plano", "cyl_biconcave"]


def _case(ns, name):
    for nm, shape, extent in scenes.profile_shapes(ns):
        if nm == name:
            return shape, extent
    raise KeyError(name)


def _finite_same(a, b, what):
    assert a.shape == b.shape, what
    assert pattern_mismatch(a, b) == 0, f"{what}: hit/miss pattern differs in {pattern_mismatch(a, b)} entries"


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_the_profile_scan(ns, name):
    g = golden("shape_profiles")
    shape, extent = _case(ns, name)
    prog = shape._program()
    for axis in (0, 1):
        rays = scenes.profile_rays(ns, extent, axis)
        p, d = rays.pos, rays.dir
        tm = O.intersect_matrix(prog, p, d).numpy()
        ref = g[f"{name}_t{axis}"]
        _finite_same(tm, ref, f"{name} axis {axis}")
        tm64 = O.intersect_matrix(prog, p, d, dtype=torch.float64).numpy()
        assert_parity(tm.reshape(-1), ref.reshape(-1), tm64.reshape(-1), what=f"{name} axis {axis} t matrix")
        for j in (0, 1):
            S = O._records(prog[j:j + 1], torch.float32)[0]
            S64 = O._records(prog[j:j + 1], torch.float64)[0]
            t, hg, ng, hl = (x.numpy() for x in O.surface_forward(S, p, d))
            t64, hg64, ng64, hl64 = (x.numpy() for x in O.surface_forward(S64, p.double(), d.double()))
            key = f"{name}_fw{axis}_{j}_"
            _finite_same(t, g[key + "t"], key)
            ok = np.isfinite(t)
            assert_parity(t[ok], g[key + "t"][ok], t64[ok], what=key + "t")
            assert_parity(hg[ok], g[key + "hit"][ok], hg64[ok], what=key + "hit")
            assert_parity(ng[ok], g[key + "n"][ok], ng64[ok], what=key + "n")
            assert_parity(hl[ok], g[key + "local"][ok], hl64[ok], what=key + "local")


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_package_reproduces_the_profile_scan(ns, name):
    g = golden("shape_profiles")
    shape, extent = _case(ns, name)
    prog = shape._program()
    for axis in (0, 1):
        rays = scenes.profile_rays(ns, extent, axis)
        p, d = rays.pos.clone(), rays.dir.clone()
        tm = shape.intersectTest(rays.to("cuda:0")).cpu().numpy()
        ref = g[f"{name}_t{axis}"]
        _finite_same(tm, ref, f"{name} axis {axis}")
        tm64 = O.intersect_matrix(prog, p, d, dtype=torch.float64).numpy()
        assert_parity(tm.reshape(-1), ref.reshape(-1), tm64.reshape(-1), what=f"{name} axis {axis} t matrix")
        for j in (0, 1):
            t, hg, ng, hl = (x.cpu().numpy() for x in shape.forward(rays.to("cuda:0"), j))
            S64 = O._records(prog[j:j + 1], torch.float64)[0]
            t64, hg64, ng64, hl64 = (x.numpy() for x in O.surface_forward(S64, p.double(), d.double()))
            key = f"{name}_fw{axis}_{j}_"
            _finite_same(t, g[key + "t"], key)
            ok = np.isfinite(t)
            assert_parity(t[ok], g[key + "t"][ok], t64[ok], what=key + "t")
            assert_parity(hg[ok], g[key + "hit"][ok], hg64[ok], what=key + "hit")
            assert_parity(ng[ok], g[key + "n"][ok], ng64[ok], what=key + "n")
            assert_parity(hl[ok], g[key + "local"][ok], hl64[ok], what=key + "local")
