"""CPU: the oracle (oracle/rtt_oracle.py), driven by surface programs packed by the product's scene
compiler, reproduces the golden vectors recorded from the unmodified reference."""
import numpy as np
import pytest
import torch

import rtt_oracle as O
import scenes
from util import INF, assert_parity, golden, pattern_mismatch, rel_err

import raytracetorch_amd.program as P

TOL = 5e-6  # float32 restatement vs float32 reference: a few ulp (3x3 products go through BLAS there)


def test_record_layout_matches_oracle():
    assert P.SURFACE_DTYPE == O.SURFACE_DTYPE
    assert P.RECORD_BYTES == 256


@pytest.mark.parametrize("name", [
    "plane_c1", "plane_posed", "sphere", "cylinder", "quadric_parabola", "quadric_hyperbola", "quadric_zy",
    "cone", "single_cone", "half_sphere", "bounded_half_sphere", "half_cyl", "disk_inverted", "rectangle",
    "ellipse", "kat_plane", "kat_sphere", "kat_parabola", "kat_implicit"])
def test_primitive_forward(ns, name):
    g = golden("prim_" + name)
    cases = dict(scenes.primitive_cases(ns)); cases.update(scenes.kat_rays(ns))
    surf = cases[name][0]()
    prog = surf._surface_record()[None]
    S = O._records(prog, torch.float32)[0]
    p, d = torch.from_numpy(g["in_pos"]), torch.from_numpy(g["in_dir"])
    t_test = O.test_column(S, p, d)
    t, hit, nrm, hl = O.surface_forward(S, p, d)
    assert pattern_mismatch(t_test, g["t_test"][:, 0]) == 0
    assert rel_err(t_test, g["t_test"][:, 0]) < TOL
    assert rel_err(t, g["t"]) < TOL
    m = np.isfinite(g["t"])
    for got, ref in ((hit, g["hit_global"]), (nrm, g["normal_global"]), (hl, g["hit_local"])):
        assert rel_err(got.numpy()[m], ref[m]) < TOL


def test_reference_known_answers(ns):
    """tests/test_primitive.py:40-163 of the reference, asserted for real."""
    g = golden("prim_kat_plane")
    assert np.allclose(g["hit_global"][:, 2], 0.0, atol=1e-6) and np.allclose(g["normal_global"], [0, 0, 1])
    g = golden("prim_kat_sphere")
    assert np.isinf(g["t"][2]) and np.allclose(np.linalg.norm(g["hit_global"][:2], axis=1), 10.0, atol=1e-4)
    g = golden("prim_kat_parabola")
    assert abs(g["hit_global"][0, 2] - 0.2) < 1e-5 and abs(g["hit_global"][1, 2] - 1.25) < 1e-5
    g = golden("prim_kat_implicit")
    h = g["hit_global"][np.isfinite(g["t"])]
    assert np.abs(0.1 * (h[:, 0] ** 2 + h[:, 1] ** 2) + 0.1 * h[:, 2] ** 2 - 2 * h[:, 2]).max() < 1e-4


SEQ = ["singlet", "singlet_tilted_bundle", "singlet_decentred_bundle", "singlet_posed", "singlet_plano",
       "singlet_clear_edge", "singlet_tir", "singlet_sensor", "system5", "doublet", "triplet", "dead_rays",
       "aperture_plain", "aperture_invert", "mirrors", "cyl_singlet", "ideal_thin_lens", "ideal_mirror", "box"]


@pytest.mark.parametrize("name", SEQ)
def test_sequential_trace(ns, name):
    g = golden("seq_" + name)
    elements = scenes.sequential_cases(ns)[name][0]()
    prog = P.compile_elements(elements)
    p, d, I = g["in_pos"], g["in_dir"], g["in_intensity"]
    # per-element intersectTest matrices at the initial state
    tm = O.intersect_matrix(prog, p, d).numpy()
    col = 0
    for k, el in enumerate(elements):
        ref = g[f"it_{k}"]
        got = tm[:, col:col + ref.shape[1]]
        col += ref.shape[1]
        assert pattern_mismatch(got, ref) == 0, (name, k)
        assert rel_err(got, ref) < TOL, (name, k)
    # Element.forward of element 0 on its hit rays
    for j in range(len(elements[0].shape)):
        m = g[f"fw_{j}_mask"]
        if not m.any():
            continue
        S = O._records(prog[j:j + 1], torch.float32)[0]
        hp, nd, im, _ = O.element_forward(S, torch.from_numpy(p[m]), torch.from_numpy(d[m]))
        S64 = O._records(prog[j:j + 1], torch.float64)[0]
        hp64, nd64, _, _ = O.element_forward(S64, torch.from_numpy(p[m]).double(), torch.from_numpy(d[m]).double())
        assert_parity(hp, g[f"fw_{j}_pos"], hp64, what=f"{name} fw{j} pos")
        assert_parity(nd, g[f"fw_{j}_dir"], nd64, what=f"{name} fw{j} dir")
        assert np.array_equal(im.numpy(), g[f"fw_{j}_imult"])
    # element-prefix states and the final state
    n_before = 0
    for k, el in enumerate(elements):
        n_before += len(el.shape)
        o = O.trace_sequential(prog[:n_before], p, d, I)
        o64 = O.trace_sequential(prog[:n_before], p, d, I, dtype=torch.float64)
        for f in ("pos", "dir", "intensity"):
            assert_parity(o[f], g[f"after_{k}_{f}"], o64[f], what=f"{name} after {k} {f}")
    o = O.trace_sequential(prog, p, d, I)
    o64 = O.trace_sequential(prog, p, d, I, dtype=torch.float64)
    alive = g["out_intensity"] > 0
    assert np.array_equal(o["intensity"].numpy() > 0, alive)
    for f in ("pos", "dir", "intensity"):
        assert_parity(o[f], g[f"out_{f}"], o64[f], what=f"{name} out {f}")
    # sensor hit lists
    for slot, mask, hl, w in o["sensor"]:
        k = int(prog[prog["sensor_slot"] == slot]["element_id"][0])
        if f"sensor_{k}_locs" not in g:
            assert not mask.any()
            continue
        assert int(mask.sum()) == g[f"sensor_{k}_locs"].shape[0]
        assert_parity(hl[mask], g[f"sensor_{k}_locs"], what=f"{name} sensor locs")
        assert_parity(w[mask], g[f"sensor_{k}_w"], what=f"{name} sensor w")


def test_naive_ray_cast_and_step(ns):
    g = golden("naive32")
    prog = P.compile_elements(scenes.naive32(ns))
    assert prog.shape[0] == 32
    p, d, I = g["in_pos"], g["in_dir"], g["in_intensity"]
    tmin, win, tm = O.ray_cast(prog, p, d)
    assert pattern_mismatch(tm, g["t_matrix"]) == 0
    assert rel_err(tm, g["t_matrix"]) < TOL
    ref_win = np.where(g["hit_mask"], g["t_matrix"].argmin(1), -1)
    assert np.array_equal(win.numpy(), ref_win)
    hit = g["hit_mask"]
    assert np.array_equal(prog["element_id"][ref_win[hit]], g["el_ids"][hit])
    assert np.array_equal(prog["surface_id"][ref_win[hit]], g["sf_ids"][hit])
    o = O.step_nearest(prog, p, d, I)
    for f in ("pos", "dir", "intensity"):
        assert_parity(o[f], g[f"step1_{f}"], what=f"naive step {f}")


def test_spot_size_loss_value(ns):
    """Loss of the C5 system recomputed by the oracle from the same seeded bundle."""
    g = golden("spot_f32")
    torch.manual_seed(0)
    lens, sensor = scenes.spot_system(ns, grads=False)
    rays = scenes.disk_bundle(ns).sample(65536)
    assert np.array_equal(rays.pos[:64].numpy(), g["first_pos"])
    assert abs(rays.pos.double().sum().item() - g["checksum"][0]) < 1e-6
    prog = P.compile_elements([lens, sensor])
    o = O.trace_sequential(prog, rays.pos, rays.dir, rays.intensity)
    _, mask, hl, w = o["sensor"][0]
    loss = O.spot_size_loss(hl[mask][:, 0], hl[mask][:, 1], w[mask])
    assert int((w[mask] > 0).sum()) == int(g["n_hits"])
    assert abs(loss.item() - float(g["loss"])) / float(g["loss"]) < 1e-5
