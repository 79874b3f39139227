"""Helpers shared by the CPU and GPU tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INF = float("inf")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def rel_err(a, b):
    """Largest relative deviation of `a` from `b` over entries finite in both.

    [N] arrays: |a-b| / max(1, |b|).  [N,3] arrays (positions, directions, normals): per ray
    max_c |a-b| / max(1, ||b||_2), i.e. relative to the length of the reference vector."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        raise AssertionError(f"shape mismatch {a.shape} vs {b.shape}")
    if a.size == 0:
        return 0.0
    if a.ndim == 2 and a.shape[1] == 3:
        fin = np.isfinite(a).all(1) & np.isfinite(b).all(1)
        if not fin.any():
            return 0.0
        a, b = a[fin], b[fin]
        return float(np.max(np.abs(a - b).max(1) / np.maximum(1.0, np.linalg.norm(b, axis=1))))
    fin = np.isfinite(a) & np.isfinite(b)
    if not fin.any():
        return 0.0
    return float(np.max(np.abs(a[fin] - b[fin]) / np.maximum(1.0, np.abs(b[fin]))))


# Relative tolerance of the parity contract (float32 results vs the reference's float32 CPU path).
PARITY_TOL = 1e-5
# Cases whose *reference* float32 result already sits further than that from its own float64
# evaluation (grazing hits on a paraboloid amplify one-ulp input differences); measured
# reference-fp32 vs fp64: 6.5e-5 relative on positions.
# Likewise a lens rotated about all three axes seen from 32 units away: reference-fp32 vs fp64 is
# 1.1e-5 on the first-surface hit positions.
LOOSE_CASES = {"mirrors": 1e-4, "singlet_posed": 3e-5}


def tol_for(name):
    return LOOSE_CASES.get(name, PARITY_TOL)


def pattern_mismatch(a, b):
    """Number of entries where exactly one of a, b is finite."""
    return int((np.isfinite(np.asarray(a)) != np.isfinite(np.asarray(b))).sum())


def rays_from_golden(ns, g, prefix="in_", device="cpu"):
    """Rebuild the exact input rays of a golden file (directions are stored normalised)."""
    pos = torch.from_numpy(g[prefix + "pos"]).to(device)
    d = torch.from_numpy(g[prefix + "dir"]).to(device)
    r = ns.Rays.from_planes(pos.t().contiguous(), d.t().contiguous(),
                            torch.from_numpy(g[prefix + "intensity"]).to(device),
                            torch.from_numpy(g[prefix + "id"]).to(device),
                            torch.zeros(pos.shape[0], device=device))
    return r
