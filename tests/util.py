"""Shared comparison helpers for the parity tests."""
import numpy as np

TOL = 1e-4  # per-channel float tolerance stated by BASELINE.json:north_star


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def same_bits(a, b):
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


def depth_report(got, want):
    """depth: +inf must match +inf; finite compared. Returns (n_bad_bitexact, n_bad_tol, max_abs)."""
    fin_g, fin_w = np.isfinite(got), np.isfinite(want)
    mism_inf = int((fin_g != fin_w).sum())
    both = fin_g & fin_w
    diff = np.zeros(got.shape, np.float64)
    diff[both] = np.abs(got[both].astype(np.float64) - want[both].astype(np.float64))
    nbits = int((bits(got) != bits(want)).sum())
    return dict(inf_mismatch=mism_inf, not_bitexact=nbits, over_tol=int((diff > TOL).sum()) + mism_inf,
                max_abs=float(diff.max()) if diff.size else 0.0)


def float_report(got, want, tol=TOL):
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    d = np.where(np.isnan(d), np.inf, d)
    px_bad = (d.reshape(d.shape[0], d.shape[1], -1) > tol).any(-1) if d.ndim == 3 else (d > tol)
    return dict(not_bitexact=int((bits(got) != bits(want)).sum()), over_tol=int(px_bad.sum()),
                max_abs=float(d.max()) if d.size else 0.0)


def assert_parity(got, want, what="", color_tol=TOL):
    """The bar: depth and normal bit-exact (pure +,-,*,/,sqrt arithmetic), colour within 1e-4
    per channel (pow() differs by ≤1 ulp between glibc and the device)."""
    dr = depth_report(got["depth"], want["depth"])
    nr = float_report(got["normal"], want["normal"])
    cr = float_report(got["color"], want["color"], color_tol)
    msg = f"{what}: depth {dr} normal {nr} color {cr}"
    assert dr["not_bitexact"] == 0, msg
    assert nr["not_bitexact"] == 0, msg
    assert cr["over_tol"] == 0, msg
    return dict(depth=dr, normal=nr, color=cr)
