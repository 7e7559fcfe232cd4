"""Comparison helpers of the parity tests (north star: 1e-5 relative, float32 flow)."""
import numpy as np

TOL = 1e-5
MEASURED = {}            # what -> largest relative error seen in this process (printed by scripts / -s runs)


def close(a, b, tol=TOL, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    fin = np.isfinite(b)
    assert (np.isfinite(a) == fin).all(), what
    assert ((a == b) | fin).all(), what                     # same +-inf
    scale = max(1.0, float(np.abs(b[fin]).max()) if fin.any() else 1.0)
    np.testing.assert_allclose(a[fin], b[fin], rtol=tol, atol=tol * scale, err_msg=what)


def rel_rows(a, b):
    """Relative error per walker.  (N, D) arrays: ``max_j |a_ij - b_ij| / max_j |b_ij|`` (a walker's coordinates
    are one vector: coordinates that happen to be near zero are measured against the walker's own size, not
    against the largest element of the whole array); (N,) arrays: ``|a_i - b_i| / |b_i|``.  Non-finite entries
    must coincide and count as zero error."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    fin = np.isfinite(b)
    assert (np.isfinite(a) == fin).all(), "finite masks differ"
    assert ((a == b) | fin | (np.isnan(a) & np.isnan(b))).all(), "different infinities"
    d = np.where(fin, np.abs(np.where(fin, a, 0.0) - np.where(fin, b, 0.0)), 0.0)
    ref = np.where(fin, np.abs(b), 0.0)
    if a.ndim == 2:
        d, ref = d.max(axis=1), ref.max(axis=1)
    return d / np.maximum(ref, np.finfo(np.float64).tiny)


def close_rel(a, b, tol=TOL, what="", cancel=None):
    """Pure relative comparison (north star: 1e-5 relative fp32), walker by walker -- no absolute slack scaled by
    the array's largest element.  ``cancel``: for a quantity that is a SUM of terms of either sign (a log-determinant
    of the float32 flow can pass through zero), the per-walker size of the sum's terms; the error is then measured
    against ``max(|b_i|, cancel_i)`` -- stated per call, never a global maximum."""
    r = rel_rows(a, b)
    if cancel is not None:
        b = np.asarray(b, np.float64)
        fin = np.isfinite(b)
        cancel = np.broadcast_to(np.asarray(cancel, np.float64), b.shape)
        cancel = np.where(np.isfinite(cancel), cancel, 0.0)
        r = r * np.where(fin, np.abs(b), 0.0) / np.maximum(np.maximum(np.where(fin, np.abs(b), 0.0), cancel), np.finfo(np.float64).tiny)
    worst = float(r.max()) if r.size else 0.0
    assert worst <= tol, f"{what}: max relative error {worst:.3e} > {tol:g} ({int((r > tol).sum())} of {r.size} walkers)"
    MEASURED[what.split(",")[0]] = max(MEASURED.get(what.split(",")[0], 0.0), worst)
    return worst
