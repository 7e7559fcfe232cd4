"""Generate golden vectors from the REFERENCE itself (run in the build container).

    python tests/golden/make_golden.py

``import pocomc`` fails here (``pocomc/__init__.py:27`` -> ``flow.py:6`` needs
the absent third-party ``zuko``), so the importable modules are loaded through
a package shim that skips ``__init__``: ``pocomc.mcmc``, ``pocomc.tools``,
``pocomc.particles``, ``pocomc.scaler``, ``pocomc.geometry``, ``pocomc.student``.
The flow handed to the reference's MCMC kernels is the oracle MAF behind the
reference's duck-typed Flow contract (``pocomc/tools.py:336-349``), so these
vectors pin everything *given* a flow; the flow itself stays parity-unpinned
(SURVEY.md section 8(c)).

Outputs (data only -- inputs and the reference's outputs): ``tests/golden/*.npz``.
The GPU box never runs this script and never sees ``/root/reference``.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

REF = "/root/reference/pocomc"


def load_reference():
    pkg = types.ModuleType("pocomc")
    pkg.__path__ = [REF]
    sys.modules["pocomc"] = pkg
    mods = {}
    for m in ("tools", "student", "geometry", "scaler", "particles", "mcmc"):
        mods[m] = importlib.import_module(f"pocomc.{m}")
    return mods


def main():
    import cases
    from oracle.maf import OracleMAF, TorchFlowAdapter

    ref = load_reference()
    out = {}

    # ------------------------------------------------------------ MCMC kernels
    for name, c in cases.MCMC_CASES.items():
        kernel = getattr(ref["mcmc"], c["kind"])
        for n_max in sorted({1, c["n_max"]}):
            state, funcs, opts, aux = cases.build_case(name, ref["scaler"].Reparameterize)
            funcs["flow"] = TorchFlowAdapter(OracleMAF(aux["spec"], aux["flat"]))
            opts["n_max"] = n_max
            np.random.seed(c["seed"])
            res = kernel(state, funcs, opts)
            tag = f"mcmc/{name}/nmax{n_max}"
            for k in ("u", "x", "logdetj", "logl", "logp"):
                out[f"{tag}/{k}"] = res[k]
            for k in ("efficiency", "accept", "steps", "calls", "proposal_scale"):
                out[f"{tag}/{k}"] = np.asarray(res[k])
        # the inputs too, so a drift of cases.py is detected
        state, funcs, opts, aux = cases.build_case(name, ref["scaler"].Reparameterize)
        for k in ("u", "x", "logdetj", "logl", "logp"):
            out[f"mcmc/{name}/in/{k}"] = state[k]
        out[f"mcmc/{name}/in/scaler_mu"] = funcs["scaler"].mu
        out[f"mcmc/{name}/in/scaler_sigma"] = funcs["scaler"].sigma
    np.savez_compressed(os.path.join(HERE, "mcmc_reference.npz"), **out)
    print("mcmc_reference.npz:", len(out), "arrays")

    # ------------------------------------------------------------------ scaler
    # the four bound types of tests/test_scaler.py:9-54, 100x10 data, seed 0
    out = {}
    rs = np.random.RandomState(0)
    D, n = 10, 100
    bound_sets = {
        "none": np.tile(np.array([[-np.inf, np.inf]]), (D, 1)),
        "left": np.tile(np.array([[0.0, np.inf]]), (D, 1)),
        "right": np.tile(np.array([[-np.inf, 5.0]]), (D, 1)),
        "both": np.tile(np.array([[-2.0, 5.0]]), (D, 1)),
    }
    for transform in ("probit", "logit"):
        for bname, bounds in bound_sets.items():
            if bname == "none":
                x = rs.randn(n, D)
            elif bname == "left":
                x = rs.exponential(1.0, size=(n, D))
            elif bname == "right":
                x = 5.0 - rs.exponential(1.0, size=(n, D))
            else:
                x = rs.uniform(-2.0, 5.0, size=(n, D))
            sc = ref["scaler"].Reparameterize(D, bounds, transform=transform)
            sc.fit(x)
            u = sc.forward(x)
            xr, ldj = sc.inverse(u)
            u_far = rs.randn(n, D) * 3.0
            xf, ldjf = sc.inverse(u_far)
            tag = f"scaler/{transform}/{bname}"
            out[f"{tag}/bounds"] = bounds
            out[f"{tag}/x"] = x
            out[f"{tag}/mu"] = sc.mu
            out[f"{tag}/sigma"] = sc.sigma
            out[f"{tag}/u"] = u
            out[f"{tag}/x_rt"] = xr
            out[f"{tag}/ldj"] = ldj
            out[f"{tag}/u_far"] = u_far
            out[f"{tag}/x_far"] = xf
            out[f"{tag}/ldj_far"] = ldjf
    # boundary conditions
    bounds = np.tile(np.array([[-1.0, 2.0]]), (4, 1))
    sc = ref["scaler"].Reparameterize(4, bounds, periodic=[0, 1], reflective=[2])
    xb = rs.uniform(-9.0, 9.0, size=(64, 4))
    xb[:, 3] = rs.uniform(-1.0, 2.0, size=64)
    out["scaler/bc/bounds"] = bounds
    out["scaler/bc/x"] = xb
    out["scaler/bc/x_bc"] = sc.apply_boundary_conditions_x(xb)
    np.savez_compressed(os.path.join(HERE, "scaler_reference.npz"), **out)
    print("scaler_reference.npz:", len(out), "arrays")

    # full (non-diagonal) affine map, scaler.py:172-178 / :288-313 (round 2; its own file, the vectors above stay as they are)
    out = {}
    rs = np.random.RandomState(5)
    D, n = 6, 200
    mix = np.eye(D) + 0.4 * rs.randn(D, D)
    for transform in ("probit", "logit"):
        for bname, bounds in (("none", np.tile(np.array([[-np.inf, np.inf]]), (D, 1))),
                              ("both", np.tile(np.array([[-3.0, 4.0]]), (D, 1)))):
            x = rs.randn(n, D) @ mix.T if bname == "none" else rs.uniform(-3.0, 4.0, size=(n, D))
            sc = ref["scaler"].Reparameterize(D, bounds, transform=transform, diagonal=False)
            sc.fit(x)
            u = sc.forward(x)
            xr, ldj = sc.inverse(u)
            u_far = rs.randn(n, D) * 2.0
            xf, ldjf = sc.inverse(u_far)
            tag = f"scaler_full/{transform}/{bname}"
            for k, v in (("bounds", bounds), ("x", x), ("mu", sc.mu), ("L", sc.L), ("log_det_L", np.asarray(sc.log_det_L)),
                         ("u", u), ("x_rt", xr), ("ldj", ldj), ("u_far", u_far), ("x_far", xf), ("ldj_far", ldjf)):
                out[f"{tag}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "scaler_full_reference.npz"), **out)
    print("scaler_full_reference.npz:", len(out), "arrays")

    # ------------------------------------------------------------------- tools
    out = {}
    T = ref["tools"]
    rs = np.random.RandomState(7)
    for n in (1, 17, 1000, 5000):
        lw = rs.randn(n) * 3.0
        w = np.exp(lw - lw.max())
        out[f"tools/n{n}/logw"] = lw
        out[f"tools/n{n}/ess"] = np.asarray(T.effective_sample_size(w.copy()))
        out[f"tools/n{n}/uss"] = np.asarray(T.unique_sample_size(w.copy()))
        out[f"tools/n{n}/uss_k64"] = np.asarray(T.unique_sample_size(w.copy(), k=64))
        out[f"tools/n{n}/compute_ess"] = np.asarray(T.compute_ess(lw))
        out[f"tools/n{n}/increment_logz"] = np.asarray(T.increment_logz(lw))
        if n >= 17:
            idx, wt = T.trim_weights(np.arange(n), w.copy(), ess=0.99, bins=1000)
            out[f"tools/n{n}/trim_idx"] = idx
            out[f"tools/n{n}/trim_w"] = wt
            wn = w / w.sum()
            for off_seed in (0, 1):
                np.random.seed(off_seed)
                offset = np.random.random()
                np.random.seed(off_seed)
                out[f"tools/n{n}/syst_{off_seed}"] = T.systematic_resample(min(n, 256), wn.copy())
                out[f"tools/n{n}/syst_{off_seed}_offset"] = np.asarray(offset)
                np.random.seed(off_seed)
                uni = np.random.random_sample(min(n, 256))
                np.random.seed(off_seed)
                out[f"tools/n{n}/mult_{off_seed}"] = np.random.choice(np.arange(n), size=min(n, 256), replace=True, p=wn)
                out[f"tools/n{n}/mult_{off_seed}_uniforms"] = uni
    # the reference's only KAT (tests/test_tools.py:10-14)
    for v in (1.0, 251.0, -421.0, -421.125251, 0.0):
        assert T.compute_ess(np.array([v])) == 1.0

    # persistent-sampling log-weights (particles.py:215-231)
    P = ref["particles"].Particles(128, 5)
    Tn = 7
    betas = np.sort(rs.uniform(0, 1, Tn)); betas[0] = 0.0
    logzs = np.cumsum(rs.randn(Tn)) * 0.5; logzs[0] = 0.0
    logl = rs.randn(Tn, 128) * 4.0 - 5.0
    for t in range(Tn):
        P.update(dict(logl=logl[t], beta=betas[t], logz=logzs[t]))
    out["particles/logl"] = logl
    out["particles/beta"] = betas
    out["particles/logz"] = logzs
    for bf in (0.3, 1.0):
        lw, lz = P.compute_logw_and_logz(bf)
        out[f"particles/logw_b{bf}"] = lw
        out[f"particles/logz_b{bf}"] = np.asarray(lz)
        lw2, _ = P.compute_logw_and_logz(bf, normalize=False)
        out[f"particles/logw_raw_b{bf}"] = lw2

    # geometry / student-t (geometry.py:31-59, student.py:5-85)
    G = ref["geometry"].Geometry()
    th = rs.standard_t(4.0, size=(600, 5)) @ (np.eye(5) + 0.3 * rs.randn(5, 5))
    G.fit(th)
    out["geometry/theta"] = th
    out["geometry/t_mean"] = G.t_mean
    out["geometry/t_cov"] = G.t_cov
    out["geometry/t_nu"] = np.asarray(G.t_nu)
    out["geometry/normal_mean"] = G.normal_mean
    out["geometry/normal_cov"] = G.normal_cov
    wq = rs.uniform(0.1, 1.0, 600); wq /= wq.sum()
    np.random.seed(11)
    G2 = ref["geometry"].Geometry()
    G2.fit(th, weights=wq)
    out["geometry/w"] = wq
    out["geometry/w_t_mean"] = G2.t_mean
    out["geometry/w_t_cov"] = G2.t_cov
    out["geometry/w_t_nu"] = np.asarray(G2.t_nu)
    out["geometry/w_normal_cov"] = G2.normal_cov
    np.savez_compressed(os.path.join(HERE, "tools_reference.npz"), **out)
    print("tools_reference.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
