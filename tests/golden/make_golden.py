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

Round 4: ``pocomc.sampler`` is imported too -- its only obstacle is the ``import zuko`` line of ``pocomc/flow.py:6``; an
EMPTY module object under that name lets the import statement pass (nothing of zuko is executed, faked or restated by
it: the reference's ``Flow`` class is never instantiated) -- and the orchestrator methods ``Sampler._reweight``
(``sampler.py:717-805``), ``_resample`` (``:680-715``) and ``_compute_evidence`` (``:869-920``) are called unbound on a
duck-typed ``self`` that carries the reference's own ``Particles`` / ``Reparameterize`` and the oracle flow behind the
``.sample`` contract.  Like the MCMC vectors they pin the orchestrator arithmetic *given* a flow.

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
    # pocomc/flow.py:6 ``import zuko``: an empty module satisfies the statement; pocomc.flow.Flow is never constructed
    if "zuko" not in sys.modules:
        sys.modules["zuko"] = types.ModuleType("zuko")
    mods["sampler"] = importlib.import_module("pocomc.sampler")
    return mods


class _Bar:
    """Stand-in for the tqdm wrapper (``tools.py:189-224``): the orchestrator only reports to it."""

    def update_iter(self):
        pass

    def update_stats(self, info):
        pass


def sampler_goldens(ref, out):
    from oracle.maf import OracleMAF
    from pocomc_amd.maf_spec import MAFSpec
    import cases
    S = ref["sampler"].Sampler
    T = ref["tools"]
    for name, c in cases.SAMPLER_CASES.items():
        betas, logzs, rows = cases.sampler_pool(c["seed"])
        nT, N, D = rows["u"].shape
        P = ref["particles"].Particles(N, D)
        for t in range(nT):
            P.update(dict(u=rows["u"][t], x=rows["x"][t], logdetj=rows["logdetj"][t], logl=rows["logl"][t],
                          logp=rows["logp"][t], beta=betas[t], logz=logzs[t]))
        me = types.SimpleNamespace(t=nT, pbar=_Bar(), particles=P, metric=c["metric"], n_effective=c["n_effective"],
                                   dynamic=c["dynamic"], n_active=c["n_active"], have_blobs=False,
                                   dynamic_ratio=T.unique_sample_size(np.ones(c["n_effective"]), k=c["n_active"]) / c["n_active"])
        cur = S._reweight(me, {})
        tag = f"sampler/{name}"
        out[f"{tag}/beta"] = np.asarray(cur["beta"])
        out[f"{tag}/logz"] = np.asarray(cur["logz"])
        out[f"{tag}/ess"] = np.asarray(cur["ess"])
        out[f"{tag}/n_effective_after"] = np.asarray(me.n_effective)
        out[f"{tag}/weights"] = cur["weights"]
        out[f"{tag}/idx"] = cur["u"][:, 0].astype(np.int64)
        for k in ("x", "logdetj", "logl", "logp"):
            out[f"{tag}/kept_{k}"] = cur[k]
        # _resample on that selection, both schemes (sampler.py:702-705), numpy's legacy stream seeded
        for scheme in ("mult", "syst"):
            me2 = types.SimpleNamespace(resample=scheme, n_active=c["n_active"], have_blobs=False)
            np.random.seed(100 + c["seed"])
            r = S._resample(me2, dict(cur))
            out[f"{tag}/{scheme}/rows"] = r["u"][:, 0].astype(np.int64)
            for k in ("x", "logdetj", "logl", "logp"):
                out[f"{tag}/{scheme}/{k}"] = r[k]

    # _compute_evidence (sampler.py:869-920): the oracle flow behind ``.sample``, the reference's scaler, a prior with a
    # bounded support (part of the draws is dropped, :898-901), numpy's legacy stream for the bootstrap
    import torch
    for name, (Dn, Tn, uni, n, seed) in {"maf3_d5": (5, 3, "affine", 600, 17), "nsf3_d4": (4, 3, "rqs", 400, 18)}.items():
        spec = MAFSpec(Dn, Tn, univariate=uni)
        flat = cases.flow_params(spec, seed, gain=1.0)
        maf = OracleMAF(spec, flat)
        prior = cases.CutPrior(Dn)
        rs = np.random.RandomState(seed)
        x_fit = prior.rvs(512, np.random.default_rng(seed))
        sc = ref["scaler"].Reparameterize(Dn, prior.bounds)
        sc.fit(x_fit)
        z = rs.randn(n, Dn).astype(np.float32)

        class _F:
            def sample(self, size):
                assert size == n
                x, lq = maf.sample_from(z)
                return torch.from_numpy(x), torch.from_numpy(lq)
        like = lambda x: -0.5 * np.sum(((x - 0.3) / 0.8) ** 2, axis=1) - 0.1 * x[:, 0] ** 4
        me = types.SimpleNamespace(flow=_F(), scaler=sc, log_prior=prior.logpdf, calls=0, pbar=_Bar())
        me._log_like = lambda x: (like(x), None)
        np.random.seed(seed)
        logz, dlogz = S._compute_evidence(me, n)
        tag = f"evidence/{name}"
        out[f"{tag}/x_fit"] = x_fit
        out[f"{tag}/z"] = z
        out[f"{tag}/logz"] = np.asarray(logz)
        out[f"{tag}/dlogz"] = np.asarray(dlogz)
        out[f"{tag}/calls"] = np.asarray(me.calls)
        out[f"{tag}/spec"] = np.asarray([Dn, Tn, 1 if uni == "rqs" else 0, n, seed])


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

    # --------------------------------------------- BASELINE-size one-step calls (row subsample of the reference's output)
    out = {}
    for name, c in cases.BIG_GOLDEN_CASES.items():
        kernel = getattr(ref["mcmc"], c["kind"])
        state, funcs, opts, aux = cases.build_case(name, ref["scaler"].Reparameterize)
        funcs["flow"] = TorchFlowAdapter(OracleMAF(aux["spec"], aux["flat"]))
        opts["n_max"] = 1
        tag = f"mcmc_big/{name}"
        sl = slice(None, None, c["stride"])
        for k in ("u", "x", "logdetj", "logl", "logp"):
            out[f"{tag}/in/{k}"] = state[k][sl]
        out[f"{tag}/in/scaler_mu"] = funcs["scaler"].mu
        out[f"{tag}/in/scaler_sigma"] = funcs["scaler"].sigma
        np.random.seed(c["seed"])
        res = kernel(state, funcs, opts)
        for k in ("u", "x", "logdetj", "logl", "logp"):
            out[f"{tag}/{k}"] = res[k][sl]
        for k in ("efficiency", "accept", "steps", "calls", "proposal_scale"):
            out[f"{tag}/{k}"] = np.asarray(res[k])
        print(tag, "accept", res["accept"], "calls", res["calls"])
    np.savez_compressed(os.path.join(HERE, "mcmc_big_reference.npz"), **out)
    print("mcmc_big_reference.npz:", len(out), "arrays")

    # ------------------------------------------------- orchestrator methods of pocomc/sampler.py
    out = {}
    sampler_goldens(ref, out)
    np.savez_compressed(os.path.join(HERE, "sampler_reference.npz"), **out)
    print("sampler_reference.npz:", len(out), "arrays")

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
