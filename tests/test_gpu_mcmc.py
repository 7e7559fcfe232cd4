"""Parity of the device MCMC step against the oracle (which is pinned to the
reference by ``tests/test_oracle_golden.py``), in replay mode: the random
variates the reference drew from numpy's legacy stream are inputs.

Two levels:
* teacher-forced, step by step: the oracle's pre-step state goes in, the proposal
  (theta', u', x', log-dets), alpha and the accept decisions come out and are compared.
  Decisions may only differ where ``|u_rand - alpha|`` is below the fp32 flow's noise.
* whole kernel call against the golden vectors generated from the reference itself.
"""
import numpy as np
import pytest
import torch

import cases
from oracle import mcmc as omcmc
from oracle.maf import OracleMAF, TorchFlowAdapter
from oracle.scaler import Reparameterize as OracleScaler

pytestmark = pytest.mark.gpu

TOL = 1e-5
# Spline flows (cases with flow="rqs"): two float32 evaluations of the rational-quadratic spline agree to eps * cond
# (knot differences cancel), not to 1e-5 -- tests/test_gpu_flow.py::test_nsf_float32_evaluations_against_the_float64_yardstick
# measures both the oracle's and the kernels' distance to the float64 evaluation of the same parameters; the bounds
# below are the ones of tests/test_gpu_flow.py (x: 5e-5, log-determinants: 1e-4 of their terms).
NSF_X, NSF_LADJ = 5e-5, 1e-4


def tols(case):
    """(state tolerance, log-determinant tolerance) of a case: the north star's 1e-5, or the spline flows' stated bounds."""
    return (NSF_X, NSF_LADJ) if case.get("flow") == "rqs" else (TOL, TOL)


from parity import close, rel_rows, close_rel                      # noqa: E402


def product_case(name):
    """The case with the product's scaler and flow."""
    from pocomc_amd import Flow, Reparameterize
    state, funcs, opts, aux = cases.build_case(name, Reparameterize)
    flow = Flow(aux["spec"].n_dim, aux["spec"])
    flow.set_params(aux["flat"])
    funcs["flow"] = flow
    return state, funcs, opts, aux


def oracle_case(name):
    state, funcs, opts, aux = cases.build_case(name, OracleScaler)
    funcs["flow"] = TorchFlowAdapter(OracleMAF(aux["spec"], aux["flat"]))
    return state, funcs, opts, aux


@pytest.mark.parametrize("name", list(cases.MCMC_CASES))
def test_inputs_built_with_device_scaler_match_golden(name, golden_dir):
    """scaler.fit / forward / inverse on the device reproduce the inputs the reference saw."""
    g = np.load(f"{golden_dir}/mcmc_reference.npz")
    state, funcs, _, _ = product_case(name)
    np.testing.assert_allclose(funcs["scaler"].mu, g[f"mcmc/{name}/in/scaler_mu"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(funcs["scaler"].sigma, g[f"mcmc/{name}/in/scaler_sigma"], rtol=1e-12, atol=1e-12)
    for k in ("u", "logdetj"):
        np.testing.assert_allclose(state[k], g[f"mcmc/{name}/in/{k}"], rtol=1e-11, atol=1e-11, err_msg=k)


class VerifiedInverse(TorchFlowAdapter):
    """For sizes where zuko's D-pass inverse (``oracle.maf.OracleMAF.inverse``: (D+1) dense passes per transform) does
    not finish in seconds: the inverse comes from the device and is VERIFIED by the oracle before the oracle's step
    uses it -- a bijection's inverse is right iff the oracle's forward map takes it back to the input
    (``tests/test_flow.py:88``) and the log-determinants are opposite (``:164``): every row, at 10 x TOL (the forward map
    amplifies); and 256 evenly spaced rows go through the oracle's D-pass inverse itself and are compared at TOL."""

    def __init__(self, maf, product_flow, tol=TOL):
        super().__init__(maf)
        self.product_flow, self.tol, self.worst = product_flow, tol, 0.0
        self.subsample, self.worst_direct, self.n_direct = 256, 0.0, 0

    def inverse(self, theta):
        x, l = self.product_flow.inverse(theta)
        xn = x.numpy()
        # rows the float32 flow cannot represent (a far-out theta' whose inverse overflows: |x| beyond 1e4 or not finite)
        # have no meaningful round trip in float32 -- zuko's inverse overflows on them alike; the step rejects them
        # through the finite mask / the prior's support (mcmc.py:100-109).  They must be rare.
        ok = np.isfinite(xn).all(axis=1) & (np.abs(np.where(np.isfinite(xn), xn, 0.0)).max(axis=1) < 1e4) & np.isfinite(l.numpy())
        assert ok.mean() > 0.995, f"{(~ok).sum()} of {len(ok)} rows overflow in the inverse"
        back, lf = self.maf.forward(xn[ok])
        # the forward map amplifies an error of x by the flow's Jacobian; measured per walker against |theta|
        self.worst = max(self.worst, close_rel(back, theta.numpy()[ok], 10 * self.tol, "oracle.forward(device inverse)"))
        close_rel(-lf, l.numpy()[ok], 10 * self.tol, "ladj antisymmetry", cancel=self.maf.ladj_abs_terms(xn[ok]))
        # ... and next to the round trip (a bijection argument, 10 x TOL) a direct comparison at TOL: 256 of the rows, evenly
        # spaced, through the oracle's own D-pass inverse (zuko's algorithm; seconds at this size)
        sub = np.flatnonzero(ok)[:: max(1, int(ok.sum()) // self.subsample)][: self.subsample]
        xo, lo = self.maf.inverse(theta.numpy()[sub])
        fin = np.isfinite(xo).all(axis=1) & np.isfinite(lo)
        assert fin.mean() > 0.99
        self.worst_direct = max(self.worst_direct,
                                close_rel(xn[sub][fin], xo[fin], self.tol, "device inverse vs oracle D-pass inverse (subsample)"))
        close_rel(l.numpy()[sub][fin], lo[fin], self.tol, "ladj of the inverse vs oracle (subsample)",
                  cancel=self.maf.ladj_abs_terms(xo[fin]))
        self.n_direct += int(fin.sum())
        return x, l


@pytest.mark.parametrize("name", list(cases.MCMC_CASES))
def test_step_teacher_forced(name):
    teacher_forced(name)


def teacher_forced(name, verified_inverse=False):
    from pocomc_amd.mcmc import StepEngine
    c = cases.find_case(name)
    TOL, TOL_L = tols(c)
    kind = c["kind"]
    pre = kind.startswith("preconditioned")
    tpcn = kind in ("preconditioned_pcn", "pcn")
    # oracle run with trace and recorded variates
    state, funcs, opts, aux = oracle_case(name)
    pstate, pfuncs, popts, paux = product_case(name)
    if verified_inverse:
        funcs["flow"] = VerifiedInverse(funcs["flow"].maf, pfuncs["flow"], tol=max(TOL, TOL_L))
    oflow = funcs["flow"].maf if pre else None              # (log-determinants are measured against the size of their terms)
    # spline flows: next to the float32 oracle (stated bounds NSF_X / NSF_LADJ) the proposal's u' is held to the north
    # star's 1e-5 against the EXACT inverse of the same float32 parameters (float64 arithmetic, OracleMAF(dtype=float64))
    yard = (OracleMAF(oflow.spec, oflow.flat, dtype=np.float64)
            if (pre and c.get("flow") == "rqs" and not verified_inverse) else None)
    rng = omcmc.LegacyStream()
    trace = []
    np.random.seed(c["seed"])
    getattr(omcmc, kind)(state, funcs, opts, rng=rng, trace=trace)
    assert len(trace) >= 1

    N, D = c["N"], c["D"]
    geo = pfuncs["theta_geometry"]
    eng = StepEngine(kind, N, D, pfuncs["flow"] if pre else None, pfuncs["scaler"])
    eng.load_state(state["u"], state["x"], state["logdetj"], state["logl"], state["logp"])
    nu = float(geo.t_nu) if tpcn else 0.0
    if tpcn:
        eng.set_geometry(mu=geo.t_mean, cov=geo.t_cov)
    else:
        eng.set_geometry(cov=geo.normal_cov)

    sigma = np.minimum(opts["proposal_scale"], 0.99) if tpcn else opts["proposal_scale"]
    mu = np.array(geo.t_mean, dtype=float)
    prev = None
    n_flips = 0
    worst = {}
    for i, tr in enumerate(trace):
        if prev is not None:
            # teacher forcing: start every step from the oracle's state
            eng.u.copy_(torch.from_numpy(prev["u"])); eng.x.copy_(torch.from_numpy(prev["x"]))
            eng.logdetj.copy_(torch.from_numpy(prev["logdetj"]))
            eng.logl.copy_(torch.from_numpy(prev["logl"])); eng.logp.copy_(torch.from_numpy(prev["logp"]))
            if pre:
                eng.theta32.copy_(torch.from_numpy(prev["theta"].astype(np.float32)))
                eng.ldjf.copy_(torch.from_numpy(prev["logdetj_flow"].astype(np.float32)))
            sigma = prev["sigma"]
            if kind == "preconditioned_pcn":
                mu = prev["mu"]
        elif pre:
            # the theta the device derived from u must be the oracle's
            th0, l0 = omcmc.flow_numpy_wrapper(funcs["flow"]).forward(state["u"])
            close_rel(eng.theta32.cpu().numpy(), th0, TOL, "theta0")
            close_rel(eng.ldjf.cpu().numpy(), l0, TOL_L, "logdetj_flow0", cancel=oflow.ladj_abs_terms(state["u"]))
            eng.theta32.copy_(torch.from_numpy(th0)); eng.ldjf.copy_(torch.from_numpy(l0))
        if tpcn:
            eng.set_mu(mu)
        rec = rng.record[i]
        eng.propose(sigma, nu, dict(gamma=rec.get("gamma"), z=rec["z"], u=rec["u"]))
        # ---- proposal: pure relative per walker (north star 1e-5; theta' itself is float64 arithmetic).  The flow's
        # log-determinant is a float32 sum of T*D log-scales of either sign: measured against the size of its terms
        # (OracleMAF.ladj_abs_terms); the scaler's (float64) against max(|log|, 1)
        worst["theta_prime"] = max(worst.get("theta_prime", 0), close_rel(
            eng.p_theta64.cpu().numpy(), tr["theta_prime"], 1e-12 if not pre else 2e-7, "theta_prime"))
        worst["u_prime"] = max(worst.get("u_prime", 0), close_rel(eng.p_u.cpu().numpy(), tr["u_prime"], TOL, "u_prime"))
        worst["x_prime"] = max(worst.get("x_prime", 0), close_rel(eng.p_x.cpu().numpy(), tr["x_prime"], TOL, "x_prime"))
        if yard is not None:
            u64, l64 = yard.inverse(tr["theta_prime"].astype(np.float32))
            fin = np.isfinite(u64).all(axis=1) & np.isfinite(tr["u_prime"]).all(axis=1)
            # against the float64 evaluation of the same parameters: the north star's 1e-5, or -- where float32 arithmetic
            # itself does not reach it -- twice the float32 ORACLE's own distance to that evaluation (zuko's arithmetic in
            # float32; the worst walker of tpcn_n256_d10_nsf6 sits 1.1-1.8e-5 away in either float32 evaluation: six
            # transforms amplify a rounding of the hidden layers ~100-fold there, whatever the spline's formula does)
            o32 = float(rel_rows(tr["u_prime"][fin], u64[fin]).max())
            worst["oracle_f32_u_prime_f64"] = max(worst.get("oracle_f32_u_prime_f64", 0), o32)
            worst["u_prime_f64"] = max(worst.get("u_prime_f64", 0), close_rel(
                eng.p_u.cpu().numpy()[fin], u64[fin], max(TOL, 2.0 * o32) if D >= 4 else NSF_X, "u_prime vs the float64 evaluation"))
            worst["ldjf_prime_f64"] = max(worst.get("ldjf_prime_f64", 0), close_rel(
                eng.p_ldjf.cpu().numpy()[fin], l64[fin], NSF_LADJ, "logdetj_flow_prime vs the float64 evaluation",
                cancel=oflow.ladj_abs_terms(tr["u_prime"][fin])))
        # (the scaler's log-determinant is a float64 function of the float32 u': it inherits u's tolerance through
        #  d logdetj / d u_j ~ -t_j sigma_j, i.e. |d logdetj| <= 1e-5 (1 + sum_j u_j'^2) to first order)
        worst["logdetj_prime"] = max(worst.get("logdetj_prime", 0), close_rel(
            eng.p_logdetj.cpu().numpy(), tr["logdetj_prime"], TOL, "logdetj_prime",
            cancel=1.0 + np.sum(np.where(np.isfinite(tr["u_prime"]), tr["u_prime"], 0.0) ** 2, axis=1)))
        if pre:
            worst["logdetj_flow_prime"] = max(worst.get("logdetj_flow_prime", 0), close_rel(
                eng.p_ldjf.cpu().numpy(), tr["logdetj_flow_prime"], TOL_L, "logdetj_flow_prime",
                cancel=oflow.ladj_abs_terms(tr["u_prime"])))
        calls, _ = eng.evaluate(pfuncs["logprior"], pfuncs["loglike"])
        assert abs(calls - int(tr["finite"].sum())) <= 1
        cur = eng.download()                                  # the state the accept kernel starts from
        cur_ldjf = eng.ldjf.cpu().numpy().astype(np.float64) if pre else None
        sums = eng.accept_reduce(c["beta"], nu, want_mask=True)
        alpha = eng.alpha.cpu().numpy()
        acc = eng.h_accept.numpy().astype(bool)
        # ---- M1d: alpha is mcmc.py:124-134 applied to what the device holds (its own x', the host's logl'/logp' for
        # that x'): float64 arithmetic, checked to 1e-9 relative.  Against the oracle's alpha the difference is the
        # likelihood's response to a 1e-7-relative change of x' (Rosenbrock: d logl ~ 1e4 |dx|), not kernel error:
        # |d log alpha| <= 2e-3 is asserted below as well.
        p_logl, p_logp = eng.p_logl.cpu().numpy(), eng.p_logp.cpu().numpy()
        expo = (p_logl * c["beta"] - cur["logl"] * c["beta"] + p_logp - cur["logp"]
                + eng.p_logdetj.cpu().numpy() - cur["logdetj"])
        if pre:
            expo = expo + eng.p_ldjf.cpu().numpy().astype(np.float64) - cur_ldjf
        if tpcn:
            A = -(D + nu) / 2 * np.log(1 + eng.p_quad.cpu().numpy() / nu)
            B = -(D + nu) / 2 * np.log(1 + eng.quad.cpu().numpy() / nu)
            expo = expo - A + B
        with np.errstate(over="ignore", invalid="ignore"):
            alpha_chk = np.minimum(1.0, np.exp(expo))
        alpha_chk[np.isnan(alpha_chk)] = 0.0
        np.testing.assert_allclose(alpha, alpha_chk, rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(alpha, tr["alpha"], rtol=2e-3, atol=2e-5)
        # ---- M1e: the decision is exactly u_rand < alpha; it differs from the oracle's only where u_rand lies between
        # the two alphas.  Flip budget: 2 per case over all steps (expected: sum |alpha_dev - alpha_oracle| ~ 0.1)
        assert np.array_equal(acc, rec["u"] < alpha), f"step {i}: decision != (u < alpha)"
        flips = acc != tr["accept"]
        lo_, hi_ = np.minimum(alpha, tr["alpha"]), np.maximum(alpha, tr["alpha"])
        assert ((rec["u"][flips] >= lo_[flips]) & (rec["u"][flips] <= hi_[flips])).all()
        n_flips += int(flips.sum())
        ok = ~flips
        post = eng.download()
        for k in ("u", "x", "logdetj", "logl", "logp"):
            # non-flipped walkers: pure 1e-5 relative (accepted ones carry the device's proposal, the others the oracle's
            # own previous state bit for bit)
            worst["post_" + k] = max(worst.get("post_" + k, 0), close_rel(
                post[k][ok], tr[k][ok], TOL, f"post {k}",
                cancel=None if k in ("u", "x") else 1.0 + np.sum(tr["u"][ok] ** 2, axis=1) if k == "logdetj" else 1.0))
        if not flips.any():
            np.testing.assert_allclose(sums[0] / N, tr["alpha"].mean(), rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(sums[1] / N, (tr["logl"] + tr["logp"]).mean(), rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(sums[3], tr["accept"].sum())
            moved = tr["theta"] if pre else tr["u"]
            np.testing.assert_allclose(sums[4:4 + D] / N, moved.mean(axis=0, dtype=np.float64), rtol=1e-4, atol=1e-5)
        prev = tr
    print(f"{name}: accept flips {n_flips} / {N * len(trace)}; worst relative errors {worst}")
    assert n_flips <= 2


def _off_trajectory(res, g, tag, tol, rows=slice(None)):
    """Walkers whose final state differs from the reference's by more than ``tol`` (pure relative per walker; logs are
    compared as densities / determinants below |log| = 1), and the worst error of the others."""
    off = np.zeros(len(g[f"{tag}/x"]), dtype=bool)
    worst = 0.0
    for k in ("u", "x", "logdetj", "logl", "logp"):
        a, b = res[k][rows], g[f"{tag}/{k}"]
        r = rel_rows(a, b)
        if a.ndim == 1:
            r = r * np.abs(b) / np.maximum(np.abs(b), 1.0)
            r = np.where(np.isfinite(r), r, 0.0)
        off |= r > tol
        worst = max(worst, float(r[~off].max()) if (~off).any() else 0.0)
    return off, worst


GOLDEN_RUNS = [(name, 0) for name in cases.MCMC_CASES] + [
    # the spline flows once more with the proposal / sweep / scaler as separate launches (PMC_NO_FUSE = 3): the default
    # run above goes through the fused maf_inverse_nsf2 instances (proposal prologue, scaler + prior epilogue)
    (name, 3) for name, c in cases.MCMC_CASES.items() if c.get("flow") == "rqs"]


@pytest.mark.parametrize("name,no_fuse", GOLDEN_RUNS)
def test_kernel_call_matches_reference_golden(name, no_fuse, golden_dir, monkeypatch):
    """Whole call through the reference's contract vs the vectors the reference produced."""
    from pocomc_amd import mcmc as pmcmc
    monkeypatch.setenv("PMC_NO_FUSE", str(no_fuse))
    g = np.load(f"{golden_dir}/mcmc_reference.npz")
    c = cases.MCMC_CASES[name]
    TOL = tols(c)[0]
    for n_max in sorted({1, c["n_max"]}):
        state, funcs, opts, aux = product_case(name)
        opts["n_max"] = n_max
        np.random.seed(c["seed"])
        res = getattr(pmcmc, c["kind"])(state, funcs, opts, replay=omcmc.LegacyStream())
        tag = f"mcmc/{name}/nmax{n_max}"
        same_path = res["steps"] == int(g[f"{tag}/steps"])
        assert abs(res["steps"] - int(g[f"{tag}/steps"])) <= 1
        if not same_path:
            continue
        assert abs(res["calls"] - int(g[f"{tag}/calls"])) <= 2
        # Walker by walker against the reference's final state, pure relative (rel_rows).  A walker is OFF the
        # reference's trajectory only through an accept flip (u_rand between the float32 flow's alpha and the
        # reference's: teacher-forced test above).  Budget, stated: in a ONE-step call (n_max = 1) at most 2
        # walkers may be off at 1e-5 relative (spline flows: at their stated bound, NSF_X).  In a longer call every
        # walker's step k+1 proposal is scaled by sigma_{k+1} = f(mean alpha_k) (mcmc.py:152-156): the float32 flow's
        # noise in alpha (and any flip, 1/N) reaches ALL walkers through sigma and mu, so the set follows the reference
        # at ~1e-3, not 1e-5; the walker-by-walker 1e-5 statement for whole calls of several steps is
        # test_whole_call_with_the_references_sigma_and_mu_follows_it_walker_by_walker below (sigma_k and mu_k from the
        # oracle, everything else on the device), the step-by-step one the teacher-forced test.
        off, worst = _off_trajectory(res, g, tag, TOL)
        print(f"{tag}: {int(off.sum())} of {off.size} walkers off the reference trajectory at {TOL:g} relative "
              f"(worst of the others {worst:.2e}); sigma ratio {res['proposal_scale'] / float(g[f'{tag}/proposal_scale']) - 1.0:.2e}")
        if n_max == 1:
            assert off.sum() <= 2, f"{tag}: {int(off.sum())} walkers off the reference trajectory"
        else:
            for k in ("x", "logl"):
                a, b = res[k], g[f"{tag}/{k}"]
                fin = np.isfinite(b)
                rel = np.abs(a - b)[fin] / np.maximum(1.0, np.abs(b[fin]))
                assert (rel < 1e-3).mean() > 0.97, f"{tag}/{k}: {(rel < 1e-3).mean()}"
        np.testing.assert_allclose(res["accept"], g[f"{tag}/accept"], atol=0.02)
        np.testing.assert_allclose(res["proposal_scale"], g[f"{tag}/proposal_scale"], rtol=5e-3)


@pytest.mark.parametrize("name", list(cases.MCMC_CASES))
def test_whole_call_with_the_references_sigma_and_mu_follows_it_walker_by_walker(name, golden_dir, monkeypatch):
    """The multi-step parity statement (mcmc.py:74-180).  A whole kernel call of ``n_max`` steps runs on the device; the
    walkers evolve there and nowhere else.  The ONLY quantities taken from the oracle's trace are the two global ones of
    every step -- sigma_k and mu_k, functions of means over all walkers (mcmc.py:152-156), through which one accept flip
    or the float32 flow's noise in alpha would otherwise reach every walker's next proposal -- and the stop decision that
    depends on them.  Then every walker must sit within 1e-5 (spline flows: their stated bound) of the reference's FINAL
    golden state, except walkers whose accept decision flipped, and a decision may only flip where the uniform draw lies
    between the device's alpha and the oracle's: at most 2 new flips per step."""
    from pocomc_amd import mcmc as pmcmc
    g = np.load(f"{golden_dir}/mcmc_reference.npz")
    c = cases.MCMC_CASES[name]
    TOL = tols(c)[0]
    kind = c["kind"]
    # the oracle's run: trace (sigma_k, mu_k, alpha_k, decisions) and the variates the reference drew
    state, funcs, opts, _ = oracle_case(name)
    rng, otrace = omcmc.LegacyStream(), []
    np.random.seed(c["seed"])
    ores = getattr(omcmc, kind)(state, funcs, opts, rng=rng, trace=otrace)
    n_steps = len(otrace)
    tag = f"mcmc/{name}/nmax{c['n_max']}"
    assert n_steps == int(g[f"{tag}/steps"]) == ores["steps"]

    class Forced(pmcmc.Adaptation):
        """The product's adaptation with its two global results replaced by the oracle's after every step."""

        def update(self, sums):
            super().update(sums)
            tr = otrace[self.i - 1]
            self.sigma = tr["sigma"]
            if self.kind == "preconditioned_pcn":
                self.mu = tr["mu"].copy()
            return self.i >= n_steps

    monkeypatch.setattr(pmcmc, "Adaptation", Forced)
    pstate, pfuncs, popts, _ = product_case(name)
    ptrace = []
    res = getattr(pmcmc, kind)(pstate, pfuncs, popts, replay=omcmc.Replay(rng.record), trace=ptrace)
    assert res["steps"] == n_steps == len(ptrace)
    # ---- step by step: decisions are u < alpha; a walker leaves the reference's trajectory only through a flip inside
    # the alpha gap; walkers that left are not looked at again
    N = c["N"]
    on = np.ones(N, dtype=bool)
    flips_per_step = []
    for k in range(n_steps):
        u_rand, a_dev, a_ref = rng.record[k]["u"], ptrace[k]["alpha"], otrace[k]["alpha"]
        assert np.array_equal(ptrace[k]["accept"], u_rand < a_dev), f"step {k}: decision != (u < alpha)"
        flips = on & (ptrace[k]["accept"] != otrace[k]["accept"])
        lo, hi = np.minimum(a_dev, a_ref), np.maximum(a_dev, a_ref)
        assert ((u_rand[flips] >= lo[flips]) & (u_rand[flips] <= hi[flips])).all(), f"step {k}: a flip outside the alpha gap"
        np.testing.assert_allclose(a_dev[on], a_ref[on], rtol=2e-3, atol=2e-5)
        flips_per_step.append(int(flips.sum()))
        on &= ~flips
    assert max(flips_per_step) <= 2, flips_per_step
    # ---- the final state of every walker that never flipped, against the REFERENCE's own output
    off, worst = _off_trajectory(res, g, tag, TOL)
    print(f"{tag}: sigma / mu forced, {n_steps} steps: {int((off & on).sum())} of {int(on.sum())} unflipped walkers off the "
          f"reference's final state at {TOL:g} (worst {worst:.2e}); flips per step {flips_per_step}")
    assert not (off & on).any(), f"{tag}: {int((off & on).sum())} walkers left the reference's trajectory without a flip"
    assert abs(res["calls"] - int(g[f"{tag}/calls"])) <= 2 * n_steps


@pytest.mark.parametrize("name", list(cases.BIG_GOLDEN_CASES))
@pytest.mark.parametrize("no_fuse", [0, 3])
def test_one_step_call_at_baseline_size_matches_reference_golden(name, no_fuse, golden_dir, monkeypatch):
    """1e4 walkers x 32-D, one step (maf3 and nsf3; fused launch and separate launches): every 16th walker against the
    reference's own output for it.  Rosenbrock from prior draws: d logl ~ 1e4 |dx|, so logl carries x's tolerance times
    the likelihood's slope -- it is compared through alpha's consequences only (the accept decision) and against the
    host likelihood of the device's own x."""
    from pocomc_amd import mcmc as pmcmc
    monkeypatch.setenv("PMC_NO_FUSE", str(no_fuse))
    g = np.load(f"{golden_dir}/mcmc_big_reference.npz")
    c = cases.BIG_GOLDEN_CASES[name]
    TOL = tols(c)[0]
    state, funcs, opts, aux = product_case(name)
    opts["n_max"] = 1
    np.random.seed(c["seed"])
    res = pmcmc.preconditioned_pcn(state, funcs, opts, replay=omcmc.LegacyStream())
    tag = f"mcmc_big/{name}"
    rows = slice(None, None, c["stride"])
    assert res["steps"] == 1 and abs(res["calls"] - int(g[f"{tag}/calls"])) <= 2
    off = np.zeros(len(g[f"{tag}/x"]), dtype=bool)
    worst = {}
    for k in ("u", "x", "logdetj", "logp"):
        a, b = res[k][rows], g[f"{tag}/{k}"]
        r = rel_rows(a, b)
        if a.ndim == 1:
            r = np.where(np.isfinite(r), r * np.abs(b) / np.maximum(np.abs(b), 1.0), 0.0)
        off |= r > TOL
        worst[k] = float(r[~off].max())
    # logl: the host likelihood of the state the device holds, exactly; and the reference's logl within the likelihood's
    # response to TOL (Rosenbrock: |d logl| <= |grad| |dx|)
    np.testing.assert_allclose(res["logl"], aux["target"](res["x"]), rtol=1e-12)
    print(f"{tag} (no_fuse={no_fuse}): {int(off.sum())} of {off.size} sampled walkers off the reference at {TOL:g}; worst of the others {worst}")
    assert off.sum() <= 2
    np.testing.assert_allclose(res["accept"], g[f"{tag}/accept"], atol=2e-3)
    np.testing.assert_allclose(res["proposal_scale"], g[f"{tag}/proposal_scale"], rtol=1e-3)


def test_philox_mode_statistics():
    """Throughput mode: counter-based RNG.  Moments of the proposal noise and the gamma
    scale, acceptance in a sane range, determinism for a fixed seed."""
    from pocomc_amd.mcmc import StepEngine
    name = "tpcn_n512_d32_uniform"
    state, funcs, opts, aux = product_case(name)
    c = cases.MCMC_CASES[name]
    N, D = c["N"], c["D"]
    geo = funcs["theta_geometry"]
    outs = []
    for rep in range(2):
        eng = StepEngine("preconditioned_pcn", N, D, funcs["flow"], funcs["scaler"], seed=1234)
        eng.load_state(state["u"], state["x"], state["logdetj"], state["logl"], state["logp"])
        eng.set_geometry(mu=geo.t_mean, cov=geo.t_cov)
        eng.propose(0.3, 5.0)
        outs.append(eng.p_theta64.cpu().numpy().copy())
        theta = eng.theta32.cpu().numpy().astype(np.float64)
    np.testing.assert_array_equal(outs[0], outs[1])
    # theta' - mu - sqrt(1-s^2)(theta-mu) = s*sqrt(s_k) L z  ->  whiten and check moments
    L = np.linalg.cholesky(geo.t_cov)
    r = outs[0] - geo.t_mean - (1 - 0.3 ** 2) ** 0.5 * (theta - geo.t_mean)
    w = np.linalg.solve(L, r.T).T / 0.3                   # sqrt(s_k) z_k
    zn = w / np.sqrt((w ** 2).mean(axis=1, keepdims=True))
    assert abs(zn.mean()) < 0.02 and abs((zn ** 2).mean() - 1) < 1e-6
    assert abs(np.mean(zn[:, 0] * zn[:, 1])) < 0.15
    s = (w ** 2).mean(axis=1)                              # ~ s_k = 1/Gamma((D+nu)/2, 2/(nu+delta))
    assert np.all(s > 0) and np.isfinite(s).all()
    calls, _ = eng.evaluate(funcs["logprior"], funcs["loglike"])
    sums = eng.accept_reduce(c["beta"], 5.0)
    assert 0.0 <= sums[0] / N <= 1.0


@pytest.mark.parametrize("D,N,kind", [(4, 100, 0), (10, 333, 0), (32, 1000, 0), (50, 77, 0), (6, 64, 1)])
def test_fused_proposal_and_inverse_equal_the_two_launches(D, N, kind):
    """pmc_propose_inverse (the proposal as prologue of the flow-inverse kernel) gives bit for bit the
    theta', quadratic forms, u' and log-determinant of pmc_propose followed by pmc_maf_inverse."""
    import ctypes as C
    import torch
    import pocomc_amd as pc
    from pocomc_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(D)
    flow = pc.Flow(D, "maf3", seed=1)
    A = rng.normal(size=(D, D))
    cov = A @ A.T / D + np.eye(D)
    up = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda()
    mu, icov, chol = up(rng.normal(size=D)), up(np.linalg.inv(cov)), up(np.linalg.cholesky(cov))
    cur32 = up(rng.normal(size=(N, D)), torch.float32)
    r = _lib.pmc_rng_t(gamma=None, normal=None, uniform=None, seed=1234, step=7, offset=5)
    mk = lambda *s, dt=torch.float64: torch.zeros(*s, dtype=dt, device="cuda")
    st = _lib.stream_handle()
    # two launches
    t64, t32, qa, qb = mk(N, D), mk(N, D, dt=torch.float32), mk(N), mk(N)
    _lib.check(lib.pmc_propose(kind, _lib.ptr(cur32), None, _lib.ptr(mu), _lib.ptr(icov), _lib.ptr(chol), 5.0, 0.4,
                               float((1 - 0.4 ** 2) ** 0.5), C.byref(r), _lib.ptr(t64), _lib.ptr(t32), _lib.ptr(qa),
                               _lib.ptr(qb), N, D, st))
    u_a, l_a = mk(N, D, dt=torch.float32), mk(N, dt=torch.float32)
    # (the sweep the fused launch contains: two waves per 16 rows -- AUTO would take the lane-per-walker sweep at D = 50)
    _lib.check(lib.pmc_maf_inverse(C.byref(flow._desc), _lib.ptr(t32), _lib.ptr(u_a), _lib.ptr(l_a), N, 7, st))
    # one launch
    t64f, qaf, qbf = mk(N, D), mk(N), mk(N)
    u_b, l_b = mk(N, D, dt=torch.float32), mk(N, dt=torch.float32)
    _lib.check(lib.pmc_propose_inverse(kind, _lib.ptr(cur32), _lib.ptr(mu), _lib.ptr(icov), _lib.ptr(chol), 5.0, 0.4,
                                       float((1 - 0.4 ** 2) ** 0.5), C.byref(r), _lib.ptr(t64f), _lib.ptr(qaf),
                                       _lib.ptr(qbf), C.byref(flow._desc), _lib.ptr(u_b), _lib.ptr(l_b), N, st))
    for a, b in ((t64, t64f), (u_a, u_b), (l_a, l_b)) + (((qa, qaf), (qb, qbf)) if kind == 0 else ()):
        assert torch.equal(a, b)


def test_variates_drawn_ahead_equal_inline_draws():
    """pmc_rng_fill writes exactly the Philox variates the kernels draw inline, so a kernel call with the
    variates of step k+1 generated behind step k (rng_prefill) is bit-identical to one that draws inline."""
    from scipy.stats import uniform
    import pocomc_amd as pc
    from pocomc_amd import mcmc as pmcmc
    from pocomc_amd.geometry import Geometry
    import torch
    D, N = 7, 600
    prior = pc.Prior([uniform(-5, 10)] * D)
    rng = np.random.default_rng(5)
    scaler = pc.Reparameterize(D, bounds=prior.bounds)
    scaler.fit(prior.rvs(2000))
    x = rng.uniform(-4, 4, size=(N, D))
    u = scaler.forward(x)
    like = lambda xx: (-0.5 * np.sum(xx ** 2, axis=1), None)
    flow = pc.Flow(D, "maf3", seed=0)
    geo = Geometry()
    geo.fit(flow.forward(torch.from_numpy(u).float())[0].numpy().astype(np.float64))
    for kind in ("preconditioned_pcn", "preconditioned_rwm"):
        res = []
        for pf in (True, False):
            state = dict(u=u.copy(), x=x.copy(), logdetj=scaler.inverse(u)[1], logl=like(x)[0], logp=prior.logpdf(x),
                         beta=0.5, blobs=None)
            funcs = dict(loglike=like, logprior=prior.logpdf, scaler=scaler, flow=flow, theta_geometry=geo)
            opts = dict(n_max=7, n_steps=10 ** 6, progress_bar=None, proposal_scale=2.38 / D ** 0.5, seed=11,
                        rng_prefill=pf)
            res.append(getattr(pmcmc, kind)(state, funcs, opts))
        for k in ("u", "x", "logl", "logp", "logdetj"):
            assert np.array_equal(res[0][k], res[1][k]), (kind, k)
        assert res[0]["accept"] == res[1]["accept"]


@pytest.mark.parametrize("lanes", [1, 2])
@pytest.mark.parametrize("kind", ["preconditioned_pcn", "rwm"])
def test_pipelined_call_with_holes_in_the_likelihood(kind, lanes):
    """-inf and NaN regions of the likelihood and proposals outside the prior's support (mcmc.py:100-121, :134) on
    the pipelined paths (device-side adaptation, lanes): the same call as with host-side adaptation on the whole
    set, rejected proposals included."""
    from scipy.stats import uniform
    import pocomc_amd as pc
    from pocomc_amd import mcmc as pmcmc
    from pocomc_amd.geometry import Geometry
    import torch
    D, N = 5, 800
    prior = pc.Prior([uniform(-3, 6)] * D)                 # narrow support: many proposals fall outside
    rng = np.random.default_rng(17)
    scaler = pc.Reparameterize(D, bounds=np.array([[-10.0, 10.0]] * D))   # wider than the support: x' leaves it
    x = rng.uniform(-2.5, 2.5, size=(N, D))
    scaler.fit(x)
    u = scaler.forward(x)

    def like(xx):
        ll = -0.5 * np.sum(xx ** 2, axis=1)
        ll = np.where(xx[:, 0] > 1.5, -np.inf, ll)
        ll = np.where(xx[:, 1] < -1.8, np.nan, ll)
        return ll, None
    logl0 = np.nan_to_num(like(x)[0], nan=-1e3, neginf=-1e3)
    flow = pc.Flow(D, "maf3", seed=0)
    geo = Geometry()
    geo.fit(flow.forward(torch.from_numpy(u).float())[0].numpy().astype(np.float64))
    geo.normal_cov = np.cov(u.T)
    res = []
    for opts_extra in (dict(pipeline=False), dict(lanes=lanes)):
        state = dict(u=u.copy(), x=x.copy(), logdetj=scaler.inverse(u)[1], logl=logl0.copy(), logp=prior.logpdf(x),
                     beta=0.7, blobs=None)
        funcs = dict(loglike=like, logprior=prior.logpdf, scaler=scaler, flow=flow, theta_geometry=geo, u_geometry=geo)
        opts = dict(n_max=6, n_steps=10 ** 6, progress_bar=None, proposal_scale=2.38 / D ** 0.5, seed=4, x_order="F",
                    **opts_extra)
        res.append(getattr(pmcmc, kind)(state, funcs, opts))
    a, b = res
    assert a["calls"] == b["calls"] < 6 * N                # some proposals never reached the likelihood
    np.testing.assert_allclose(a["accept"], b["accept"], rtol=1e-12)
    same = np.isclose(a["u"], b["u"], rtol=1e-9, atol=1e-12).all(axis=1)
    assert same.mean() >= 0.995
    assert np.isfinite(b["logl"]).all() and np.isfinite(b["logp"]).all()     # nothing invalid was accepted
    moved = ~np.isclose(b["x"], x).all(axis=1)
    assert moved.any() and (b["x"][moved, 0] <= 1.5).all() and (np.abs(b["x"]) <= 3).all()


@pytest.mark.parametrize("flow_name", ["maf3", "nsf3"])
@pytest.mark.parametrize("kind", ["preconditioned_pcn", "preconditioned_rwm", "pcn", "rwm"])
@pytest.mark.parametrize("N,n_max", [(600, 9), (50, 1), (1000, 2)])
def test_device_side_adaptation_equals_host_side(kind, N, n_max, flow_name):
    """Pipelined kernel call (sigma / mu adapted by the accept kernel's last block, the pre-step of step k+1
    enqueued behind the accept of step k) against the plain call that adapts on the host between the steps:
    same Philox variates, same update expressions -- the only difference is sqrt(1 - sigma^2) on the device
    against Python's (1 - sigma**2)**0.5 (identical unless pow() is off by an ulp)."""
    from scipy.stats import uniform
    import pocomc_amd as pc
    from pocomc_amd import mcmc as pmcmc
    from pocomc_amd.geometry import Geometry
    import torch
    D = 7
    prior = pc.Prior([uniform(-5, 10)] * D)
    rng = np.random.default_rng(N)
    scaler = pc.Reparameterize(D, bounds=prior.bounds)
    scaler.fit(prior.rvs(2000))
    x = rng.uniform(-4, 4, size=(N, D))
    u = scaler.forward(x)
    like = lambda xx: (-0.5 * np.sum(xx ** 2, axis=1), None)
    flow = pc.Flow(D, flow_name, seed=0)
    geo = Geometry()
    geo.fit(flow.forward(torch.from_numpy(u).float())[0].numpy().astype(np.float64))
    geo.normal_cov = np.cov(u.T)
    res = []
    for pipe in (False, True):
        state = dict(u=u.copy(), x=x.copy(), logdetj=scaler.inverse(u)[1], logl=like(x)[0], logp=prior.logpdf(x),
                     beta=0.5, blobs=None)
        funcs = dict(loglike=like, logprior=prior.logpdf, scaler=scaler, flow=flow, theta_geometry=geo, u_geometry=geo)
        opts = dict(n_max=n_max, n_steps=10 ** 6, progress_bar=None, proposal_scale=2.38 / D ** 0.5, seed=11,
                    x_order="F", pipeline=pipe)
        res.append(getattr(pmcmc, kind)(state, funcs, opts))
    a, b = res
    assert a["calls"] == b["calls"] and a["steps"] == b["steps"] == n_max
    assert a["proposal_scale"] == b["proposal_scale"] and a["accept"] == b["accept"]
    same = np.isclose(a["u"], b["u"], rtol=1e-9, atol=1e-12).all(axis=1)
    assert same.mean() >= 0.995, same.mean()
    for k in ("x", "logl", "logp", "logdetj"):
        np.testing.assert_allclose(a[k][same], b[k][same], rtol=1e-8, atol=1e-10, err_msg=k)


@pytest.mark.parametrize("N,lanes,bounds,flow_name", [(333, 1, "box", "maf3"), (1000, 2, "box", "maf3"), (77, 1, "mixed", "maf3"),
                                                      (333, 1, "box", "nsf3"), (1000, 2, "mixed", "nsf3")])
def test_scaler_epilogue_of_the_sweep_equals_the_scaler_launch(monkeypatch, N, lanes, bounds, flow_name):
    """The fused proposal + inverse launch applies the scaler and the prior to its 16 walkers as an epilogue
    (``scaler_body.h``: the same element code as ``scaler_inverse_kernel``, float64 in numpy's order).  PMC_NO_FUSE=2 keeps
    the scaler a launch of its own: the whole kernel call is the same bit for bit, boundary conditions included.  The spline
    flows' fused instances (``maf_inverse_nsf2.hip``) are also compared with no fusion at all (PMC_NO_FUSE=1: proposal
    kernel, plain sweep, scaler kernel)."""
    from scipy.stats import uniform, norm
    import pocomc_amd as pc
    from pocomc_amd import mcmc as pmcmc
    from pocomc_amd.geometry import Geometry
    import torch
    D = 9
    if bounds == "box":
        prior = pc.Prior([uniform(-5, 10)] * D)
        periodic = reflective = None
    else:
        prior = pc.Prior([uniform(-5, 10)] * 4 + [norm(0.0, 2.0)] * 5)
        periodic, reflective = [1], [2]
    rng = np.random.default_rng(N)
    scaler = pc.Reparameterize(D, bounds=prior.bounds, periodic=periodic, reflective=reflective)
    scaler.fit(prior.rvs(2000))
    x = np.column_stack([rng.uniform(-4, 4, size=N) for _ in range(D)])
    u = scaler.forward(x)
    like = lambda xx: (-0.5 * np.sum(xx ** 2, axis=1), None)
    flow = pc.Flow(D, flow_name, seed=0)
    geo = Geometry()
    geo.fit(flow.forward(torch.from_numpy(u).float())[0].numpy().astype(np.float64))
    geo.normal_cov = np.cov(u.T)
    res = []
    for no_fuse in (("0", "2", "1") if flow_name.startswith("nsf") else ("0", "2")):
        monkeypatch.setenv("PMC_NO_FUSE", no_fuse)
        state = dict(u=u.copy(), x=x.copy(), logdetj=scaler.inverse(u)[1], logl=like(x)[0], logp=prior.logpdf(x),
                     beta=0.5, blobs=None)
        funcs = dict(loglike=like, logprior=prior.logpdf, scaler=scaler, flow=flow, theta_geometry=geo, u_geometry=geo)
        opts = dict(n_max=6, n_steps=10 ** 6, progress_bar=None, proposal_scale=2.38 / D ** 0.5, seed=11, x_order="F",
                    lanes=lanes)
        res.append(pmcmc.preconditioned_pcn(state, funcs, opts))
    a = res[0]
    for b in res[1:]:
        assert a["calls"] == b["calls"] and a["accept"] == b["accept"] and a["proposal_scale"] == b["proposal_scale"]
        for k in ("u", "x", "logl", "logp", "logdetj"):
            assert np.array_equal(a[k], b[k]), k


def test_cache_warmer_threads_do_not_touch_the_results():
    """option host_prefetch: helper threads read x' behind the completion word (csrc/host_prefetch.hip) -- same call bit for bit."""
    from scipy.stats import uniform
    import pocomc_amd as pc
    from pocomc_amd import mcmc as pmcmc
    from pocomc_amd.geometry import Geometry
    import torch
    D, N = 8, 700
    prior = pc.Prior([uniform(-5, 10)] * D)
    rng = np.random.default_rng(2)
    scaler = pc.Reparameterize(D, bounds=prior.bounds)
    scaler.fit(prior.rvs(2000))
    x = rng.uniform(-4, 4, size=(N, D))
    u = scaler.forward(x)
    like = lambda xx: (-0.5 * np.sum(xx ** 2, axis=1), None)
    flow = pc.Flow(D, "maf3", seed=0)
    geo = Geometry()
    geo.fit(flow.forward(torch.from_numpy(u).float())[0].numpy().astype(np.float64))
    geo.normal_cov = np.cov(u.T)
    res = []
    for k in (0, 2):
        state = dict(u=u.copy(), x=x.copy(), logdetj=scaler.inverse(u)[1], logl=like(x)[0], logp=prior.logpdf(x),
                     beta=0.5, blobs=None)
        funcs = dict(loglike=like, logprior=prior.logpdf, scaler=scaler, flow=flow, theta_geometry=geo, u_geometry=geo)
        opts = dict(n_max=8, n_steps=10 ** 6, progress_bar=None, proposal_scale=2.38 / D ** 0.5, seed=4, x_order="F",
                    lanes=2, host_prefetch=k)
        res.append(pmcmc.preconditioned_pcn(state, funcs, opts))
    a, b = res
    assert a["calls"] == b["calls"] and a["accept"] == b["accept"]
    for key in ("u", "x", "logl", "logp", "logdetj"):
        assert np.array_equal(a[key], b[key]), key


def test_host_threads_give_the_same_call():
    """option host_threads: the likelihood is evaluated on row chunks by several threads (numpy releases the GIL);
    rows are independent, so the kernel call is bit for bit the single-threaded one."""
    from scipy.stats import uniform
    import threading
    import pocomc_amd as pc
    from pocomc_amd import mcmc as pmcmc
    from pocomc_amd.geometry import Geometry
    import torch
    D, N = 6, 900
    prior = pc.Prior([uniform(-5, 10)] * D)
    rng = np.random.default_rng(8)
    scaler = pc.Reparameterize(D, bounds=prior.bounds)
    scaler.fit(prior.rvs(2000))
    x = rng.uniform(-4, 4, size=(N, D))
    u = scaler.forward(x)
    seen = set()

    def like(xx):
        seen.add(threading.get_ident())
        return -0.5 * np.sum(xx ** 2, axis=1), None
    flow = pc.Flow(D, "maf3", seed=0)
    geo = Geometry()
    geo.fit(flow.forward(torch.from_numpy(u).float())[0].numpy().astype(np.float64))
    res = []
    for th in (1, 3):
        state = dict(u=u.copy(), x=x.copy(), logdetj=scaler.inverse(u)[1], logl=like(x)[0], logp=prior.logpdf(x),
                     beta=0.5, blobs=None)
        funcs = dict(loglike=like, logprior=prior.logpdf, scaler=scaler, flow=flow, theta_geometry=geo)
        opts = dict(n_max=5, n_steps=10 ** 6, progress_bar=None, proposal_scale=2.38 / D ** 0.5, seed=3, x_order="F",
                    host_threads=th)
        seen.clear()
        res.append(pmcmc.preconditioned_pcn(state, funcs, opts))
        assert (len(seen) == 1) if th == 1 else (1 < len(seen) <= th)     # (a quick worker may take two chunks)
    for k in ("u", "x", "logl", "logp", "logdetj"):
        assert np.array_equal(res[0][k], res[1][k]), k
    assert res[0]["calls"] == res[1]["calls"]


def test_plateau_stop_with_a_pre_step_in_flight():
    """The stop rule of mcmc.py:171-180 fires while the next pre-step is already enqueued: the call returns the
    state of the last accepted step, and a second call on the same inputs gives the same answer."""
    from scipy.stats import uniform
    import pocomc_amd as pc
    from pocomc_amd import mcmc as pmcmc
    from pocomc_amd.geometry import Geometry
    import torch
    D, N = 5, 300
    prior = pc.Prior([uniform(-5, 10)] * D)
    rng = np.random.default_rng(3)
    scaler = pc.Reparameterize(D, bounds=prior.bounds)
    scaler.fit(prior.rvs(2000))
    x = rng.normal(size=(N, D)) * 0.3                     # already at the mode: logP stops improving at once
    u = scaler.forward(x)
    like = lambda xx: (-0.5 * np.sum(xx ** 2, axis=1), None)
    flow = pc.Flow(D, "maf3", seed=0)
    geo = Geometry()
    geo.fit(flow.forward(torch.from_numpy(u).float())[0].numpy().astype(np.float64))
    out = []
    for pipe in (True, True, False):
        state = dict(u=u.copy(), x=x.copy(), logdetj=scaler.inverse(u)[1], logl=like(x)[0], logp=prior.logpdf(x),
                     beta=1.0, blobs=None)
        funcs = dict(loglike=like, logprior=prior.logpdf, scaler=scaler, flow=flow, theta_geometry=geo)
        opts = dict(n_max=500, n_steps=2, progress_bar=None, proposal_scale=2.38 / D ** 0.5, seed=5, x_order="F",
                    pipeline=pipe)
        out.append(pmcmc.preconditioned_pcn(state, funcs, opts))
    assert 1 < out[0]["steps"] < 500
    assert out[0]["steps"] == out[1]["steps"] == out[2]["steps"]
    for k in ("u", "x", "logl"):
        assert np.array_equal(out[0][k], out[1][k])
        np.testing.assert_allclose(out[0][k], out[2][k], rtol=1e-8, atol=1e-10)
    assert np.array_equal(like(out[0]["x"])[0], out[0]["logl"])


@pytest.mark.parametrize("x_order", ["C", "F"])
@pytest.mark.parametrize("kind", ["preconditioned_pcn", "preconditioned_rwm", "pcn", "rwm"])
@pytest.mark.parametrize("N,lanes", [(600, 2), (1000, 3), (40, 2)])
def test_laned_kernel_call_equals_the_whole_set_call(kind, N, lanes, x_order):
    """mcmc.LanedEngine (row ranges of the walkers stepped as a pipeline, device work of one range behind the
    host likelihood of another) draws the same Philox variates as the whole-set call -- they are keyed on the
    walker index -- and differs from it only in the order the D+4 sums are added: same trajectory up to the
    last bits of sigma / mu."""
    from scipy.stats import uniform
    import pocomc_amd as pc
    from pocomc_amd import mcmc as pmcmc
    from pocomc_amd.geometry import Geometry
    import torch
    D = 7
    prior = pc.Prior([uniform(-5, 10)] * D)
    rng = np.random.default_rng(N)
    scaler = pc.Reparameterize(D, bounds=prior.bounds)
    scaler.fit(prior.rvs(2000))
    x = rng.uniform(-4, 4, size=(N, D))
    u = scaler.forward(x)
    seen = []

    def like(xx):
        seen.append(len(xx))
        return -0.5 * np.sum(xx ** 2, axis=1), None
    flow = pc.Flow(D, "maf3", seed=0)
    geo = Geometry()
    geo.fit(flow.forward(torch.from_numpy(u).float())[0].numpy().astype(np.float64))
    geo.normal_cov = np.cov(u.T)
    res = []
    for ln in (1, lanes):
        state = dict(u=u.copy(), x=x.copy(), logdetj=scaler.inverse(u)[1], logl=like(x)[0], logp=prior.logpdf(x),
                     beta=0.5, blobs=None)
        funcs = dict(loglike=like, logprior=prior.logpdf, scaler=scaler, flow=flow, theta_geometry=geo, u_geometry=geo)
        opts = dict(n_max=6, n_steps=10 ** 6, progress_bar=None, proposal_scale=2.38 / D ** 0.5, seed=11, lanes=ln,
                    x_order=x_order)
        del seen[:]
        res.append(getattr(pmcmc, kind)(state, funcs, opts))
        assert sum(seen) == res[-1]["calls"]
        assert len(seen) == 6 * ln                     # one likelihood call per lane and step
    a, b = res
    assert a["calls"] == b["calls"] and a["steps"] == b["steps"] == 6
    np.testing.assert_allclose(a["proposal_scale"], b["proposal_scale"], rtol=1e-12)
    np.testing.assert_allclose(a["accept"], b["accept"], rtol=1e-12)
    same = np.isclose(a["u"], b["u"], rtol=1e-9, atol=1e-12).all(axis=1)
    assert same.mean() >= 0.995, same.mean()
    for k in ("x", "logl", "logp", "logdetj"):
        np.testing.assert_allclose(a[k][same], b[k][same], rtol=1e-8, atol=1e-10, err_msg=k)


@pytest.mark.parametrize("kind", ["preconditioned_pcn", "preconditioned_rwm", "pcn", "rwm"])
@pytest.mark.parametrize("N,lanes", [(1000, 1), (1000, 2), (4100, 3)])
def test_the_pipeline_behind_the_c_abi_equals_the_python_pipeline_bit_for_bit(kind, N, lanes, monkeypatch):
    """``pmc_pipeline_next`` (one C call per lane and step: accept of the lane just evaluated, next pre-steps, waits)
    enqueues exactly the launches ``LanedEngine.step_pipelined`` enqueued from Python in round 2 (``PMC_C_PIPELINE=0``
    keeps that path): identical walkers, sums, scale, counts -- every kernel of ``mcmc.py:74-156``'s four variants,
    with a likelihood that rejects part of the space and a plateau stop in the middle of the call."""
    from scipy.stats import uniform
    import pocomc_amd as pc
    from pocomc_amd import mcmc as pmcmc
    from pocomc_amd.geometry import Geometry
    import torch
    D = 6
    prior = pc.Prior([uniform(-5, 10)] * D)
    rng = np.random.default_rng(N + lanes)
    scaler = pc.Reparameterize(D, bounds=prior.bounds)
    scaler.fit(prior.rvs(2000))
    x = rng.uniform(-4, 4, size=(N, D))
    u = scaler.forward(x)

    def like(xx):
        l = -0.5 * np.sum(xx ** 2, axis=1)
        l[xx[:, 0] > 3.5] = -np.inf
        return l, None
    flow = pc.Flow(D, "maf3", seed=0)
    geo = Geometry()
    geo.fit(flow.forward(torch.from_numpy(u).float())[0].numpy().astype(np.float64))
    geo.normal_cov = np.cov(u.T)
    res, used = [], []
    for c_pipe in ("1", "0"):
        monkeypatch.setenv("PMC_C_PIPELINE", c_pipe)
        made = []
        real = pmcmc.LanedEngine.start_pipeline

        def spy(self, *a, _real=real, _made=made, **k):
            _real(self, *a, **k)
            _made.append(bool(self._pipe))
        monkeypatch.setattr(pmcmc.LanedEngine, "start_pipeline", spy)
        state = dict(u=u.copy(), x=x.copy(), logdetj=scaler.inverse(u)[1], logl=like(x)[0], logp=prior.logpdf(x),
                     beta=0.5, blobs=None)
        funcs = dict(loglike=like, logprior=prior.logpdf, scaler=scaler, flow=flow, theta_geometry=geo, u_geometry=geo)
        opts = dict(n_max=40, n_steps=2, progress_bar=None, proposal_scale=2.38 / D ** 0.5, seed=5, lanes=lanes, x_order="F")
        res.append(getattr(pmcmc, kind)(state, funcs, opts))
        monkeypatch.setattr(pmcmc.LanedEngine, "start_pipeline", real)
        used.append(made)
    assert used == [[True], [False]], used                  # the first call went through pmc_pipeline_*, the second did not
    a, b = res
    assert 2 <= a["steps"] == b["steps"] and a["calls"] == b["calls"]
    for k in ("u", "x", "logl", "logp", "logdetj"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert a["proposal_scale"] == b["proposal_scale"] and a["accept"] == b["accept"] and a["efficiency"] == b["efficiency"]


@pytest.mark.parametrize("c_pipe", ["1", "0"])
@pytest.mark.parametrize("lanes", [1, 2])
def test_a_drained_pipeline_resumes_on_the_same_trajectory(lanes, c_pipe, monkeypatch):
    """``LanedEngine.resume_pipeline`` (``pmc_pipeline_start`` on the existing pipeline object): a step with ``more=False``
    enqueues no pre-step of its successor; resuming issues exactly the launches the step would have enqueued.  Six steps
    with the pipeline drained and resumed behind steps 2 and 4 leave the same walkers, bit for bit, as six steps in one go --
    what ``bench.py`` relies on to put every launch of its timed steps behind ``t0`` (closed region, round 5)."""
    from scipy.stats import uniform
    import pocomc_amd as pc
    from pocomc_amd.mcmc import LanedEngine, Adaptation
    from pocomc_amd.geometry import Geometry
    import torch
    monkeypatch.setenv("PMC_C_PIPELINE", c_pipe)
    D, N, beta, nu = 6, 1600, 0.5, 5.0
    prior = pc.Prior([uniform(-5, 10)] * D)
    rng = np.random.default_rng(31 + lanes)
    scaler = pc.Reparameterize(D, bounds=prior.bounds)
    scaler.fit(prior.rvs(2000))
    x = rng.uniform(-4, 4, size=(N, D))
    u = scaler.forward(x)
    like = lambda xx: (-0.5 * np.sum(xx ** 2, axis=1), None)
    flow = pc.Flow(D, "maf3", seed=0)
    geo = Geometry()
    geo.fit(flow.forward(torch.from_numpy(u).float())[0].numpy().astype(np.float64))
    out = []
    for drains in ((), (2, 4)):
        eng = LanedEngine("preconditioned_pcn", N, D, flow, scaler, lanes=lanes, seed=77, x_order="F", streams=False)
        assert eng.set_device_prior(prior)
        eng.load_state(u, x, scaler.inverse(u)[1], like(x)[0], prior.logpdf(x))
        eng.set_geometry(mu=geo.t_mean, cov=geo.t_cov)
        ad = Adaptation("preconditioned_pcn", D, N, n_steps=10 ** 9, n_max=10 ** 9, sigma0=2.38 / D ** 0.5, mu0=geo.t_mean,
                        logp2_0=-np.inf)
        assert eng.can_pipeline()
        eng.start_pipeline(float(ad.sigma), ad.mu, nu)
        assert bool(eng._pipe) == (c_pipe == "1")
        for k in range(1, 7):
            last = k in drains or k == 6
            _, sums = eng.step_pipelined(beta, nu, ad.coefficients(), N, prior.logpdf, like, more=not last)
            ad.update(np.array(sums, copy=True))
            if k in drains:
                eng.finish_pipeline()                       # nothing of the next step is in flight ...
                eng.resume_pipeline(nu)                      # ... until it is asked for
        eng.finish_pipeline()
        st = eng.download()
        out.append((st, float(ad.sigma), ad.mu.copy()))
    (a, sa, ma), (b, sb, mb) = out
    assert sa == sb and np.array_equal(ma, mb)
    for k in ("u", "x", "logl", "logp", "logdetj"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert not np.array_equal(a["x"], x)                     # (walkers did move)


@pytest.mark.parametrize("kind,flow_name,D", [("preconditioned_pcn", "maf3", 6), ("preconditioned_pcn", "maf6", 50),
                                              ("preconditioned_rwm", "nsf3", 6), ("pcn", None, 6), ("rwm", None, 6)])
@pytest.mark.parametrize("lanes", [1, 2])
def test_rows_that_do_not_reach_the_likelihood_are_filled_not_gathered(kind, flow_name, D, lanes):
    """``pmc_step_t.fill_rejected``: a proposal outside the prior's support (or not finite) is rejected whatever its
    likelihood (``mcmc.py:118-121``: logl' = -inf for the rows left out of ``x'[mask]``, ``:117``).  With a few such rows the
    device puts the walker's CURRENT x into their host rows of x' and the host hands the whole block to the likelihood
    instead of gathering the others: same walkers, sums and call counts as with the gather, bit for bit; the likelihood
    never sees a point outside the support; it is called on whole blocks.  Fused launch (maf3), lane sweep + scaler launch
    (maf6, D = 50), spline sweep, and the kernels without a flow."""
    from scipy.stats import uniform
    import pocomc_amd as pc
    from pocomc_amd import mcmc as pmcmc
    from pocomc_amd.geometry import Geometry
    import torch
    N = 2048
    prior = pc.Prior([uniform(-3, 6)] * D)                  # narrower than the scaler's box: x' can leave the support
    rng = np.random.default_rng(D + lanes)
    scaler = pc.Reparameterize(D, bounds=np.array([[-10.0, 10.0]] * D))
    x = rng.uniform(-2.0, 2.0, size=(N, D))
    x *= 0.5                                                    # the bulk well inside ...
    x[: N // 64] = rng.uniform(2.9, 2.99, size=(N // 64, D))    # ... and a few walkers right at the edge: < 5 % of the rows fall out
    scaler.fit(x)
    u = scaler.forward(x)
    shapes = []

    def like(xx):
        assert np.isfinite(xx).all() and (np.abs(xx) <= 3.0).all()      # (mcmc.py:117 never passes anything else)
        shapes.append(xx.shape[0])
        acc = np.zeros(xx.shape[0])
        for j in range(xx.shape[1]):                         # (column after column: the same bits whatever the layout of xx --
            acc += xx[:, j] ** 2                             #  a gathered x'[mask] is a C-ordered copy, the block is F-ordered)
        return -0.5 * acc, None
    flow = pc.Flow(D, flow_name or "maf3", seed=0)
    flow.set_params(0.25 * flow.params.cpu())               # (a tame map: the walkers' moves stay small at D = 50 too)
    geo = Geometry()
    geo.fit(flow.forward(torch.from_numpy(u).float())[0].numpy().astype(np.float64))
    geo.normal_cov = np.cov(u.T)
    res, seen = [], []
    for fill in (True, False):
        del shapes[:]
        state = dict(u=u.copy(), x=x.copy(), logdetj=scaler.inverse(u)[1], logl=like(x)[0], logp=prior.logpdf(x),
                     beta=0.5, blobs=None)
        funcs = dict(loglike=like, logprior=prior.logpdf, scaler=scaler, flow=flow, theta_geometry=geo, u_geometry=geo)
        opts = dict(n_max=8, n_steps=10 ** 6, progress_bar=None, proposal_scale=0.25 / D ** 0.5, seed=5, lanes=lanes, x_order="F",
                    fill_rejected=fill)
        res.append(getattr(pmcmc, kind)(state, funcs, opts))
        seen.append(list(shapes[1:]))
    a, b = res
    assert a["steps"] == b["steps"] == 8 and a["calls"] == b["calls"] < 8 * N      # some rows never reached the likelihood
    for k in ("u", "x", "logl", "logp", "logdetj"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert a["proposal_scale"] == b["proposal_scale"] and a["accept"] == b["accept"]
    blocks = {N} if lanes == 1 else {N // 2}
    whole = [c for c in seen[0] if c in blocks]
    assert len(whole) >= len(seen[0]) // 2 and sum(seen[0]) > a["calls"], seen[0]   # filled: whole blocks (but for steps with > 5 % out)
    assert sum(seen[1]) == b["calls"] < 8 * N                                      # gathered: exactly the rows of x'[mask]


@pytest.mark.parametrize("name", ["tpcn_n256_d10_normal", "tpcn_n128_d6_mixed_bc"])
def test_likelihoods_with_holes_match_the_oracle(name):
    """Edge cases of mcmc.py:100-134: a likelihood that is -inf on part of the space and NaN on another part
    (NaN acceptance ratios count as 0, mcmc.py:134), next to non-finite priors / proposals of the case itself.
    Same numpy stream for the oracle (the reference's algorithm) and the product (replayed variates)."""
    from pocomc_amd import mcmc as pmcmc
    c = cases.MCMC_CASES[name]

    def holes(base):
        def f(x):
            l, b = base(x)
            l = np.array(l, dtype=np.float64, copy=True)
            l[x[:, 0] > 0.8] = -np.inf
            l[(x[:, 1] < -0.9) & (x[:, 0] <= 0.8)] = np.nan
            return l, b
        return f

    out = []
    for which in ("oracle", "product"):
        state, funcs, opts, aux = oracle_case(name) if which == "oracle" else product_case(name)
        funcs["loglike"] = holes(funcs["loglike"])
        opts["n_max"] = 4
        np.random.seed(c["seed"])
        if which == "oracle":
            out.append(getattr(omcmc, c["kind"])(state, funcs, opts))
        else:
            out.append(getattr(pmcmc, c["kind"])(state, funcs, opts, replay=omcmc.LegacyStream()))
    o, p_ = out
    assert o["steps"] == p_["steps"] and abs(o["calls"] - p_["calls"]) <= 2
    # a walker never moves into a hole
    assert np.isfinite(p_["logl"]).all()
    same = np.isclose(o["logl"], p_["logl"], rtol=1e-6, atol=1e-8)
    assert same.mean() > 0.97
    close(p_["x"][same], o["x"][same], 1e-4, "x")
    np.testing.assert_allclose(p_["accept"], o["accept"], atol=0.02)
