"""``Sampler`` -- the SMC orchestrator with the surface of ``pocomc.sampler.Sampler``
(``pocomc/sampler.py:27-1062``): same constructor arguments and defaults, ``run``,
``posterior``, ``evidence``, ``results``, ``save_state`` / ``load_state``.

Host logic only: the beta bisection, dynamic ESS, bookkeeping and the user's prior / likelihood
callbacks stay in Python exactly as in the reference; every array operation on the hot path is a
call into the gfx950 kernels -- ``mcmc.*`` (mutate), ``Flow.fit / forward / sample`` (train,
evidence), ``tools.compute_logw_and_logz / effective_sample_size / unique_sample_size /
trim_weights / *_resample`` (reweight, resample), ``Reparameterize`` (warm-up).
"""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np
import torch

from . import mcmc as _mcmc
from .flow import Flow
from .geometry import Geometry
from .particles import Particles
from .scaler import Reparameterize
from .tools import (compute_logw_and_logz, effective_sample_size, multinomial_resample, systematic_resample,
                    trim_weights, unique_sample_size)


class FunctionWrapper:
    """``pocomc/tools.py:227-264``."""

    def __init__(self, f, args, kwargs):
        self.f = f
        self.args = [] if args is None else args
        self.kwargs = {} if kwargs is None else kwargs

    def __call__(self, x):
        return self.f(x, *self.args, **self.kwargs)


class _Progress:
    """Minimal stand-in for the tqdm bar of ``tools.py:189-224`` (same ``info`` dict / methods)."""

    def __init__(self, show=True, initial=0):
        self.info = {}
        self.show = show
        self.n = initial

    def update_stats(self, info):
        self.info = {**self.info, **info}

    def update_iter(self):
        self.n += 1
        if self.show:
            i = self.info
            print(f"Iter {self.n}: beta={i.get('beta', 0):.4g} calls={i.get('calls', 0)} ESS={i.get('ESS', 0)} "
                  f"logZ={i.get('logZ', 0):.4g} acc={i.get('acc', 0):.3g} steps={i.get('steps', 0)}", flush=True)

    def close(self):
        pass


class Sampler:
    def __init__(self, prior, likelihood, n_dim=None, n_effective=512, n_active=256, likelihood_args=None,
                 likelihood_kwargs=None, vectorize=False, blobs_dtype=None, periodic=None, reflective=None,
                 transform="probit", pool=None, pytorch_threads=1, flow="nsf6", train_config=None,
                 train_frequency=None, precondition=True, dynamic=True, metric="ess", n_prior=None,
                 sample="tpcn", n_steps=None, n_max_steps=None, resample="mult", output_dir=None,
                 output_label=None, random_state=None, n_ess=None, group=None, mcmc_options=None):
        # ``mcmc_options``: extra keys for the MCMC kernels' option_dict (pocomc_amd/mcmc.py), e.g.
        # dict(x_order='F') hands the likelihood x as a Fortran-ordered (n, D) array and switches the kernel call to
        # its pipelined form (adaptation on the device), dict(lanes=2) steps the walkers as two row ranges.
        # sampler.py:186-373; default flow 'nsf6' like the reference (sampler.py:169)
        #
        # ``group`` / an initialised torch.distributed default group with more than one rank: ONE PROCESS PER GPU.
        # The expensive parts shard over the ranks -- every MCMC step (walkers row-sharded, one all-reduce of D+4
        # sums per step), every likelihood call, every flow fit (data parallel, gradient all-reduce) -- while the
        # bookkeeping of the persistent pool (log-weights, beta bisection, trimming, resampling indices) is
        # replicated: after each mutation the ranks all-gather their rows, so every rank holds the same pool and
        # draws the same resampling indices from the same numpy stream (seed the ranks identically:
        # ``random_state``).  n_active must be a multiple of the number of ranks.
        if n_ess is not None:
            import warnings
            n_effective = n_ess
            warnings.warn("n_ess is deprecated. Use n_effective instead.", DeprecationWarning, stacklevel=2)
        if random_state is not None:
            np.random.seed(random_state)
            torch.manual_seed(random_state)
        self.random_state = random_state
        self.group = group
        self.mcmc_options = dict(mcmc_options or {})
        self.world, self.rank = 1, 0
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        except (ImportError, RuntimeError):
            pass
        self.prior = prior
        self.log_prior = prior.logpdf
        self.sample_prior = prior.rvs
        self.bounds = prior.bounds
        self.log_likelihood = FunctionWrapper(likelihood, likelihood_args, likelihood_kwargs)
        self.blobs_dtype = blobs_dtype
        self.have_blobs = blobs_dtype is not None
        self.n_dim = prior.dim if n_dim is None else int(n_dim)
        if n_active is None and n_effective is None:
            raise ValueError("At least one of n_active or n_effective must be provided.")
        self.n_active = int(n_effective / 2) if n_active is None else int(n_active)
        self.n_effective = int(2 * n_active) if n_effective is None else int(n_effective)
        if self.world > 1 and self.n_active % self.world:
            raise ValueError("n_active must be a multiple of the number of ranks")
        self.n_steps = int(self.n_dim // 2) if n_steps is None else int(n_steps)
        self.n_max_steps = 10 * self.n_steps if n_max_steps is None else int(n_max_steps)
        self.n_total = None
        self.n_evidence = None
        self.particles = Particles(n_active, n_dim)
        self.t = 0
        self.pool = pool
        if pool is None:
            self.distribute = map
        elif isinstance(pool, int) and pool > 1:
            from multiprocess import Pool
            self.pool = Pool(pool)
            self.distribute = self.pool.map
        else:
            self.distribute = pool.map
        self.vectorize = vectorize
        if self.vectorize and self.have_blobs:
            raise ValueError("Cannot vectorize likelihood with blobs.")
        self.u_geometry = Geometry()
        self.theta_geometry = Geometry()
        self.flow = Flow(self.n_dim, flow)
        self.train_config = dict(validation_split=0.5, epochs=5000, batch_size=np.minimum(self.n_effective // 2, 512),
                                 patience=int(self.n_dim), learning_rate=1e-3, annealing=False, gaussian_scale=None,
                                 laplace_scale=None, noise=None, shuffle=True, clip_grad_norm=1.0, verbose=0)
        if train_config is not None:
            self.train_config.update(train_config)
        self.train_frequency = (np.maximum(self.n_effective // (self.n_active * 2), 1)
                                if train_frequency is None else int(train_frequency))
        self.flow_untrained = True
        if transform not in ["probit", "logit"]:
            raise ValueError(f"Invalid transform {transform}. Options are 'probit' or 'logit'.")
        self.scaler = Reparameterize(self.n_dim, bounds=self.bounds, periodic=periodic, reflective=reflective,
                                     transform=transform)
        self.output_dir = Path("states") if output_dir is None else output_dir
        self.output_label = "pmc" if output_label is None else output_label
        self.preconditioned = precondition
        if metric not in ["ess", "uss"]:
            raise ValueError(f"Invalid metric {metric}. Options are 'ess' or 'uss'.")
        self.metric = metric
        self.dynamic = dynamic
        self.dynamic_ratio = unique_sample_size(np.ones(self.n_effective), k=self.n_active) / self.n_active
        if sample not in ["tpcn", "rwm"]:
            raise ValueError(f"Invalid sample {sample}. Options are 'tpcn' or 'rwm'.")
        self.sample = sample
        self.proposal_scale = 2.38 / self.n_dim ** 0.5
        if resample not in ["mult", "syst"]:
            raise ValueError(f"Invalid resample {resample}. Options are 'mult' or 'syst'.")
        self.resample = resample
        self.n_prior = (int(2 * np.maximum(self.n_effective // self.n_active, 1) * self.n_active) if n_prior is None
                        else int(np.maximum(n_prior / self.n_active, 1) * self.n_active))
        self.prior_samples = None
        self.logz = None
        self.logz_err = None
        self.current_particles = None
        self.warmup = True
        self.calls = 0
        self.progress = None
        self.pbar = None

    # ------------------------------------------------------------------ run
    def run(self, n_total=4096, n_evidence=4096, progress=True, resume_state_path=None, save_every=None):
        """``sampler.py:375-524``."""
        if resume_state_path is not None:
            self.load_state(resume_state_path)
            t0 = self.t
            self.progress = progress
            self.pbar = _Progress(self.progress, initial=t0)
            self.pbar.update_stats(dict(calls=self.particles.get("calls", -1), beta=self.particles.get("beta", -1),
                                        logZ=self.particles.get("logz", -1)))
        else:
            t0 = self.t
            self.progress = progress
            self.pbar = _Progress(self.progress)
            self.pbar.update_stats(dict(beta=0.0, calls=self.calls, ESS=self.n_effective, logZ=0.0, logP=0.0,
                                        acc=0.0, steps=0, eff=0.0))
        self.n_total = int(n_total)
        self.n_evidence = int(n_evidence)

        def maybe_save():
            if save_every is not None and (self.t - t0) % int(save_every) == 0 and self.t != t0:
                self.save_state(Path(self.output_dir) / f"{self.output_label}_{self.t}.state")

        if self.prior_samples is None:
            self.prior_samples = self.sample_prior(self.n_prior)
            self.scaler.fit(self.prior_samples)
        if self.warmup:                                                    # sampler.py:442-489
            for i in range(self.n_prior // self.n_active):
                maybe_save()
                x = self.prior_samples[i * self.n_active:(i + 1) * self.n_active]
                u = self.scaler.forward(x)
                logdetj = self.scaler.inverse(u)[1]
                logp = self.log_prior(x)
                logl, blobs = self._log_like_sharded(x)
                self.calls += self.n_active
                bad = np.isinf(logl)
                if np.any(bad):                                            # sampler.py:456-468
                    idx_all = np.arange(len(x))
                    src = np.random.choice(idx_all[~bad], size=int(bad.sum()), replace=True)
                    for arr in (x, u, logdetj, logp, logl) + ((blobs,) if self.have_blobs else ()):
                        arr[idx_all[bad]] = arr[src]
                self.current_particles = dict(u=u, x=x, logl=logl, logp=logp, logdetj=logdetj,
                                              logw=-1e300 * np.ones(self.n_active), blobs=blobs, iter=self.t,
                                              calls=self.calls, steps=1, efficiency=1.0, ess=self.n_effective,
                                              accept=1.0, beta=0.0, logz=0.0)
                self.particles.update(self.current_particles)
                self.pbar.update_stats(dict(calls=self.calls, beta=0.0, ESS=int(self.n_effective), logZ=0.0,
                                            logP=np.mean(logp + logl), acc=1.0, steps=1, eff=1.0))
                self.pbar.update_iter()
                self.t += 1
            self.warmup = False

        while self._not_termination(self.current_particles):               # sampler.py:492-510
            maybe_save()
            self.current_particles = self._reweight(self.current_particles)
            self.current_particles = self._train(self.current_particles)
            self.current_particles = self._resample(self.current_particles)
            self.current_particles = self._mutate(self.current_particles)
            self.particles.update(self.current_particles)

        if self.n_evidence > 0 and self.preconditioned:
            self._compute_evidence(self.n_evidence)
        else:
            _, self.logz = self.particles.compute_logw_and_logz(1.0)
            self.logz_err = None
        if save_every is not None:
            self.save_state(Path(self.output_dir) / f"{self.output_label}_final.state")
        self.pbar.close()

    # ------------------------------------------------------------ sharding
    def _my_rows(self, n):
        """This rank's contiguous share of ``n`` rows."""
        return slice(self.rank * n // self.world, (self.rank + 1) * n // self.world)

    def _gather_rows(self, local, n):
        """All-gather the ranks' row shares (``_my_rows``) back into the full array, identical on every rank."""
        if self.world == 1:
            return local
        import torch.distributed as dist
        local = np.ascontiguousarray(local)
        counts = [(r + 1) * n // self.world - r * n // self.world for r in range(self.world)]
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        mine = torch.from_numpy(local).to(dev)
        if len(set(counts)) == 1:
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(parts, mine, group=self.group)
        else:
            parts = []
            for r in range(self.world):
                buf = mine if r == self.rank else torch.empty((counts[r],) + tuple(local.shape[1:]), dtype=mine.dtype, device=dev)
                dist.broadcast(buf, src=dist.get_global_rank(self.group, r) if self.group is not None else r, group=self.group)
                parts.append(buf)
        return np.concatenate([p_.cpu().numpy() for p_ in parts], axis=0)

    def _sum_over_ranks(self, v):
        if self.world == 1:
            return v
        import torch.distributed as dist
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        t = torch.tensor([float(v)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, group=self.group)
        return type(v)(t.item())

    def _log_like_sharded(self, x):
        """The likelihood of all rows of ``x`` (identical on every rank), each rank evaluating its share."""
        if self.world == 1:
            return self._log_like(x)
        sl = self._my_rows(len(x))
        logl, blobs = self._log_like(x[sl])
        if blobs is not None:
            raise NotImplementedError("blobs are not supported by the sharded Sampler")
        return self._gather_rows(np.asarray(logl, dtype=np.float64), len(x)), None

    def _ess(self, weights):
        return effective_sample_size(weights) if self.metric == "ess" else unique_sample_size(weights)

    def _not_termination(self, current_particles):
        """``sampler.py:526-548``."""
        logw, _ = self.particles.compute_logw_and_logz(1.0)
        ess = self._ess(np.exp(logw - np.max(logw)))
        return 1.0 - current_particles.get("beta") >= 1e-4 or ess < self.n_total

    # -------------------------------------------------------------- mutate
    def _mutate(self, cp):
        """``sampler.py:550-634``."""
        sl = self._my_rows(len(cp["u"]))                      # this rank's walkers (everything when world == 1)
        state = dict(u=cp["u"][sl].copy(), x=cp["x"][sl].copy(), logdetj=cp["logdetj"][sl].copy(),
                     logp=cp["logp"][sl].copy(), logl=cp["logl"][sl].copy(), beta=cp["beta"],
                     blobs=cp["blobs"][sl].copy() if self.have_blobs else None)
        if self.world > 1 and self.have_blobs:
            raise NotImplementedError("blobs are not supported by the sharded Sampler")
        funcs = dict(loglike=self._log_like, logprior=self.log_prior, scaler=self.scaler, flow=self.flow,
                     u_geometry=self.u_geometry, theta_geometry=self.theta_geometry)
        opts = dict(n_max=self.n_max_steps, n_steps=self.n_steps, progress_bar=self.pbar,
                    proposal_scale=self.proposal_scale)
        opts.update(self.mcmc_options)
        if self.world > 1:
            opts.update(group=self.group, shard_offset=sl.start)
        kernel = {(True, "tpcn"): _mcmc.preconditioned_pcn, (True, "rwm"): _mcmc.preconditioned_rwm,
                  (False, "tpcn"): _mcmc.pcn, (False, "rwm"): _mcmc.rwm}[(bool(self.preconditioned), self.sample)]
        res = kernel(state, funcs, opts)
        n_all = len(cp["u"])
        for k in ("u", "x", "logdetj", "logl", "logp"):
            cp[k] = self._gather_rows(res[k], n_all).copy()
        if self.have_blobs:
            cp["blobs"] = res["blobs"].copy()
        cp["efficiency"] = res["efficiency"] / (2.38 / self.n_dim ** 0.5)
        cp["steps"] = res["steps"]
        cp["accept"] = res["accept"]
        cp["calls"] = cp["calls"] + self._sum_over_ranks(int(res["calls"]))
        self.calls = cp["calls"]
        self.proposal_scale = res["proposal_scale"]
        return cp

    # --------------------------------------------------------------- train
    def _train(self, cp):
        """``sampler.py:636-678``."""
        u, w = cp["u"], cp["weights"]
        if self.preconditioned and (self.t % self.train_frequency == 0 or cp["beta"] == 1.0 or self.flow_untrained):
            self.flow_untrained = False
            c = self.train_config
            # sharded: every rank fits on its share of the rows (strided, so that each share follows the same
            # weight distribution); gradients and losses are all-reduced inside fit
            ut, wt = (u, w) if self.world == 1 else (u[self.rank::self.world], w[self.rank::self.world])
            if self.world > 1:                                 # equal shares: drop the remainder rows
                m_ = len(u) // self.world
                ut, wt = ut[:m_], wt[:m_]
            as32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
            self.flow.fit(as32(ut), weights=as32(wt),
                          validation_split=c["validation_split"], epochs=c["epochs"],
                          batch_size=int(np.minimum(len(u) // 2, c["batch_size"])), gaussian_scale=c["gaussian_scale"],
                          laplace_scale=c["laplace_scale"], patience=c["patience"], learning_rate=c["learning_rate"],
                          annealing=c["annealing"], noise=c["noise"], shuffle=c["shuffle"],
                          clip_grad_norm=c["clip_grad_norm"], verbose=c["verbose"], group=self.group,
                          sharded=self.world > 1)
            theta = self.flow.forward(as32(u))[0].numpy()
            self.theta_geometry.fit(theta, weights=w)
        else:
            self.u_geometry.fit(u, weights=w)
        return cp

    # ------------------------------------------------------------ resample
    def _resample(self, cp):
        """``sampler.py:680-715``."""
        w = cp["weights"]
        if self.resample == "mult":
            idx = multinomial_resample(self.n_active, w)           # = np.random.choice(len(w), n_active, p=w)
        else:
            idx = systematic_resample(self.n_active, weights=w)
        for k in ("u", "x", "logdetj", "logl", "logp") + (("blobs",) if self.have_blobs else ()):
            cp[k] = cp[k][idx]
        return cp

    # ------------------------------------------------------------ reweight
    def _reweight(self, cp):
        """``sampler.py:717-805``: next beta by bisection on the ESS of the mixture weights."""
        self.t += 1
        self.pbar.update_iter()
        beta_prev = self.particles.get("beta", index=-1)
        beta_max, beta_min = 1.0, np.copy(beta_prev)

        # the history stays on the device for the whole bisection: a trial is two launches and four doubles back
        pool = self.particles.pool_weights()

        def weights_and_ess(beta):
            return None, (pool.ess(beta) if self.metric == "ess" else pool.uss(beta))

        w_prev, ess_prev = weights_and_ess(beta_prev)
        w_max, ess_max = weights_and_ess(beta_max)
        if ess_prev <= self.n_effective:
            beta, ess_est = beta_prev, ess_prev
            logz = self.particles.get("logz", index=-1)
        elif ess_max >= self.n_effective:
            beta, ess_est = beta_max, ess_max
            _, logz = pool.logw_and_logz(beta)
        else:
            while True:
                beta = (beta_max + beta_min) * 0.5
                _, ess_est = weights_and_ess(beta)
                if np.abs(ess_est - self.n_effective) < 0.01 * self.n_effective or beta == 1.0:
                    _, logz = pool.logw_and_logz(beta)
                    break
                elif ess_est < self.n_effective:
                    beta_max = beta
                else:
                    beta_min = beta
        self.pbar.update_stats(dict(beta=beta, ESS=int(ess_est), logZ=logz))
        logw, _ = pool.logw_and_logz(beta)
        weights = np.exp(logw - np.max(logw))
        weights /= np.sum(weights)
        if self.dynamic:                                                   # sampler.py:783-790
            n_unique_active = unique_sample_size(weights, k=self.n_active)
            if n_unique_active < self.n_active * (0.95 * self.dynamic_ratio):
                self.n_effective = int(self.n_active / n_unique_active * self.n_effective)
            elif n_unique_active > self.n_active * np.minimum(1.05 * self.dynamic_ratio, 1.0):
                self.n_effective = int(n_unique_active / self.n_active * self.n_effective)
        idx, weights = trim_weights(np.arange(len(weights)), weights, ess=0.99, bins=1000)
        for k in ("u", "x", "logdetj", "logl", "logp") + (("blobs",) if self.have_blobs else ()):
            cp[k] = self.particles.get(k, index=None, flat=True)[idx]
        cp["logz"], cp["beta"], cp["weights"], cp["ess"] = logz, beta, weights, ess_est
        return cp

    # ----------------------------------------------------------- likelihood
    def _log_like(self, x):
        """``sampler.py:807-861``."""
        if self.vectorize:
            return self.log_likelihood(x), None
        results = list(self.distribute(self.log_likelihood, x))
        try:
            blob = [l[1:] for l in results if len(l) > 1]
            if not len(blob):
                raise IndexError
            logl = np.array([float(l[0]) for l in results])
            self.have_blobs = True
        except (IndexError, TypeError):
            return np.array([float(l) for l in results]), None
        if self.blobs_dtype is not None:
            dt = self.blobs_dtype
        else:
            try:
                dt = np.atleast_1d(blob[0]).dtype
            except ValueError:
                dt = np.dtype("object")
            if dt.kind in "US":
                dt = np.dtype("object")
        blob = np.array(blob, dtype=dt)
        shape = blob.shape[1:]
        if len(shape):
            axes = np.arange(len(shape))[np.array(shape) == 1] + 1
            if len(axes):
                blob = np.squeeze(blob, tuple(axes))
        return logl, blob

    # ------------------------------------------------------------- evidence
    def evidence(self):
        return self.logz, self.logz_err

    def _compute_evidence(self, n=5_000):
        """``sampler.py:869-920``: importance sampling with the flow as proposal."""
        theta_q, logq = self.flow.sample(n)
        theta_q, logq = theta_q.cpu().numpy().astype(np.float64), logq.cpu().numpy().astype(np.float64)
        x_q, logdetj = self.scaler.inverse(theta_q)
        logp = self.log_prior(x_q)
        ok = np.isfinite(logp)
        x_q, logdetj, logq, logp = x_q[ok], logdetj[ok], logq[ok], logp[ok]
        logl, _ = self._log_like_sharded(x_q)
        logw = logl + logp + logdetj - logq
        logz = np.logaddexp.reduce(logw) - np.log(len(logw))
        dlogz = np.std([np.logaddexp.reduce(logw[np.random.choice(len(logw), len(logw))]) - np.log(len(logw))
                        for _ in range(np.maximum(n, 1000))])
        self.calls += len(logw)
        self.pbar.update_stats(dict(calls=self.calls))
        self.logz, self.logz_err = logz, dlogz
        return logz, dlogz

    # ------------------------------------------------------------ posterior
    def posterior(self, resample=False, return_blobs=False, trim_importance_weights=True, return_logw=False,
                  ess_trim=0.99, bins_trim=1_000):
        """``sampler.py:937-1010``."""
        if return_blobs and not self.have_blobs:
            raise ValueError("No blobs available.")
        samples = self.particles.get("x", flat=True)
        logl = self.particles.get("logl", flat=True)
        logp = self.particles.get("logp", flat=True)
        blobs = self.particles.get("blobs", flat=True) if return_blobs else None
        logw, _ = self.particles.compute_logw_and_logz(1.0)
        weights = np.exp(logw)
        if trim_importance_weights:
            idx, weights = trim_weights(np.arange(len(samples)), weights, ess=ess_trim, bins=bins_trim)
            samples, logl, logp, logw = samples[idx], logl[idx], logp[idx], logw[idx]
            if return_blobs:
                blobs = blobs[idx]
        if resample:
            if self.resample == "mult":
                idx = multinomial_resample(len(samples), weights)
            else:
                idx = systematic_resample(len(weights), weights=weights)
            out = (samples[idx], logl[idx], logp[idx])
            return out + ((blobs[idx],) if return_blobs else ())
        out = (samples, logw if return_logw else weights, logl, logp)
        return out + ((blobs,) if return_blobs else ())

    @property
    def results(self):
        return self.particles.compute_results()

    # ----------------------------------------------------------- checkpoint
    def __getstate__(self):
        state = self.__dict__.copy()
        for k in ("pool", "distribute", "pbar", "group"):
            state.pop(k, None)
        return state

    def save_state(self, path):
        """``sampler.py:1023-1049``: dill dump to ``*.temp``, fsync, atomic rename."""
        import dill
        print(f"Saving PMC state to {path}")
        Path(path).parent.mkdir(exist_ok=True)
        temp_path = Path(path).with_suffix(".temp")
        with open(temp_path, "wb") as f:
            dill.dump(file=f, obj=self.__getstate__())
            f.flush()
            os.fsync(f.fileno())
        os.rename(temp_path, path)

    def load_state(self, path):
        """``sampler.py:1051-1061``."""
        import dill
        with open(path, "rb") as f:
            state = dill.load(file=f)
        self.__dict__ = {**self.__dict__, **state}
