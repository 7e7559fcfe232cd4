"""``Sampler`` -- preconditioned SMC around a particle pool that never leaves the GPU.

Drop-in surface of ``pocomc.sampler.Sampler``: the constructor's arguments and defaults (``pocomc/sampler.py:154-185``,
derived values ``:244-360``) and the methods ``run``, ``posterior``, ``evidence``, ``results``, ``save_state`` /
``load_state``.  The body is this build's own orchestration:

* the history of ``u, x, logdetj, logl, logp`` lives in HBM (``particles.Particles``); one SMC iteration is

      temper   beta by bisection on the ESS of the mixture weights, evaluated on the resident ``logl`` (two launches
               and four doubles per trial); importance weights, dynamic ESS and trimming on the device
      fit      the flow on the surviving rows (gathered on the device), then the geometry of theta (``pool.hip``)
      draw     resampling indices on the device from numpy's uniforms, one ``pmc_gather`` into the walker state
      mutate   the MCMC kernel call on device-resident walkers; its result is appended to the pool device to device

  -- particle rows cross PCIe only where the contract demands it: ``x'`` to the host likelihood inside the kernel
  call, prior draws in, posterior out;
* one process per GPU (``group`` / an initialised ``torch.distributed`` group): walkers, likelihood calls and flow fits
  are sharded over the ranks, the pool bookkeeping is replicated (every rank holds the same pool and draws the same
  indices from the same seeded numpy stream), see ``DESIGN.md`` section 6;
* the evidence estimate (``sampler.py:869-920``) bootstraps its error on the device (``pmc_bootstrap_logz``).

Reference lines are cited where a formula or a default is taken from them.
"""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np
import torch

from . import _lib
from . import mcmc as _mcmc
from .flow import Flow
from .geometry import Geometry
from .particles import Particles, ROW_KEYS
from .scaler import Reparameterize
from .tools import multinomial_resample, systematic_resample, unique_sample_size

_KERNELS = {(True, "tpcn"): _mcmc.preconditioned_pcn, (True, "rwm"): _mcmc.preconditioned_rwm,
            (False, "tpcn"): _mcmc.pcn, (False, "rwm"): _mcmc.rwm}


class FunctionWrapper:
    """``pocomc/tools.py:227-264``: the likelihood with its extra arguments bound."""

    def __init__(self, f, args, kwargs):
        self.f, self.args, self.kwargs = f, ([] if args is None else args), ({} if kwargs is None else kwargs)

    def __call__(self, x):
        return self.f(x, *self.args, **self.kwargs)


class _Progress:
    """One line per SMC iteration instead of the tqdm bar of ``tools.py:189-224`` (same ``info`` dict / methods; the
    MCMC kernels update it every step like ``mcmc.py:159-167``)."""

    def __init__(self, show=True, initial=0):
        self.info, self.show, self.n = {}, show, initial

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


class _Ranks:
    """The ranks of a one-process-per-GPU run and the three exchanges the Sampler needs between them."""

    def __init__(self, group):
        self.group, self.world, self.rank = group, 1, 0
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        except (ImportError, RuntimeError):
            pass

    def _dev(self):
        import torch.distributed as dist
        return (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(self.group) == "nccl"
                else torch.device("cpu"))

    def share(self, n):
        """This rank's contiguous rows of ``n``."""
        return slice(self.rank * n // self.world, (self.rank + 1) * n // self.world)

    def gather_rows(self, local, n):
        """All-gather of the ranks' shares (tensors or numpy arrays) into the full array, identical on every rank."""
        if self.world == 1:
            return local
        import torch.distributed as dist
        as_np = not isinstance(local, torch.Tensor)
        mine = (torch.from_numpy(np.ascontiguousarray(local)) if as_np else local.contiguous())
        home = mine.device
        mine = mine.to(self._dev())
        counts = [(r + 1) * n // self.world - r * n // self.world for r in range(self.world)]
        if len(set(counts)) == 1:
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(parts, mine, group=self.group)
        else:
            parts = []
            for r in range(self.world):
                buf = mine if r == self.rank else torch.empty((counts[r],) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
                dist.broadcast(buf, src=dist.get_global_rank(self.group, r) if self.group is not None else r, group=self.group)
                parts.append(buf)
        full = torch.cat(parts, dim=0)
        return full.cpu().numpy() if as_np else full.to(home)

    def total(self, v):
        if self.world == 1:
            return v
        import torch.distributed as dist
        t = torch.tensor([float(v)], dtype=torch.float64, device=self._dev())
        dist.all_reduce(t, group=self.group)
        return type(v)(t.item())

    def same_everywhere(self, t, src=0):
        """Broadcast a tensor from rank ``src`` (in place)."""
        if self.world > 1:
            import torch.distributed as dist
            buf = t.to(self._dev())
            dist.broadcast(buf, src=dist.get_global_rank(self.group, src) if self.group is not None else src, group=self.group)
            t.copy_(buf.to(t.device))
        return t

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier(group=self.group)


class Sampler:
    def __init__(self, prior, likelihood, n_dim=None, n_effective=512, n_active=256, likelihood_args=None,
                 likelihood_kwargs=None, vectorize=False, blobs_dtype=None, periodic=None, reflective=None,
                 transform="probit", pool=None, pytorch_threads=1, flow="nsf6", train_config=None,
                 train_frequency=None, precondition=True, dynamic=True, metric="ess", n_prior=None,
                 sample="tpcn", n_steps=None, n_max_steps=None, resample="mult", output_dir=None,
                 output_label=None, random_state=None, n_ess=None, group=None, mcmc_options=None):
        """Arguments and defaults of ``pocomc/sampler.py:154-185``, plus

        ``group``         a ``torch.distributed`` process group (default: the initialised default group): one process
                          per GPU, ``n_active`` a multiple of the number of ranks, ``random_state`` required (the ranks
                          replicate the pool bookkeeping from the same numpy / torch streams);
        ``mcmc_options``  extra keys of the MCMC kernels' ``option_dict`` (``pocomc_amd/mcmc.py``), e.g.
                          ``dict(x_order='F')`` (Fortran-ordered ``x`` for the likelihood, pipelined kernel call),
                          ``dict(lanes=2)``.

        Supported ``train_config`` keys: those of ``sampler.py:287-299`` (``validation_split, epochs, batch_size,
        patience, learning_rate, annealing, gaussian_scale, laplace_scale, noise, shuffle, clip_grad_norm, verbose``),
        plus ``fit_parallel`` for a sharded Sampler: ``"data"`` (every rank fits on a share of the rows, gradients all-reduced
        before the clip: RCCL), ``"replicated"`` (every rank runs the whole fit on the replicated pool: the same bits on
        every rank, no communication) or ``"auto"`` (default: data parallel from 256 local rows per batch on -- a training
        step is a latency chain that takes the same time for 64 rows as for 512, DESIGN.md section 6, so below that the
        all-reduce is pure overhead).
        """
        if n_ess is not None:
            import warnings
            n_effective = n_ess
            warnings.warn("n_ess is deprecated. Use n_effective instead.", DeprecationWarning, stacklevel=2)
        self.ranks = _Ranks(group)
        if self.ranks.world > 1 and random_state is None:
            raise ValueError("a sharded Sampler (one process per GPU) needs random_state: every rank replicates the "
                             "pool bookkeeping from the same numpy / torch random streams")
        if random_state is not None:
            np.random.seed(random_state)
            torch.manual_seed(random_state)
        self.random_state = random_state
        self.mcmc_options = dict(mcmc_options or {})

        self.prior = prior
        self.log_prior, self.sample_prior, self.bounds = prior.logpdf, prior.rvs, prior.bounds
        self.log_likelihood = FunctionWrapper(likelihood, likelihood_args, likelihood_kwargs)
        self.blobs_dtype, self.have_blobs = blobs_dtype, blobs_dtype is not None
        self.n_dim = prior.dim if n_dim is None else int(n_dim)
        if n_active is None and n_effective is None:
            raise ValueError("At least one of n_active or n_effective must be provided.")
        self.n_active = int(n_effective / 2) if n_active is None else int(n_active)
        self.n_effective = int(2 * n_active) if n_effective is None else int(n_effective)
        if self.n_active % self.ranks.world:
            raise ValueError("n_active must be a multiple of the number of ranks")
        D = self.n_dim
        self.n_steps = D // 2 if n_steps is None else int(n_steps)                       # sampler.py:244
        self.n_max_steps = 10 * self.n_steps if n_max_steps is None else int(n_max_steps)   # :250
        self.vectorize = vectorize
        if vectorize and self.have_blobs:
            raise ValueError("Cannot vectorize likelihood with blobs.")
        self.pool, self.distribute = pool, map
        if isinstance(pool, int) and pool > 1:
            from multiprocess import Pool
            self.pool = Pool(pool)
            self.distribute = self.pool.map
        elif pool is not None and not isinstance(pool, int):
            self.distribute = pool.map

        for name, value, allowed in (("transform", transform, ("probit", "logit")), ("metric", metric, ("ess", "uss")),
                                     ("sample", sample, ("tpcn", "rwm")), ("resample", resample, ("mult", "syst"))):
            if value not in allowed:
                raise ValueError(f"Invalid {name} {value}. Options are {' or '.join(repr(a) for a in allowed)}.")
        self.metric, self.sample, self.resample, self.dynamic = metric, sample, resample, dynamic
        self.preconditioned = precondition

        # (a ready Flow -- e.g. Flow(D, MAFSpec(D, 8), precision="bf16") -- is taken as it is, like the reference takes a zuko flow)
        self.flow = flow if isinstance(flow, Flow) else Flow(D, flow)
        if self.flow.n_dim != D:
            raise ValueError("flow has the wrong number of dimensions")
        if self.ranks.world > 1:                       # replicated weights: every rank starts from rank 0's init
            self.ranks.same_everywhere(self.flow.params)
            self.flow.repack()
        self.train_config = dict(validation_split=0.5, epochs=5000, batch_size=np.minimum(self.n_effective // 2, 512),
                                 patience=D, learning_rate=1e-3, annealing=False, gaussian_scale=None,
                                 laplace_scale=None, noise=None, shuffle=True, clip_grad_norm=1.0, verbose=0,   # :287-299
                                 fit_parallel="auto")
        self.train_config.update(train_config or {})
        self.train_frequency = (np.maximum(self.n_effective // (self.n_active * 2), 1) if train_frequency is None
                                else int(train_frequency))                                                       # :305
        self.flow_untrained = True
        self.scaler = Reparameterize(D, bounds=self.bounds, periodic=periodic, reflective=reflective, transform=transform)
        self.u_geometry, self.theta_geometry = Geometry(), Geometry()
        self.proposal_scale = 2.38 / D ** 0.5                                                                    # :350
        self.dynamic_ratio = unique_sample_size(np.ones(self.n_effective), k=self.n_active) / self.n_active      # :340
        self.n_prior = (int(2 * np.maximum(self.n_effective // self.n_active, 1) * self.n_active) if n_prior is None
                        else int(np.maximum(n_prior / self.n_active, 1) * self.n_active))                       # :360
        self.output_dir = Path("states") if output_dir is None else output_dir
        self.output_label = "pmc" if output_label is None else output_label

        self.particles = Particles(self.n_active, D)
        self.walkers = None            # the current walker set: dict of device tensors (+ host blobs) and scalars
        self.prior_samples = None
        self.logz = self.logz_err = None
        self.t, self.calls, self.warmup = 0, 0, True
        self.n_total = self.n_evidence = None
        self.progress = self.pbar = None

    # ranks, as the reference-style attributes
    @property
    def world(self):
        return self.ranks.world

    @property
    def rank(self):
        return self.ranks.rank

    @property
    def group(self):
        return self.ranks.group

    # ---------------------------------------------------------------------------------------------------- run
    def run(self, n_total=4096, n_evidence=4096, progress=True, resume_state_path=None, save_every=None):
        """``sampler.py:375-524``: warm-up on prior draws, SMC iterations until beta = 1 and the pool's ESS reaches
        ``n_total``, evidence."""
        if resume_state_path is not None:
            self.load_state(resume_state_path)
        t0 = self.t
        self.progress = progress
        self.pbar = _Progress(progress, initial=t0 if resume_state_path is not None else 0)
        if resume_state_path is not None:
            self.pbar.update_stats(dict(calls=self.particles.get("calls", -1), beta=self.particles.get("beta", -1),
                                        logZ=self.particles.get("logz", -1)))
        else:
            self.pbar.update_stats(dict(beta=0.0, calls=self.calls, ESS=self.n_effective, logZ=0.0, logP=0.0, acc=0.0,
                                        steps=0, eff=0.0))
        self.n_total, self.n_evidence = int(n_total), int(n_evidence)

        def checkpoint():
            if save_every is not None and (self.t - t0) % int(save_every) == 0 and self.t != t0:
                self.save_state(Path(self.output_dir) / f"{self.output_label}_{self.t}.state")

        if self.prior_samples is None:
            self.prior_samples = self.sample_prior(self.n_prior)
            self.scaler.fit(self.prior_samples)
        if self.warmup:
            for i in range(self.n_prior // self.n_active):
                checkpoint()
                self._warm_up(self.prior_samples[i * self.n_active:(i + 1) * self.n_active])
            self.warmup = False
        while self._more():
            checkpoint()
            sel = self._temper()
            self._fit(sel)
            self._draw(sel)
            self._mutate()
            self.particles.update(self.walkers)
        if self.n_evidence > 0 and self.preconditioned:
            self._compute_evidence(self.n_evidence)
        else:
            _, self.logz = self.particles.compute_logw_and_logz(1.0)
            self.logz_err = None
        if save_every is not None:
            self.save_state(Path(self.output_dir) / f"{self.output_label}_final.state")
        self.pbar.close()

    def _warm_up(self, x):
        """One block of prior draws into the pool at beta = 0 (``sampler.py:442-489``); rows with an infinite
        likelihood are replaced by copies of finite ones (``:456-468``)."""
        u = self.scaler.forward(x)
        logdetj = self.scaler.inverse(u)[1]
        logp = self.log_prior(x)
        logl, blobs = self._log_like_all(x)
        self.calls += self.n_active
        bad = np.isinf(logl)
        if np.any(bad):
            rows = np.arange(len(x))
            src = np.random.choice(rows[~bad], size=int(bad.sum()), replace=True)
            for arr in (x, u, logdetj, logp, logl) + ((blobs,) if self.have_blobs else ()):
                arr[rows[bad]] = arr[src]
        self.walkers = dict(u=u, x=x, logl=logl, logp=logp, logdetj=logdetj, blobs=blobs, iter=self.t, calls=self.calls,
                            steps=1, efficiency=1.0, ess=self.n_effective, accept=1.0, beta=0.0, logz=0.0)
        self.particles.update(self.walkers)
        self.pbar.update_stats(dict(calls=self.calls, beta=0.0, ESS=int(self.n_effective), logZ=0.0,
                                    logP=np.mean(logp + logl), acc=1.0, steps=1, eff=1.0))
        self.pbar.update_iter()
        self.t += 1

    def _pool_size(self, stats):
        """ESS (``tools.py:56-71``) or USS (``tools.py:74-93``) of the pool's weights from ``logw_stats``."""
        return (stats[1] * stats[1]) / stats[2] if self.metric == "ess" else stats[3]

    def _more(self):
        """``sampler.py:526-548``."""
        P = self.particles
        st = P.logw_stats(1.0, k=P.P if self.metric == "uss" else 0)
        return 1.0 - self.walkers.get("beta") >= 1e-4 or self._pool_size(st) < self.n_total

    # ------------------------------------------------------------------------------------------------- temper
    def _temper(self):
        """Next inverse temperature, importance weights and the trimmed pool (``sampler.py:717-805``), all on the
        resident history.  Returns the selection ``(idx, weights)`` as device tensors."""
        self.t += 1
        self.pbar.update_iter()
        P = self.particles
        hist = P._history()
        k_uss = P.P if self.metric == "uss" else 0
        size_at = lambda b: self._pool_size(P.logw_stats(b, k=k_uss, history=hist))
        logz_at = lambda st: st[0] + np.log(st[1]) - np.log(P.P)

        lo = float(P.get("beta", index=-1))
        ess_lo, ess_hi = size_at(lo), size_at(1.0)
        if ess_lo <= self.n_effective:                     # the pool cannot afford a colder target yet
            beta, ess, logz = lo, ess_lo, P.get("logz", index=-1)
        elif ess_hi >= self.n_effective:                   # the posterior itself is affordable
            beta, ess = 1.0, ess_hi
            logz = logz_at(P.logw_stats(1.0, history=hist))
        else:                                              # bisection on [previous beta, 1], sampler.py:765-777
            hi = 1.0
            while True:
                beta = (hi + lo) * 0.5
                ess = size_at(beta)
                if np.abs(ess - self.n_effective) < 0.01 * self.n_effective or beta == 1.0:
                    logz = logz_at(P.logw_stats(beta, history=hist))
                    break
                if ess < self.n_effective:
                    hi = beta
                else:
                    lo = beta
        self.pbar.update_stats(dict(beta=beta, ESS=int(ess), logZ=logz))
        st = P.logw_stats(beta, k=self.n_active if self.dynamic else 0, history=hist)     # (leaves logw at beta resident)
        if self.dynamic:                                                                  # sampler.py:783-790
            n_unique = st[3]
            if n_unique < self.n_active * (0.95 * self.dynamic_ratio):
                self.n_effective = int(self.n_active / n_unique * self.n_effective)
            elif n_unique > self.n_active * np.minimum(1.05 * self.dynamic_ratio, 1.0):
                self.n_effective = int(n_unique / self.n_active * self.n_effective)
        _, idx, w = P.select(ess=0.99, bins=1000)                                         # trim_weights, :792
        self.walkers.update(logz=logz, beta=beta, ess=ess)
        return idx, w

    # ---------------------------------------------------------------------------------------------------- fit
    def _fit(self, sel):
        """Flow and geometry on the selected rows (``sampler.py:636-678``)."""
        idx, w = sel
        u = self.particles.rows("u")[idx]                  # (row gather on the device)
        beta = self.walkers["beta"]
        if self.preconditioned and (self.t % self.train_frequency == 0 or beta == 1.0 or self.flow_untrained):
            self.flow_untrained = False
            c = self.train_config
            u32, w32 = u.to(torch.float32), w.to(torch.float32)
            bs = int(np.minimum(len(u32) // 2, c["batch_size"]))
            how = c.get("fit_parallel", "auto")
            if how not in ("auto", "data", "replicated"):
                raise ValueError("train_config['fit_parallel'] must be 'auto', 'data' or 'replicated'")
            data_parallel = self.world > 1 and (how == "data" or (how == "auto" and bs // self.world >= 256))
            if data_parallel:
                # data parallel: every rank fits on a strided share of the rows (equal shares; the remainder rows are
                # dropped), gradients and losses are all-reduced inside fit
                m = len(u32) // self.world
                ut, wt = u32[self.rank::self.world][:m], w32[self.rank::self.world][:m]
            else:
                ut, wt = u32, w32
            self.flow.fit(ut, weights=wt, validation_split=c["validation_split"], epochs=c["epochs"],
                          batch_size=bs, gaussian_scale=c["gaussian_scale"],
                          laplace_scale=c["laplace_scale"], patience=c["patience"], learning_rate=c["learning_rate"],
                          annealing=c["annealing"], noise=c["noise"], shuffle=c["shuffle"],
                          clip_grad_norm=c["clip_grad_norm"], verbose=c["verbose"], group=self.group,
                          sharded=data_parallel)
            theta = self.flow.forward(u32)[0]              # float32 on the device (tools.py:336-340)
            self.theta_geometry.fit(theta, weights=w)
        else:
            self.u_geometry.fit(u, weights=w)

    # --------------------------------------------------------------------------------------------------- draw
    def _draw(self, sel):
        """``n_active`` walkers from the selection (``sampler.py:680-715``): indices on the device from numpy's
        uniforms (``np.random.choice(p=w)`` / the systematic scheme), one gather out of the pool."""
        idx, w = sel
        if self.resample == "mult":
            pick = multinomial_resample(self.n_active, w, device_indices=True)
        else:
            pick = systematic_resample(self.n_active, weights=w, device_indices=True)
        rows = idx[pick]
        self.walkers.update(self.particles.take(rows))
        if self.have_blobs:
            self.walkers["blobs"] = self.particles.get("blobs", flat=True)[rows.cpu().numpy()]

    # ------------------------------------------------------------------------------------------------- mutate
    def _mutate(self):
        """One MCMC kernel call on the walkers (``sampler.py:550-634``).  Sharded: every rank moves its rows and the
        ranks exchange the results, so that all hold the same walker set again."""
        w, n = self.walkers, self.n_active
        sl = self.ranks.share(n)
        state = {k: w[k][sl] for k in ROW_KEYS}
        state.update(beta=w["beta"], blobs=w["blobs"][sl].copy() if self.have_blobs else None)
        funcs = dict(loglike=self._log_like, logprior=self.log_prior, scaler=self.scaler, flow=self.flow,
                     u_geometry=self.u_geometry, theta_geometry=self.theta_geometry)
        opts = dict(n_max=self.n_max_steps, n_steps=self.n_steps, progress_bar=self.pbar,
                    proposal_scale=self.proposal_scale, device_state=True)
        opts.update(self.mcmc_options)
        if self.world > 1:
            opts.update(group=self.group, shard_offset=sl.start)
        res = _KERNELS[(bool(self.preconditioned), self.sample)](state, funcs, opts)
        for k in ROW_KEYS:
            w[k] = self.ranks.gather_rows(res[k], n)
        if self.have_blobs:
            w["blobs"] = self._gather_blobs(res["blobs"], n)
        self.calls = w["calls"] = self.calls + self.ranks.total(int(res["calls"]))
        w.update(efficiency=res["efficiency"] / (2.38 / self.n_dim ** 0.5), steps=res["steps"], accept=res["accept"],
                 iter=self.t)
        self.proposal_scale = res["proposal_scale"]

    def _gather_blobs(self, local, n):
        """The ranks' blobs (host arrays of any dtype) in walker order."""
        if self.world == 1:
            return local.copy()
        import torch.distributed as dist
        parts = [None] * self.world
        dist.all_gather_object(parts, local, group=self.group)
        return np.concatenate(parts, axis=0)

    # --------------------------------------------------------------------------------------------- likelihood
    def _log_like(self, x):
        """``sampler.py:807-861``.  Vectorised likelihood: one call on the whole block, no blobs.  Otherwise the
        likelihood is mapped over the rows (``distribute``: ``map`` or a pool's); a row's return value is either the
        log-likelihood alone or a sequence ``(logl, blob, blob, ...)`` whose tail is kept as that walker's blob."""
        if self.vectorize:
            return self.log_likelihood(x), None
        per_row = list(self.distribute(self.log_likelihood, x))

        def tail(r):
            try:
                return r[1:] if len(r) > 1 else None
            except TypeError:                                   # a bare number has no len()
                return None
        tails = [tail(r) for r in per_row]
        if not any(t is not None for t in tails):
            return np.array([float(r) for r in per_row]), None
        self.have_blobs = True
        logl = np.array([float(r[0]) for r in per_row])
        kept = [t for t in tails if t is not None]
        # element type of the stored blobs: the user's ``blobs_dtype``, else what numpy sees in the first one -- text and
        # ragged returns are kept as Python objects
        dtype = self.blobs_dtype
        if dtype is None:
            try:
                dtype = np.atleast_1d(kept[0]).dtype
            except ValueError:
                dtype = np.dtype(object)
            if dtype.kind in ("U", "S"):
                dtype = np.dtype(object)
        blobs = np.array(kept, dtype=dtype)
        unit_axes = tuple(ax for ax in range(1, blobs.ndim) if blobs.shape[ax] == 1)
        return logl, (np.squeeze(blobs, unit_axes) if unit_axes else blobs)

    def _log_like_all(self, x):
        """The likelihood of all rows of ``x`` (identical on every rank), each rank evaluating its share."""
        if self.world == 1:
            return self._log_like(x)
        sl = self.ranks.share(len(x))
        logl, blobs = self._log_like(x[sl])
        logl = self.ranks.gather_rows(np.asarray(logl, dtype=np.float64), len(x))
        return logl, (None if blobs is None else self._gather_blobs(blobs, len(x)))

    # ----------------------------------------------------------------------------------------------- evidence
    def evidence(self):
        return self.logz, self.logz_err

    def _compute_evidence(self, n=5_000, replay=None):
        """Importance sampling with the flow as proposal (``sampler.py:869-920``): ``x_q`` goes to the host for the
        likelihood (the black box lives there); the error is the spread of ``max(n, 1000)`` bootstrap replicates of
        the estimate, drawn and reduced on the device.  ``replay`` (parity tests): ``dict(z=(n, D) base draw of the
        flow, draws=(B, m) bootstrap indices)`` instead of the generators."""
        if replay is not None:
            theta_q, logq = self.flow.sample(len(replay["z"]), z=torch.as_tensor(replay["z"], dtype=torch.float32))
            n = len(replay["z"])
        else:
            theta_q, logq = self.flow.sample(n)
        x_q, logdetj = self.scaler.inverse(theta_q.cpu().numpy().astype(np.float64))
        logq = logq.cpu().numpy().astype(np.float64)
        logp = self.log_prior(x_q)
        ok = np.isfinite(logp)
        x_q, logdetj, logq, logp = x_q[ok], logdetj[ok], logq[ok], logp[ok]
        logl, _ = self._log_like_all(x_q)
        logw = logl + logp + logdetj - logq
        m = len(logw)
        lib, dev = _lib.load(), self.flow.device
        lw = torch.from_numpy(np.ascontiguousarray(logw)).to(dev)
        stats = torch.zeros(4, dtype=torch.float64, device=dev)
        ws = torch.empty(int(lib.pmc_reduce_workspace_bytes(m)), dtype=torch.uint8, device=dev)
        B = int(np.maximum(n, 1000))
        reps = torch.empty(B, dtype=torch.float64, device=dev)
        draws = None
        if replay is not None:
            draws = torch.from_numpy(np.ascontiguousarray(replay["draws"], dtype=np.int64)).to(dev)
            assert draws.ndim == 2 and draws.shape[1] == m
            if m and (int(draws.min()) < 0 or int(draws.max()) >= m):
                raise ValueError("replayed bootstrap draws must index the surviving rows: 0 <= draw < %d" % m)
            B = draws.shape[0]
            reps = torch.empty(B, dtype=torch.float64, device=dev)
        else:
            seed = int(np.random.randint(0, 2 ** 31 - 1)) * 2 ** 31 + int(np.random.randint(0, 2 ** 31 - 1))
        with torch.cuda.device(dev):
            st = _lib.stream_handle()
            _lib.check(lib.pmc_logw_stats(_lib.ptr(lw), m, 0, _lib.ptr(stats), _lib.ptr(ws), st), "pmc_logw_stats")
            if draws is not None:
                _lib.check(lib.pmc_bootstrap_logz_replay(_lib.ptr(lw), m, _lib.ptr(stats), B, _lib.ptr(draws), _lib.ptr(reps), st),
                           "pmc_bootstrap_logz_replay")
            else:
                _lib.check(lib.pmc_bootstrap_logz(_lib.ptr(lw), m, _lib.ptr(stats), B, seed, _lib.ptr(reps), st), "pmc_bootstrap_logz")
        s = stats.cpu().numpy()
        logz = s[0] + np.log(s[1]) - np.log(m)
        dlogz = float(np.std(reps.cpu().numpy()))
        self.calls += m
        self.pbar.update_stats(dict(calls=self.calls))
        self.logz, self.logz_err = logz, dlogz
        return logz, dlogz

    # ---------------------------------------------------------------------------------------------- posterior
    def posterior(self, resample=False, return_blobs=False, trim_importance_weights=True, return_logw=False,
                  ess_trim=0.99, bins_trim=1_000):
        """``sampler.py:937-1010``: the pool as weighted posterior samples (numpy out)."""
        if return_blobs and not self.have_blobs:
            raise ValueError("No blobs available.")
        P = self.particles
        P.logw_stats(1.0)
        if trim_importance_weights:
            _, idx, w = P.select(ess=ess_trim, bins=bins_trim)
        else:
            w, _, _ = P.select(ess=ess_trim, bins=bins_trim)
            idx = torch.arange(P.P, device=w.device)
        st = P._h_stats.numpy()
        logw = (P._lw[:P.P] - (st[0] + np.log(st[1])))[idx].cpu().numpy()
        idx_h = idx.cpu().numpy()
        samples, logl, logp = (P.get(k, flat=True)[idx_h] for k in ("x", "logl", "logp"))
        weights = w.cpu().numpy()
        blobs = P.get("blobs", flat=True)[idx_h] if return_blobs else None
        if resample:
            pick = (multinomial_resample(len(samples), weights) if self.resample == "mult"
                    else systematic_resample(len(weights), weights=weights))
            out = (samples[pick], logl[pick], logp[pick])
            return out + ((blobs[pick],) if return_blobs else ())
        out = (samples, logw if return_logw else weights, logl, logp)
        return out + ((blobs,) if return_blobs else ())

    @property
    def results(self):
        return self.particles.compute_results()

    # --------------------------------------------------------------------------------------------- checkpoints
    _VOLATILE = ("pool", "distribute", "pbar", "ranks", "progress")

    def __getstate__(self):
        """Plain arrays and numbers (the pool and the walkers are downloaded, the flow is its parameter vector); nothing
        that belongs to THIS process -- ranks, process group, worker pool, progress line -- is saved."""
        state = {k: v for k, v in self.__dict__.items() if k not in self._VOLATILE}
        if state.get("walkers") is not None:
            state["walkers"] = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in state["walkers"].items()}
        return state

    def save_state(self, path):
        """``sampler.py:1023-1049``: dill dump to ``*.temp``, fsync, atomic rename -- written by rank 0 only; the other
        ranks wait for the file."""
        import dill
        if self.rank == 0:
            print(f"Saving PMC state to {path}")
            Path(path).parent.mkdir(exist_ok=True)
            temp_path = Path(path).with_suffix(".temp")
            with open(temp_path, "wb") as f:
                dill.dump(file=f, obj=self.__getstate__())
                f.flush()
                os.fsync(f.fileno())
            os.rename(temp_path, path)
        self.ranks.barrier()

    def load_state(self, path):
        """``sampler.py:1051-1061``; ranks / group / pools stay those of the loading process."""
        import dill
        with open(path, "rb") as f:
            state = dill.load(file=f)
        for k in self._VOLATILE:
            state.pop(k, None)
        self.__dict__.update(state)
