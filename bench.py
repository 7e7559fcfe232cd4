"""Headline benchmark: flow-preconditioned MCMC steps/s (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (``config.workload``): the tpCN kernel of ``pocomc/mcmc.py:8-183`` on the
32-D Rosenbrock likelihood (``README.md:53-55``), prior U(-10,10)^32, 10 000 walkers per
GPU, maf3 flow (H=128), beta=0.5 -- SURVEY.md section 8(d) cfg 2/4.  A *step* is one
iteration of the kernel's ``while`` loop: propose -> flow inverse -> scaler inverse ->
[host: prior + likelihood black boxes on the compacted rows] -> Metropolis accept ->
global reductions -> adaptation; the data-dependent stop is disabled for timing.
The particle state is resident in HBM when the timed region starts.

``value`` = (walkers over all ranks x steps / wall time) / 1e4, i.e. steps/s of a
1e4-walker population; at N=1 that is plain steps/s.  Weak scaling (walkers per GPU fixed).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same table: dense bf16 MFMA peak (the 5 PF headline figure includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0
TARGET_NAMES = {'rosenbrock': 'Rosenbrock', 'gaussian': 'correlated Gaussian (0.95)', 'bimodal': 'bimodal Gaussian mixture (means +-3)', 'funnel': "Neal's funnel"}


def kernel_source_hash(root=ROOT):
    """sha256 over the HIP sources and headers the library is built from: ties a committed PMC measurement
    (profiles/*_traffic.json) to the kernels it was taken on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(root, "pocomc_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    h.update(open(os.path.join(root, "include", "pocomc_amd.h"), "rb").read())
    return h.hexdigest()[:16]


def rosenbrock(x):
    """README.md:53-55 formula, -sum_i [10 (x_2i^2 - x_2i+1)^2 + (x_2i - 1)^2], written for the layout it is handed
    (same values to ~1e-16 relative: the sum over a row is taken in a different order).

    Fortran-ordered (n, D) input (the engine's x_order='F'): every coordinate is a contiguous vector of n walkers, so
    the work is three elementwise passes and two fused square-and-sum reductions (einsum) over (D/2, n) blocks, instead
    of seven elementwise passes and a reduction.  C-ordered input (the CPU baseline's arrays): the row-wise form."""
    if x.strides[0] == x.itemsize and x.strides[1] > x.itemsize:     # (column-major, or a row range of a column-major array)
        xT = x.T                                           # (D, n) view, every row a contiguous vector
        a, b = xT[::2], xT[1::2]
        t = np.multiply(a, a)
        t -= b
        s = np.einsum("ji,ji->i", t, t)
        np.subtract(a, 1.0, out=t)
        s *= 10.0
        s += np.einsum("ji,ji->i", t, t)
        return np.negative(s, out=s)
    a, b = x[:, ::2], x[:, 1::2]
    t = a * a
    t -= b
    t *= t
    t *= 10.0
    u = a - 1.0
    u *= u
    t += u
    return -t.sum(axis=1)


def make_target(name, D):
    """The host likelihood of the benchmark: BASELINE configs[3] / north_star (Rosenbrock) or configs[1]
    (correlated Gaussian, 0.95 off the diagonal: docs/source/likelihood.ipynb cell 4)."""
    if name == "rosenbrock":
        return rosenbrock
    if name == "bimodal":
        # BASELINE configs[2] (SURVEY 8(d) cfg 3): equal-weight two-component Gaussian mixture, means +-3, unit covariance
        # |x -+ 3|^2 = |x|^2 -+ 6 sum(x) + 9 D: two row reductions of x itself instead of two passes over shifted copies
        # (the same function to 1e-15 relative; round 5 -- the two-pass form cost 590 us per step at 1e4 x 50 on the host and
        # hid the device behind it)
        def bimodal(x):
            s1 = x.sum(axis=1)
            c = 0.5 * np.einsum("ij,ij->i", x, x) + 4.5 * D
            return np.logaddexp(3.0 * s1 - c, -3.0 * s1 - c) - np.log(2.0)
        return bimodal
    if name == "funnel":
        # BASELINE configs[4] (SURVEY 8(d) cfg 5): Neal's funnel as a likelihood, x0 ~ N(0, 3^2), x_i ~ N(0, e^{x0})
        def funnel(x):
            x0 = x[:, 0]
            r = x[:, 1:]
            return -x0 * x0 / 18.0 - 0.5 * np.einsum("ij,ij->i", r, r) * np.exp(-x0) - 0.5 * (D - 1) * x0
        return funnel
    cov = 0.95 * np.ones((D, D)) + 0.05 * np.eye(D)
    icov = np.linalg.inv(cov)
    norm = -0.5 * (D * np.log(2.0 * np.pi) + np.linalg.slogdet(cov)[1])

    # cov = 0.05 I + 0.95 11^T, so its inverse is a I + b 11^T (Sherman-Morrison) and the quadratic form is two row reductions
    # instead of a [n, D] x [D, D] product on the host: x^T icov x = a |x|^2 + b (sum x)^2
    a_, b_ = float(icov[0, 0] - icov[0, 1]), float(icov[0, 1])

    def gaussian(x):
        s1 = x.sum(axis=1)
        return norm - 0.5 * (a_ * np.einsum("ij,ij->i", x, x) + b_ * s1 * s1)
    return gaussian


class UniformBox:
    """Host prior (used by the CPU baseline): product of U(low, high)."""

    def __init__(self, low, high, D):
        self.low, self.high, self.D = float(low), float(high), D
        self.bounds = np.tile(np.array([[self.low, self.high]]), (D, 1))
        self.const = -D * np.log(self.high - self.low)

    def logpdf(self, x):
        inside = x >= self.low
        inside &= x <= self.high
        return np.where(inside.all(axis=1), self.const, -np.inf)


def cpu_baseline(D, n, beta, flat, spec, x0, u0_unused, geo, sigma0, seed, max_seconds=30.0, target=rosenbrock,
                 bounds=(-10.0, 10.0)):
    """The oracle (CPU restatement of the reference's algorithm: per-step numpy float64 +
    the float32 D-pass MAF inverse) timed on the host cores, on a bounded sample."""
    from threadpoolctl import threadpool_info
    from oracle import mcmc as omcmc
    from oracle.maf import OracleMAF, TorchFlowAdapter
    from oracle.scaler import Reparameterize as OracleScaler
    prior = UniformBox(bounds[0], bounds[1], D)
    sc = OracleScaler(D, bounds=prior.bounds)
    sc.fit(x0)
    # bounded sample: shrink the population until one step is affordable, then scale linearly
    n_s = n
    maf = OracleMAF(spec, flat)
    flow = TorchFlowAdapter(maf)
    t_probe = time.perf_counter()
    maf.inverse(np.zeros((256, D), np.float32))
    per_row = (time.perf_counter() - t_probe) / 256
    while n_s > 500 and per_row * n_s * 20 > max_seconds:
        n_s //= 2
    x = x0[:n_s]
    u = sc.forward(x)
    state = dict(u=u, x=x, logdetj=sc.inverse(u)[1], logl=target(x), logp=prior.logpdf(x), beta=beta, blobs=None)
    funcs = dict(loglike=lambda xx: (target(xx), None), logprior=prior.logpdf, scaler=sc, flow=flow,
                 theta_geometry=geo)
    import torch
    from threadpoolctl import threadpool_limits
    max_threads = max([i.get("num_threads", 1) for i in threadpool_info()] + [1])
    nt0 = torch.get_num_threads()
    by_threads, steps_by_threads, secs_by_threads = {}, {}, {}

    def run(th, steps):
        opts = dict(n_max=steps, n_steps=10 ** 9, progress_bar=None, proposal_scale=sigma0)
        torch.set_num_threads(th)
        with threadpool_limits(limits=th):
            np.random.seed(seed)
            st = dict(state, u=state["u"].copy(), x=state["x"].copy(), logdetj=state["logdetj"].copy(),
                      logl=state["logl"].copy(), logp=state["logp"].copy())
            t0 = time.perf_counter()
            res = omcmc.preconditioned_pcn(st, funcs, opts)
            dt = time.perf_counter() - t0
        return res["steps"], dt

    # 1 thread = the reference's default (pytorch_threads=1, sampler.py:168), and the host's cores.  BLAS
    # threads only pay for the float32 products of the flow (1e3 x 128 operands): a few help, hundreds cost more than they
    # give (round 5: 128 threads ran 3.7x SLOWER than one) -- so the legs are 1, 8 and 32 threads, the best one is the
    # baseline `value`, all are printed.  Every leg is time-boxed through a 2-step probe: <= 20 steps, never below 5.
    legs = sorted({1, min(8, max_threads), min(32, max_threads)})
    budget = {th: max_seconds * (0.4 if th == 1 else 0.6 / max(1, len(legs) - 1)) for th in legs}
    for th in legs:
        k, dt = run(th, 2)
        steps = int(min(20, max(5, budget[th] / (dt / k))))
        k, dt = run(th, steps)
        by_threads[th] = k / dt * (n_s / 1e4)               # steps/s of a 1e4-walker population
        steps_by_threads[th], secs_by_threads[th] = k, dt
    torch.set_num_threads(nt0)
    threads = max(by_threads, key=by_threads.get)
    value = by_threads[threads]
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": value, "unit": "steps/s (1e4 walkers, 32-D)", "cores": int(threads), "kind": "port",
            "sample": f"{steps_by_threads[threads]} steps of {n_s} walkers (oracle: numpy f64 step + f32 D-pass MAF inverse, "
                      f"{threads} BLAS/torch thread(s) = the fastest of the legs {legs}, all in steps_per_s_by_threads; "
                      f"host cpu_count={os.cpu_count()}); scaled linearly to 1e4 walkers",
            "cpu_model": model, "host_cpu_count": os.cpu_count(),
            "threads_1": {"steps_per_s": by_threads[1], "steps": steps_by_threads[1], "seconds": secs_by_threads[1]},
            "legs": {str(k): {"steps_per_s": by_threads[k], "steps": steps_by_threads[k], "seconds": secs_by_threads[k]}
                     for k in legs},
            "blas_max_threads": int(max_threads),
            "steps_per_s_by_threads": {str(k): v for k, v in sorted(by_threads.items())}}


def self_launch(args):
    """``python bench.py --gpus N`` without a launcher: re-exec under ``torch.distributed.run`` with one rank per
    GPU (RCCL).  On a box with fewer GPUs than N the ranks share device 0 over gloo -- a functional check of the
    N > 1 code path, flagged ``shared_gpu`` in the output line."""
    import socket
    import subprocess
    import torch
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.device_count() < args.gpus:
        env["PMC_BENCH_BACKEND"], env["PMC_BENCH_SHARE_GPU"] = "gloo", "1"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def comm_kind_name():
    """Which kind of mailbox the sharded step's exchange runs on (pocomc_amd.mcmc.small_comm)."""
    from pocomc_amd import mcmc as pmcmc, _lib
    lib = _lib.load()
    kinds = sorted({int(lib.pmc_comm_kind(v[0])) for v in pmcmc._COMMS.values() if v[0]})
    names = {0: "uncached HBM behind hipIpc handles: peer stores over xGMI", 1: "pinned host memory in POSIX shared memory: PCIe"}
    return " / ".join(names.get(k, str(k)) for k in kinds) or "none"




def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--particles", type=int, default=10000, help="walkers per GPU")
    ap.add_argument("--dim", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-flow-bench", action="store_true", help="skip the config-5 flow sub-metric (f32 vs bf16 log_prob)")
    ap.add_argument("--host-threads", type=int, default=1, help="host threads evaluating the prior/likelihood")
    ap.add_argument("--host-prefetch", type=int, default=1,
                    help="helper threads that warm the cache with x' for the likelihood (csrc/host_prefetch.hip; they run "
                         "nothing of the likelihood, which stays on the driver thread; 0 = off)")
    ap.add_argument("--host-prior", action="store_true",
                    help="evaluate Prior.logpdf on the host (default: on the device, it is a product of scipy.stats "
                         "uniform factors, not a black box)")
    ap.add_argument("--x-order", choices=["C", "F"], default="F",
                    help="memory order of the (n, D) array handed to the host prior/likelihood")
    ap.add_argument("--no-steady-state", action="store_true",
                    help="skip the second closed region of >= 200 steps (side key steady_state)")
    ap.add_argument("--event-every", type=int, default=10,
                    help="record the HIP event pair around the flow-inverse launch on every k-th timed step (an event "
                         "pair per step costs ~5 %% of the step rate: it splits the pre-phase's back-to-back launches)")
    ap.add_argument("--target", choices=["rosenbrock", "gaussian", "bimodal", "funnel"], default="rosenbrock",
                    help="host likelihood: Rosenbrock (north_star / BASELINE configs[3], default), the 0.95-correlated "
                         "Gaussian of configs[1], the bimodal mixture of configs[2] (--dim 50 --flow maf6) or Neal's funnel of "
                         "configs[4] (--dim 128 --particles 5000 --flow custom8; prior U(-30, 30))")
    ap.add_argument("--no-pin-calibration", action="store_true",
                    help="pin the driver thread to the fixed core of its rank instead of the fastest of four probed cores")
    ap.add_argument("--no-pin", action="store_true",
                    help="do not pin the driver thread to the core it starts on (the host likelihood is single-threaded "
                         "numpy; migrations between cores cost ~8 %% and most of the run-to-run noise)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="adapt sigma / mu on the host and enqueue every pre-step after the sums of the step before "
                         "(default: adaptation on the device, the next pre-step is enqueued behind the accept)")
    ap.add_argument("--lanes", type=int, default=0,
                    help="row ranges of the walker set stepped as a pipeline (mcmc.LanedEngine: the device works on one "
                         "lane while the host evaluates the likelihood of another); 1 = the whole set at once; 0 (default) = "
                         "2 for the affine flows (a lane's proposal + sweep launch is shorter than the whole set's), 1 for "
                         "the spline flows (their sweep takes the same time for 5e3 and 1e4 walkers)")
    ap.add_argument("--first-lane", type=float, default=None,
                    help="fraction of the walkers in the first of two lanes (0.5 = equal).  The sweeps of the two lanes run one "
                         "after the other and cost the same whatever their size; a larger first lane puts more of the host "
                         "likelihood behind the second sweep: measured 2605 (0.5) / 2767 (0.65) / 2844 (0.7) / 2843 (0.8) steps/s.  "
                         "Default: 0.65; 0.75 for the flows that take the lane-per-walker sweep (config 3, f16 helpers: 762 (0.5) / "
                         "811 (0.6) / 864 (0.75) steps/s -- a lane of <= 8192 walkers is one round of that sweep whatever its size); the spline "
                         "flows: 8192 walkers (nsf6 1628 (0.65) / 1651 (0.75) / 1661 (0.82), nsf3 2425 / 2483 / 2529)")
    ap.add_argument("--flow", default="maf3", help="maf3 | maf6 | maf12 | nsf3 | nsf6 | nsf12 | customN = N-transform MAF (BASELINE configs use maf3; configs[4] is custom8 at --dim 128 --particles 5000)")
    ap.add_argument("--inverse", choices=["auto", "triangular", "naive", "solo", "duo", "lane"], default="auto")
    ap.add_argument("--precision", choices=["f32", "bf16", "f16"], default="f32",
                    help="f32: float32 flow everywhere (default, the 1e-5 path).  bf16 (BASELINE configs[4] names it) / f16: the wide flows' "
                         "inverse sweep multiplies everything left of the diagonal tile with 16-bit operands and float32 accumulation "
                         "(Flow(inverse_precision=...); the dependent chain stays float32) -- opt-in precision, tolerance stated in "
                         "tests/test_gpu_config.py")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist
    from pocomc_amd import Flow, Reparameterize
    from pocomc_amd.flow import LANE16_BOUND          # the bound Flow's own guard uses
    from pocomc_amd.geometry import Geometry
    from pocomc_amd.mcmc import StepEngine, LanedEngine, Adaptation

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # PMC_BENCH_BACKEND=gloo PMC_BENCH_SHARE_GPU=1: functional check of the N>1 code path on a 1-GPU box
    backend = os.environ.get("PMC_BENCH_BACKEND", "nccl")
    if os.environ.get("PMC_BENCH_SHARE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    # everything this process allocates and every thread it starts from here on stays on the NUMA node the GPU hangs
    # off (its PCI device's local_cpulist): pinned host buffers are first-touched where their allocator runs, and the
    # step loops below read them from a core of that node
    node_cpus = None
    if not args.no_pin:
        try:
            pr = torch.cuda.get_device_properties(local)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            cl = open(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read().strip()
            cpus = set()
            for part in cl.split(","):
                lo_, _, hi_ = part.partition("-")
                cpus.update(range(int(lo_), int(hi_ or lo_) + 1))
            cpus &= os.sched_getaffinity(0)
            if cpus:
                os.sched_setaffinity(0, cpus)
                node_cpus = sorted(cpus)
        except (OSError, ValueError, AttributeError):
            node_cpus = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    D, n, beta = args.dim, args.particles, 0.5
    p_lo, p_hi = (-30.0, 30.0) if args.target == "funnel" else (-10.0, 10.0)      # config 5's prior is U(-30, 30)
    prior = UniformBox(p_lo, p_hi, D)
    rng = np.random.default_rng(1000 + rank)
    # ---- synthetic setup (untimed): x ~ prior, scaler fitted on prior draws, flow, geometry
    fit_rng = np.random.default_rng(7)                      # identical on every rank
    x_fit = fit_rng.uniform(p_lo, p_hi, size=(2 * n, D))
    scaler = Reparameterize(D, bounds=prior.bounds)
    scaler.fit(x_fit)
    x = rng.uniform(p_lo, p_hi, size=(n, D))
    u = scaler.forward(x)
    logdetj = scaler.inverse(u)[1]
    target = make_target(args.target, D)
    logl, logp = target(x), prior.logpdf(x)
    if args.flow.startswith("custom"):                      # customN: N-transform MAF with the default hidden width (configs[4]: custom8 @ 128-D)
        from pocomc_amd.maf_spec import MAFSpec
        flow = Flow(D, MAFSpec(D, int(args.flow[6:])), seed=0, inverse_precision=args.precision)
    else:
        flow = Flow(D, args.flow, seed=0, inverse_precision=args.precision if not args.flow.startswith("nsf") else "f32")   # replicated weights
    flow.inverse_algo = {"auto": 0, "triangular": 1, "naive": 2, "solo": 6, "duo": 7, "lane": 8}[args.inverse]
    torch.manual_seed(0)                                    # same shuffles / batches on every rank
    u_fit = torch.from_numpy(scaler.forward(x_fit[:n])).float().cuda()
    from pocomc_amd.train import _train_state
    _train_state(flow)                                      # one-time host-side index maps / buffers of the Flow
    torch.cuda.synchronize()
    tf0 = time.perf_counter()
    # D <= 64: 50 epochs (a setup-time shortcut of the Sampler's fit, sampler.py:655-669: epochs = 5000, patience = D, the best
    # validation state restored at the early stop, flow.py:364-374).  At config 5 (D = 128, 2500 training rows, 1.6e6
    # parameters) the shortcut is not harmless: the validation loss is best at epoch ~5 and 1e15 at epoch 50 -- cut there
    # WITHOUT the restore, the flow's inverse overflows float32 on 58 % of the walkers' own theta = forward(u), nothing is
    # ever accepted and the likelihood is never called (scripts/config5_flow_health.py; every config-5 line up to
    # profiles/r04_a_* was measured in that state).  The wide flows therefore run the Sampler's rule to its early stop.
    fit_epochs = 50 if D <= 64 else 1000
    hist = flow.fit(u_fit, epochs=fit_epochs, batch_size=512, validation_split=0.5, patience=D, annealing=False, verbose=0)
    torch.cuda.synchronize()
    fit_s = time.perf_counter() - tf0
    flow_trained = True
    n_ep = len(hist["loss"])
    # Sampler-style fit (sampler.py:655-669): half the rows train, half validate, every epoch
    flow_fit = {"epochs": n_ep, "rows": n, "batch_size": 512, "validation_split": 0.5,
                "ms_per_epoch": fit_s / n_ep * 1e3, "rows_per_s": n * n_ep / fit_s,
                "note": "untimed setup of the step benchmark, reported as the flow-training sub-metric"}
    if world > 1:                                           # replicated weights: bit-identical on every rank
        dist.broadcast(flow.params, 0)
        flow.repack()
    # geometry of theta = flow.forward(u): fitted on rank 0's shard, broadcast (replicated input of the step)
    theta0 = flow.forward(torch.from_numpy(u).float())[0].numpy().astype(np.float64)
    geo = Geometry()
    geo.fit(theta0)
    if world > 1:
        g = torch.tensor(np.concatenate([geo.t_mean, geo.t_cov.ravel(), [geo.t_nu]]), device="cuda")
        dist.broadcast(g, 0)
        g = g.cpu().numpy()
        geo.t_mean, geo.t_cov, geo.t_nu = g[:D], g[D:D + D * D].reshape(D, D), float(g[-1])
    nu = float(geo.t_nu)
    sigma0 = 2.38 / D ** 0.5

    # numpy temporaries of the host black boxes: keep them on the heap instead of one mmap +
    # page-fault storm per call (glibc M_MMAP_THRESHOLD / M_TRIM_THRESHOLD)
    try:
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 1 << 30)
        libc.mallopt(-1, 1 << 30)
    except OSError:
        pass
    eng = StepEngine("preconditioned_pcn", n, D, flow, scaler, group=None, shard_offset=rank * n, seed=20240928,
                     x_order=args.x_order)
    eng.host_threads = args.host_threads
    host_cores = None
    from scipy.stats import uniform as sp_uniform
    from pocomc_amd import Prior
    pc_prior = Prior([sp_uniform(p_lo, p_hi - p_lo)] * D)                   # pocoMC's own prior object
    device_prior = (not args.host_prior) and eng.set_device_prior(pc_prior)
    eng.load_state(u, x, logdetj, logl, logp)
    eng.set_geometry(mu=geo.t_mean, cov=geo.t_cov)
    if getattr(flow, "_lane16", None) is not None and flow.inverse_guard_enabled:
        # as mcmc._run does at the head of a call: the guard on proposals drawn at the starting sigma / mu, every rank in
        # the reduction whatever its own state
        from pocomc_amd.mcmc import _proposal_draws
        armed = flow.inverse_precision_active != "f32"
        guard = flow.check_inverse_precision(theta=_proposal_draws("preconditioned_pcn", eng.theta32, geo, min(sigma0, 0.99), 4096),
                                             rows=4096) if armed else None
        fall = (not armed) or (guard is not None and not guard["passed"])
        if world > 1:
            fl_ = torch.tensor([1.0 if fall else 0.0], device="cuda")
            dist.all_reduce(fl_)
            fall = float(fl_.item()) > 0
        if fall and flow._desc.lane16:
            flow._desc.lane16 = None
            if flow.inverse_guard is not None:
                flow.inverse_guard["passed"] = False
    ad = Adaptation("preconditioned_pcn", D, n * world, n_steps=10 ** 9, n_max=10 ** 9, sigma0=sigma0,
                    mu0=geo.t_mean, logp2_0=-np.inf)
    # the timed region steps the same walkers as a pipeline of row ranges (what mcmc._run does from 4096 walkers
    # on); `eng` (the whole set at once) stays for the instrumented passes below
    if args.lanes <= 0:
        # (two row ranges wherever a two-wave sweep covers the flow: 512 walker sets are resident at a time, a launch of
        #  all 10000 walkers would run in two rounds)
        args.lanes = 2 if (flow.spec.tri_ok and D <= 64 and (flow.spec.univariate == "rqs" or flow.spec.nOT <= 8)) else 1
    leng = None
    pipelined = (not args.no_pipeline) and args.x_order == "F" and D <= 256
    bufsize0 = None
    if args.lanes > 1 or args.host_threads > 1:
        # numpy's buffered iterator copies a strided operand (x[:, ::2] of the likelihood) through a buffer when
        # the inner loop is shorter than its buffer size (8192 elements): 25 instead of 16.5 ns/row for calls on
        # fewer than 8192 rows (scripts/hosttest.py).  A lane of 5008 rows (a thread's chunk of 2500) stays on the
        # direct path with 1024.
        bufsize0 = np.setbufsize(1024)
    if args.first_lane is None:
        import ctypes as _ct
        # (the lane sweep takes the same time for any number of walkers up to a round: the first lane takes a full round of it,
        #  8192 walkers at two subsets per workgroup -- config 3: 808 (0.65) / 829 (0.75) / 845 (0.82) steps/s, profiles/r05_g_*)
        args.first_lane = min(0.82, 8192.0 / n) if (args.inverse in ("auto", "triangular", "lane")
                                                    and flow.lib.pmc_maf_inverse_auto_is_lane(_ct.byref(flow._desc))) else 0.65
        if flow.spec.univariate == "rqs":
            # the spline sweep is longer than the whole set's likelihood: nothing of the first lane's likelihood is left to hide,
            # the second lane's comes behind its sweep -- the first lane takes all the walker sets one round holds (512 x 16)
            args.first_lane = min(0.82, 8192.0 / n)
    if args.lanes > 1 or pipelined:
        leng = LanedEngine("preconditioned_pcn", n, D, flow, scaler, lanes=args.lanes, group=None,
                           shard_offset=rank * n, seed=20240928, x_order=args.x_order, streams=not pipelined,
                           first_fraction=args.first_lane)
        if device_prior:
            leng.set_device_prior(pc_prior)
        leng.load_state(u, x, logdetj, logl, logp)
        leng.set_geometry(mu=geo.t_mean, cov=geo.t_cov)
        ad_l = Adaptation("preconditioned_pcn", D, n * world, n_steps=10 ** 9, n_max=10 ** 9, sigma0=sigma0,
                          mu0=geo.t_mean, logp2_0=-np.inf)
        assert leng.can_pipeline() == pipelined
    loglike = lambda xx: (target(xx), None)
    t_host = [0.0]

    t_seg = {"propose_call": 0.0, "evaluate_call": 0.0, "accept_call": 0.0, "adapt": 0.0}

    def step():
        ta = time.perf_counter()
        eng.propose(ad.sigma, nu)
        th = time.perf_counter()
        eng.evaluate(pc_prior.logpdf, loglike)
        tb = time.perf_counter()
        t_host[0] += tb - th
        sums = eng.accept_reduce(beta, nu)
        tc = time.perf_counter()
        ad.update(sums)
        eng.set_mu(ad.mu)
        td = time.perf_counter()
        t_seg["propose_call"] += th - ta; t_seg["evaluate_call"] += tb - th
        t_seg["accept_call"] += tc - tb; t_seg["adapt"] += td - tc

    t_lseg = {"step_call": 0.0, "adapt": 0.0}

    def step_laned(more=True):
        ta = time.perf_counter()
        if pipelined:
            # sigma / mu adapted on the device, the next pre-steps enqueued before the host sees the sums (more = False: the
            # call's last step enqueues none, like the last step of a preconditioned_pcn call)
            _, sums = leng.step_pipelined(beta, nu, ad_l.coefficients(), n * world, pc_prior.logpdf, loglike, more=more)
            tb = time.perf_counter()
            ad_l.update(sums)
        else:
            _, sums = leng.step(ad_l.sigma, nu, beta, pc_prior.logpdf, loglike)
            tb = time.perf_counter()
            ad_l.update(sums)
            leng.set_mu(ad_l.mu)
        tc = time.perf_counter()
        t_lseg["step_call"] += tb - ta; t_lseg["adapt"] += tc - tb

    timed_step = step_laned if leng is not None else step
    roof_eng = leng.lanes[0] if leng is not None else eng          # the launch the live event pair brackets

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # pin the driver thread for the step loops only (threads created during the setup above, e.g. the BLAS
    # pool the CPU baseline uses later, keep the full affinity mask)
    affinity0 = os.sched_getaffinity(0)
    pinned_core = None
    pin_probe = None
    if not args.no_pin:
        try:
            import ctypes
            pinned_core = ctypes.CDLL("libc.so.6").sched_getcpu()
            # a core of the NUMA node the GPU hangs off (node_cpus, set up at the start)
            cand = [c for c in (node_cpus or []) if c in affinity0]
            if cand:
                pinned_core = cand[(4 + 2 * rank) % len(cand)]          # (rank, not local: ranks that share a GPU in the
                                                                          #  functional test must not share a core)
                if not args.no_pin_calibration and world == 1 and len(cand) >= 48:   # (ranks keep their fixed cores: no two on one)
                    # A core of a shared box is not always at its best: scripts/core_probe.py (14 fresh processes x 4 cores
                    # of this node, profiles/r05_l_core_probe.txt) finds one core sample in ten 10 % slower for tens of
                    # milliseconds and one in fifty 2x slower (a busy sibling); a process pinned to such a core runs EVERY
                    # step slower (rounds 4 and 5: one bench process in ten).  So the likelihood is timed for ~3 ms on four
                    # cores of the node, on the buffer the step hands it, and the driver thread takes the fastest.
                    xb = (leng.lanes[0] if leng is not None else eng)._np_x
                    probe = []
                    for off in (0, 8, 16, 32):
                        c_ = cand[(4 + 2 * rank + off) % len(cand)]
                        os.sched_setaffinity(0, {c_})
                        ts_ = []
                        for _ in range(3):
                            target(xb)
                        t_end = time.perf_counter() + 3e-3
                        while time.perf_counter() < t_end or len(ts_) < 5:
                            tq = time.perf_counter(); target(xb); ts_.append(time.perf_counter() - tq)
                        probe.append((c_, float(np.median(ts_)) * 1e6))
                    os.sched_setaffinity(0, affinity0)
                    pinned_core = min(probe, key=lambda t_: t_[1])[0]
                    pin_probe = [{"core": c_, "likelihood_us": round(u_, 1)} for c_, u_ in probe]
                # cores for the likelihood's helper threads: the next ones of the same node
                i0 = cand.index(pinned_core)
                host_cores = [cand[(i0 + 1 + j) % len(cand)] for j in range(max(0, args.host_threads - 1))]
            if pinned_core in affinity0:
                os.sched_setaffinity(0, {pinned_core})
                if args.host_threads > 1:
                    for e_ in [eng] + (leng.lanes if leng is not None else []):
                        e_.host_threads, e_.host_cores = args.host_threads, host_cores
                if args.host_prefetch > 0:
                    from pocomc_amd.mcmc import host_prefetch
                    pf_cores = [cand[(i0 + 1 + j) % len(cand)] for j in range(args.host_prefetch)]
                    pf = host_prefetch(eng.lib, args.host_prefetch, pf_cores)
                    for e_ in [eng] + (leng.lanes if leng is not None else []):
                        e_.prefetcher = pf
            else:
                pinned_core = None
        except (OSError, AttributeError):
            pinned_core = None
    # the step loops run the host likelihood on the (pinned) driver thread: keep BLAS from spawning a thread per
    # core for a 1e4 x 32 x 32 product (its spinning workers would also delay the runtime's launch path)
    from threadpoolctl import threadpool_limits
    blas_limit = threadpool_limits(limits=1)
    if leng is not None and pipelined:
        leng.start_pipeline(float(ad_l.sigma), ad_l.mu, nu)      # the pre-steps of the first step
    # events, collector pass: before the warm-up, so that nothing but the barrier stands between the warm-up steps and the
    # timed ones (a collector pass takes tens of ms in which the GPU idles and clocks down: the first timed step then took
    # 420-480 us against 297)
    lib = eng.lib
    ev_pairs = [(lib.pmc_event_create(), lib.pmc_event_create()) for _ in range(args.steps)]
    import gc
    gc.collect()
    gc.disable()          # (no collector pause inside the 6 ms the driver times at --steps 20; re-enabled behind the region)
    for _ in range(args.warmup):
        step()
    if leng is not None:                                     # (the timed path last, back to back: W warm-up steps of it)
        if args.warmup == 0 and pipelined:
            leng.finish_pipeline()                           # (drain the primed pre-steps; they are re-issued behind t0)
        for w in range(args.warmup):
            # the last warm-up step enqueues NO pre-step of timed step 1: every launch of a timed step is issued behind t0
            step_laned(more=w + 1 < args.warmup)
    for k in t_seg:
        t_seg[k] = 0.0
    for k in t_lseg:
        t_lseg[k] = 0.0
    # ---- timed region: K steps through the composite entry points; the only instrumentation is one
    #      HIP event pair per step around the flow-inverse launch (recorded inside pmc_step_pre, on the
    #      stream the kernel is launched on)
    step_times = [] if os.environ.get("PMC_BENCH_STEP_TIMES") else None      # (debugging aid: distribution of the step times)
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_THREAD) if step_times is not None else None
    ev_used = []
    barrier()
    t0 = time.perf_counter()
    if leng is not None and pipelined:
        # CLOSED region, like one preconditioned_pcn call of K steps (pocomc_amd/mcmc.py::_run): the pre-steps (proposal +
        # flow inverse + scaler / prior epilogue + hand-over of x') of timed step 1 are enqueued HERE, behind t0, and run
        # fully exposed; steps 1 .. K-1 enqueue their successor's, the K-th enqueues none.  K steps = K proposal + sweep
        # launches, K likelihoods, K accepts.
        roof_eng._step.ev_inv0, roof_eng._step.ev_inv1 = None, None
        leng.resume_pipeline(nu)
    for k in range(args.steps):
        # (the event pair brackets the sweep the step ENQUEUES -- step k + 1's; the first step's own sweep, enqueued by
        #  resume_pipeline above, and the K-th step, which enqueues none, carry no pair)
        if k % args.event_every == 0 and (k + 1 < args.steps or not (leng is not None and pipelined)):
            roof_eng._step.ev_inv0, roof_eng._step.ev_inv1 = ev_pairs[k]
            ev_used.append(ev_pairs[k])
        else:
            roof_eng._step.ev_inv0, roof_eng._step.ev_inv1 = None, None
        if leng is not None and pipelined:
            step_laned(more=k + 1 < args.steps)
        else:
            timed_step()
        if step_times is not None:
            step_times.append(time.perf_counter())
    barrier()
    dt = time.perf_counter() - t0
    dt_local = dt
    # the host clocks of the K timed steps, taken HERE: the steady-state loop below adds to the same accumulators (round 5
    # divided their sum over K + 200 steps by K: step_call read 11x a step)
    seg_region = dict(t_lseg if leng is not None else t_seg)
    # ---- the same closed region once more over >= 200 steps (side key "steady_state": the exposed first pre-step is
    #      1 / K of the region, 5 % of it at the driver's K = 20 and 0.5 % here)
    roof_eng._step.ev_inv0, roof_eng._step.ev_inv1 = None, None
    n_ss = 0 if args.no_steady_state else max(200, args.steps)
    dt_ss = None
    if n_ss:
        if step_times is not None:
            step_times_k, step_times = step_times, None
        barrier()
        ts0 = time.perf_counter()
        if leng is not None and pipelined:
            leng.resume_pipeline(nu)
        ss_pairs = []
        for k in range(n_ss):
            if not ev_used and k % args.event_every == 0 and k + 1 < n_ss and len(ss_pairs) < 8:
                # (a timed region too short to carry an event pair, e.g. --steps 1: the launch is timed here instead)
                ss_pairs.append((lib.pmc_event_create(), lib.pmc_event_create()))
                roof_eng._step.ev_inv0, roof_eng._step.ev_inv1 = ss_pairs[-1]
            elif ss_pairs:
                roof_eng._step.ev_inv0, roof_eng._step.ev_inv1 = None, None
            if leng is not None and pipelined:
                step_laned(more=k + 1 < n_ss)
            else:
                timed_step()
        roof_eng._step.ev_inv0, roof_eng._step.ev_inv1 = None, None
        ev_used = ev_used or ss_pairs
        barrier()
        dt_ss = time.perf_counter() - ts0
        if world > 1:
            tt = torch.tensor([dt_ss], device="cuda", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ss = float(tt.item())
        if os.environ.get("PMC_BENCH_STEP_TIMES"):
            step_times = step_times_k
    gc.enable()
    if leng is not None and pipelined:
        leng.resume_pipeline(nu)                                 # (the passes below step on: their first pre-steps)
    if step_times is not None and rank == 0:
        st_ = np.diff(np.array([t0] + step_times)) * 1e6
        ru1 = resource.getrusage(resource.RUSAGE_THREAD)
        # what the driver thread went through inside the timed region: a step of tens of ms with involuntary switches is a
        # preemption of the pinned core, with page faults a first touch, with neither the device / runtime
        print(f"[step times us] driver thread: involuntary switches {ru1.ru_nivcsw - ru0.ru_nivcsw}, voluntary {ru1.ru_nvcsw - ru0.ru_nvcsw}, "
              f"minor faults {ru1.ru_minflt - ru0.ru_minflt}, major {ru1.ru_majflt - ru0.ru_majflt}; slowest step #{int(st_.argmax())} "
              f"{st_.max():.0f} us", file=sys.stderr)
        if len(st_) <= 40:
            print("[step times us] all", np.round(st_, 0).tolist(), file=sys.stderr)
        print(f"[step times us] median {np.median(st_):.1f} p90 {np.percentile(st_, 90):.1f} p99 {np.percentile(st_, 99):.1f} max {st_.max():.1f} "
              f"first5 {np.round(st_[:5], 1).tolist()} slow(>1.5x median) {int((st_ > 1.5 * np.median(st_)).sum())}", file=sys.stderr)
    roof_eng._step.ev_inv0, roof_eng._step.ev_inv1 = None, None
    seg_timed = {k: v / args.steps * 1e6 for k, v in seg_region.items()}
    laned_host = None
    if leng is not None:                                    # same pipeline again with host timers
        leng.host_timers = {"wait_device": 0.0, "prior": 0.0, "likelihood": 0.0}
        for k in t_lseg:
            t_lseg[k] = 0.0
        n_lt = max(20, min(args.steps, 100))
        leng.pipeline_stats(reset=True)
        for _ in range(n_lt):
            step_laned()
        torch.cuda.synchronize()
        laned_host = {**{k: v / n_lt * 1e6 for k, v in leng.host_timers.items()},
                      **{k: v / n_lt * 1e6 for k, v in t_lseg.items()}}
        ps = leng.pipeline_stats(reset=True)
        if ps is not None:
            # the lane pipeline behind the C ABI keeps its own clocks: waits (completion words of x' and of the sums) and
            # the time it spends enqueuing; what is left of a step call is the interpreter
            laned_host["wait_device"] = (ps["wait_x"] + ps["wait_sums"]) / n_lt * 1e6
            laned_host["wait_x"] = ps["wait_x"] / n_lt * 1e6
            laned_host["wait_sums"] = ps["wait_sums"] / n_lt * 1e6
            laned_host["enqueue_accept"] = ps["enqueue_accept"] / n_lt * 1e6
            laned_host["enqueue_next_pre"] = ps["enqueue_next_pre"] / n_lt * 1e6
            laned_host["pipeline"] = "pmc_pipeline_next (C ABI)"
        laned_host["python_overhead"] = (laned_host["step_call"] - laned_host["likelihood"] - laned_host["prior"]
                                         - laned_host["wait_device"] - laned_host.get("enqueue_accept", 0.0)
                                         - laned_host.get("enqueue_next_pre", 0.0) - laned_host.get("enqueue_adapt", 0.0)
                                         - laned_host.get("wait_sums", 0.0))
        leng.host_timers = None
        if os.environ.get("PMC_BENCH_EPI_STAMPS") and rank == 0 and not hasattr(lib, "pmc_debug_set_epilogue_stamps"):
            print("[epilogue stamps] this library was built without the measurement hooks: make -C pocomc_amd/csrc clean all DEBUG_HOOKS=1",
                  file=sys.stderr)
        elif os.environ.get("PMC_BENCH_EPI_STAMPS") and rank == 0:
            # measurement only: where the fused launch of lane 0 spends its epilogue (100 MHz stamps per workgroup)
            import ctypes
            nb = (leng.lanes[0].n + 15) // 16
            stamps = torch.zeros(nb, 8, dtype=torch.int64, device="cuda")
            fn = lib.pmc_debug_set_epilogue_stamps
            fn.restype, fn.argtypes = None, [ctypes.c_void_p, ctypes.c_int64]
            fn(stamps.data_ptr(), leng.lanes[0].n)
            for _ in range(3):
                step_laned()
            torch.cuda.synchronize()
            fn(None, 0)
            st = stamps.cpu().numpy()
            t0_ = st[:, 6].min()
            us_ = lambda a: (a - t0_) / 100.0
            names = ["epilogue entry", "elements done", "rows entry", "before x' stores", "x' stores issued",
                     "fence passed", "kernel entry"]
            for sel, nm in ((slice(0, nb), "blocks"),):
                for i in (6, 0, 1, 2, 3, 4, 5):
                    v = us_(st[sel, i])
                    print(f"[epilogue stamps] {nm:12s} {names[i]:42s} min {v.min():7.1f} median {np.median(v):7.1f} p90 {np.percentile(v, 90):7.1f} max {v.max():7.1f} us", file=sys.stderr)
            cu = st[:, 7] & 0xffffffff
            xcc = (st[:, 7] >> 32) & 0xf
            key = (xcc << 16) | ((cu >> 8) & 0xff) | (((cu >> 13) & 7) << 8)       # (xcc, se, sh+cu)
            uniq, cnt = np.unique(key, return_counts=True)
            late = us_(st[:, 0])
            per = {k: c for k, c in zip(uniq, cnt)}
            solo = np.array([per[k] == 1 for k in key])
            print(f"[epilogue stamps] CUs used {len(uniq)}, blocks {nb}; sweep end (epilogue entry) median: blocks alone on their CU "
                  f"{np.median(late[solo]) if solo.any() else float('nan'):.1f} us, blocks sharing a CU {np.median(late[~solo]) if (~solo).any() else float('nan'):.1f} us",
                  file=sys.stderr)
        if pipelined:
            leng.finish_pipeline()
    inv_us_live = (float(np.mean([lib.pmc_event_elapsed_ms(a, b) for a, b in ev_used])) * 1e3 if ev_used else None)
    # the same launch without the scaler epilogue (pmc_step_t.no_fuse bit 1: the scaler as a launch of its own), a few
    # untimed steps: what the sweep alone takes -- side key of the roofline object
    inv_us_sweep_only = None
    scaler_epilogue = int(roof_eng._step.no_fuse) & 3 == 0
    if scaler_epilogue:
        roof_eng._step.no_fuse = int(roof_eng._step.no_fuse) | 2
        extra = [(lib.pmc_event_create(), lib.pmc_event_create()) for _ in range(10)]
        for pair in extra:
            roof_eng._step.ev_inv0, roof_eng._step.ev_inv1 = pair
            timed_step()
        torch.cuda.synchronize()
        roof_eng._step.ev_inv0, roof_eng._step.ev_inv1 = None, None
        roof_eng._step.no_fuse = int(roof_eng._step.no_fuse) & ~2
        inv_us_sweep_only = float(np.mean([lib.pmc_event_elapsed_ms(a, b) for a, b in extra[2:]])) * 1e3
        for a, b in extra:
            lib.pmc_event_destroy(a); lib.pmc_event_destroy(b)
    for a, b in ev_pairs:
        lib.pmc_event_destroy(a); lib.pmc_event_destroy(b)
    # ---- composite path again with host timers only (no HIP events): where the host thread's time goes
    eng.host_timers = {"wait_device": 0.0, "prior": 0.0, "likelihood": 0.0}
    for k in t_seg:
        t_seg[k] = 0.0
    n_ht = max(20, min(args.steps, 100))
    for _ in range(n_ht):
        step()
    torch.cuda.synchronize()
    composite_host = {**{k: v / n_ht * 1e6 for k, v in eng.host_timers.items()},
                      **{k: v / n_ht * 1e6 for k, v in t_seg.items()}}
    # ---- instrumented pass (same steps, fine-grained entry points + HIP events + host timers):
    #      per-kernel durations for the roofline object and the host/device breakdown
    eng.events = []
    eng.host_timers = {"wait_device": 0.0, "prior": 0.0, "likelihood": 0.0}
    t_host[0] = 0.0
    for k in t_seg:
        t_seg[k] = 0.0
    n_inst = max(20, min(args.steps, 100))
    ti0 = time.perf_counter()
    for _ in range(n_inst):
        step()
    torch.cuda.synchronize()
    dt_inst = time.perf_counter() - ti0
    blas_limit.restore_original_limits()
    if bufsize0 is not None:
        np.setbufsize(bufsize0)
    if pinned_core is not None:
        os.sched_setaffinity(0, affinity0)
    per_rank = None
    if world > 1:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # every rank's own clocks, so that a multi-GPU line explains itself: wait_sums is the host's wait for the closing
        # launch of a step -- at world > 1 that launch holds the exchange (comm_adapt_kernel polls the peers' sequence words),
        # so a rank that waits longer there than the others waited for a LATE PEER, not for its own device
        mine = {"rank": rank, "device": torch.cuda.current_device(), "pinned_core": pinned_core, "timed_region_s": dt_local,
                **{k: (laned_host or {}).get(k) for k in ("wait_x", "wait_sums", "likelihood", "enqueue_accept", "enqueue_next_pre",
                                                           "python_overhead", "step_call")}}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    # ---- per-kernel device times from the HIP events recorded inside the timed region
    ev = eng.events
    seg = lambda a, b: float(np.mean([e[a].elapsed_time(e[b]) for e in ev])) * 1e3     # us
    us = {"propose": seg(0, 1), "maf_inverse": seg(1, 2), "scaler_inverse": seg(2, 3), "d2h_x": seg(3, 4),
          "accept_reduce": seg(5, 6)}
    spec = flow.spec
    n_launch = roof_eng.n                                         # walkers of the launch the event pair brackets
    algo_flops = n_launch * spec.flops_inverse_naive()            # (D+1)*F_fwd per walker
    actual_flops = n_launch * 2 * spec.macs_masked()              # what the triangular sweep needs
    if inv_us_live is None:
        inv_us_live = us["maf_inverse"]                     # (--steps 1 --no-steady-state: the instrumented pass's figure)
    us["maf_inverse_timed_region"] = inv_us_live            # HIP events inside the timed region
    t_inv = inv_us_live * 1e-6
    lane_auto = args.inverse in ("auto", "triangular") and bool(lib.pmc_maf_inverse_auto_is_lane(ctypes.byref(flow._desc)))
    nsf2 = spec.univariate == "rqs" and bool(lib.pmc_maf_inverse_auto_is_nsf2(ctypes.byref(flow._desc)))
    fused = (eng.pre and spec.tri_ok and not lane_auto and args.inverse in ("auto", "triangular")
             and ((spec.univariate == "affine" and spec.nOT <= 8) or nsf2))
    duo = bool(lib.pmc_maf_inverse_auto_is_duo(ctypes.byref(flow._desc), n_launch))
    roof_kernel = ("maf_dense_kernel<1>" if (args.inverse == "naive" or not spec.tri_ok) else
                   "maf_inverse_nsf2_kernel" if (nsf2 and args.inverse in ("auto", "triangular", "duo")) else
                   "maf_inverse_tri_nsf_kernel" if spec.univariate == "rqs" else
                   "maf_inverse_tri6_kernel" if (args.inverse == "lane" or lane_auto or os.environ.get("PMC_INVERSE_LANE", "0") != "0") else
                   "maf_inverse_tri4_kernel" if args.inverse == "solo" else
                   "maf_inverse_tri5_kernel" if (args.inverse == "duo" or duo) else "maf_inverse_tri4_kernel")
    # HBM bytes per launch of the dominant kernel: PMC passes cannot run inside this process; the value is the
    # committed rocprofv3 measurement of this very command (scripts/collect_profile.sh -> profiles/*_traffic.json),
    # used only when it was taken on the kernel sources this run was built from (their hash is stored with it) and
    # for the same kernel / walkers per launch -- otherwise null
    traffic = None
    try:
        src_hash = kernel_source_hash(ROOT)
        for cand in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json")), reverse=True):
            pm = json.load(open(os.path.join(ROOT, "profiles", cand)))
            if (pm.get("kernel_source_hash") == src_hash and n == 10000 and D == 32 and args.inverse == "auto"
                    and args.flow == "maf3" and pm.get("kernel", "").startswith(roof_kernel)
                    and pm.get("walkers_per_launch") == n_launch):
                traffic = {"hbm_bytes_per_launch": pm["hbm_bytes_per_launch"], "unit": "B", "source": pm["source"], "file": "profiles/" + cand,
                           "correction": pm["correction"], "algorithmic_bytes": pm.get("algorithmic_bytes", {}).get("total"),
                           "kernel_source_hash": src_hash, "note": pm.get("note")}
                break
    except (OSError, KeyError, ValueError, ImportError):
        pass
    # the roofline fraction is what the kernel EXECUTES (the masked multiply-adds, once) over the f32 MFMA peak; the
    # SURVEY 8(d) naive-equivalent figure ((D+1) dense passes per transform, which this algorithm does not perform) is
    # kept as a side key
    achieved = actual_flops / t_inv / 1e12
    lane16 = args.precision != "f32" and roof_kernel == "maf_inverse_tri6_kernel" and bool(flow._desc.lane16)
    peak = PEAK_BF16_MFMA_TFLOPS if lane16 else PEAK_F32_MFMA_TFLOPS
    roofline = {"bound": "mfma", "kernel": roof_kernel, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak,
                "peak_note": ("dense bf16 / f16 MFMA peak: the left-looking products (the bulk of the executed flops) run on "
                              "v_mfma_f32_16x16x32_bf16 / _f16; the dependent chain (diagonal tiles) stays on v_mfma_f32_4x4x1_16b_f32"
                              if lane16 else "float32 MFMA peak (= vector peak)"),
                # HBM bytes per launch come from PMC counters, which cannot be collected inside this process: null here;
                # the committed rocprofv3 PMC measurement of this very command (same kernel sources, by hash) is kept under
                # `traffic_committed_profile` for reference
                "traffic": None,
                "traffic_committed_profile": traffic,
                "avg_launch_us": inv_us_live, "launches_timed": len(ev_used),
                "walkers_per_launch": n_launch,
                "flops_per_launch": actual_flops,
                "note": "achieved = executed flops (2 x the unmasked multiply-adds of the flow, each once: the triangular "
                        "sweep) / launch time, measured with a HIP event pair inside the timed region"
                        + ("; the launch also proposes theta' for its walkers (fused proposal prologue)" if fused else "")
                        + ("; and applies the scaler + prior to them and stores x' to pinned host memory over PCIe (fused "
                           "epilogue, ~1.3 MB per launch at ~40 GB/s): that part of the launch executes no flow flops -- "
                           "sweep_only is the same launch without it" if (fused and inv_us_sweep_only) else ""),
                "fused_proposal": bool(fused),
                "fused_scaler_epilogue": bool(fused and inv_us_sweep_only),
                "sweep_only": None if not (fused and inv_us_sweep_only) else
                {"avg_launch_us": inv_us_sweep_only, "achieved": actual_flops / inv_us_sweep_only / 1e6,
                 "frac": actual_flops / inv_us_sweep_only / 1e6 / PEAK_F32_MFMA_TFLOPS,
                 "note": "proposal + sweep without the scaler epilogue (10 untimed steps with pmc_step_t.no_fuse = 2)"},
                "naive_equivalent": {"flops_per_launch": algo_flops, "tflops": algo_flops / t_inv / 1e12,
                                     "note": "SURVEY 8(d) F_inv = (D+1) F_fwd per walker: zuko's D-pass algorithm, not executed"}}
    # HBM-nominal sweeps: achieved GB/s against ~8 TB/s.  At this size every array (1.28 MB f32 /
    # 2.56 MB f64) is cache-resident, so these kernels are launch / latency limited -- reported with that caveat.
    b_accept = n * 4 * (6 * D + 13)                           # SURVEY figure (fp32 state)
    b_accept_f64 = n * (8 * (4 * D + 9) + 4 * 2 * D)          # what this build moves: f64 u, x (+theta f32/f64), 9 scalars
    b_scaler = n * (8 * D + 4)
    sweeps = {"bound": "hbm", "peak": 8000.0, "unit": "GB/s",
              "accept_kernel": {"avg_launch_us": us["accept_reduce"], "algorithmic_bytes": b_accept,
                                "achieved": b_accept / us["accept_reduce"] * 1e-3,
                                "frac": b_accept / us["accept_reduce"] * 1e-3 / 8000.0,
                                "bytes_moved_f64_state": b_accept_f64},
              "scaler_inverse_kernel": {"avg_launch_us": us["scaler_inverse"], "algorithmic_bytes": b_scaler,
                                        "achieved": b_scaler / us["scaler_inverse"] * 1e-3,
                                        "frac": b_scaler / us["scaler_inverse"] * 1e-3 / 8000.0},
              "note": "state is cache-resident at 1e4 x 32: launch/latency bound, not bandwidth bound (SURVEY 8(d) caveat); "
                      "durations from the instrumented pass (fine-grained launches with device-to-host copies)"}
    # resample sweep (sampler.py:680-715) on a config-4 sized persistent pool: multinomial indices + gather of 1e4 rows
    # out of 8e4 pooled particles, f64 state (what this build moves; SURVEY counts 4 B elements)
    try:
        P_pool = 8 * n
        g = torch.Generator(device="cuda").manual_seed(5)
        pu = torch.randn(P_pool, D, dtype=torch.float64, device="cuda", generator=g)
        px = torch.randn(P_pool, D, dtype=torch.float64, device="cuda", generator=g)
        ps = [torch.randn(P_pool, dtype=torch.float64, device="cuda", generator=g) for _ in range(3)]
        wts = torch.rand(P_pool, dtype=torch.float64, device="cuda", generator=g); wts /= wts.sum()
        unif = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
        cdf = torch.empty(P_pool, dtype=torch.float64, device="cuda")
        ridx = torch.empty(n, dtype=torch.int64, device="cuda")
        ou, ox = torch.empty(n, D, dtype=torch.float64, device="cuda"), torch.empty(n, D, dtype=torch.float64, device="cuda")
        os_ = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(3)]
        from pocomc_amd import _lib as _L
        st_h = _L.stream_handle()

        def resample_once():
            _L.check(lib.pmc_resample_multinomial(_L.ptr(wts), P_pool, _L.ptr(unif), n, _L.ptr(cdf), _L.ptr(ridx), st_h))

        def gather_once():
            _L.check(lib.pmc_gather(_L.ptr(ridx), n, D, _L.ptr(pu), _L.ptr(px), _L.ptr(ps[0]), _L.ptr(ps[1]), _L.ptr(ps[2]),
                                    _L.ptr(ou), _L.ptr(ox), _L.ptr(os_[0]), _L.ptr(os_[1]), _L.ptr(os_[2]), st_h))

        def timed(fn, reps=10):
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3
        t_res, t_gat = timed(resample_once), timed(gather_once)
        b_gather = n * (2 * D + 3) * 8 * 2 + n * 8                       # rows read + written (f64) + indices
        sweeps["resample_gather"] = {"pool": P_pool, "rows_out": n, "indices_us": t_res, "gather_us": t_gat,
                                     "algorithmic_bytes": 4 * n * (3 * D + 3), "bytes_moved_f64_state": b_gather,
                                     "achieved": b_gather / t_gat * 1e-3, "frac": b_gather / t_gat * 1e-3 / 8000.0,
                                     "note": "achieved / frac use the f64 bytes this build moves for the gather launch"}
        del pu, px, ps, ou, ox, os_, cdf
    except Exception as exc:                                              # (a sub-metric must not take the bench down)
        sweeps["resample_gather"] = {"error": repr(exc)}
    # BASELINE configs[4] flow (128-D, 8 transforms, H = 512; "8-layer MAF bf16"): log_prob on 5000 rows per GPU with the
    # float32 and the bf16 matrix-core kernels -- rows/s and executed-flop fraction of the respective dense MFMA peak
    flow_cfg5 = None
    if rank == 0 and not args.no_flow_bench:
        try:
            from pocomc_amd.maf_spec import MAFSpec
            sp5 = MAFSpec(128, 8)
            x5 = torch.randn(5000, 128, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
            flow_cfg5 = {"workload": "log_prob of 5000 rows, 128-D MAF, 8 transforms, H=512 (BASELINE configs[4] per-GPU shard)",
                         "flops_per_row_executed": 2 * sp5.macs_masked(), "flops_per_row_dense": sp5.flops_forward_dense()}
            for prec, peak in (("f32", PEAK_F32_MFMA_TFLOPS), ("bf16", PEAK_BF16_MFMA_TFLOPS)):
                f5 = Flow(128, sp5, seed=0, precision=prec)
                for _ in range(3):
                    f5.log_prob(x5)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    f5.log_prob(x5)
                e1.record()
                torch.cuda.synchronize()
                us5 = e0.elapsed_time(e1) / 10 * 1e3
                tf = 5000 * 2 * sp5.macs_masked() / us5 / 1e6
                flow_cfg5[prec] = {"us_per_call": us5, "rows_per_s": 5000 / us5 * 1e6, "achieved_tflops": tf, "peak_tflops": peak,
                                   "frac": tf / peak, "dense_equivalent_tflops": 5000 * sp5.flops_forward_dense() / us5 / 1e6}
                if prec == "f32":
                    # the sweep every MCMC step of this config runs (flow.py:116-132; float32 whatever the flow's precision,
                    # csrc/maf_inverse_tri6.hip): 5000 walkers are two rounds of 256 workgroups x 16 walkers, 4096 are one
                    z5 = torch.randn(5000, 128, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
                    inv = {}
                    for rows in (4096, 5000):
                        for _ in range(2):
                            f5.inverse(z5[:rows])
                        e0.record()
                        for _ in range(5):
                            f5.inverse(z5[:rows])
                        e1.record()
                        torch.cuda.synchronize()
                        us_inv = e0.elapsed_time(e1) / 5 * 1e3
                        tf_inv = rows * 2 * sp5.macs_masked() / us_inv / 1e6
                        inv[str(rows)] = {"us_per_call": us_inv, "rows_per_s": rows / us_inv * 1e6, "achieved_tflops": tf_inv,
                                          "peak_tflops": PEAK_F32_MFMA_TFLOPS, "frac": tf_inv / PEAK_F32_MFMA_TFLOPS}
                    flow_cfg5["inverse_f32"] = inv
                # Flow.fit's optimizer step at the reference's batch size (sampler.py:289: 512 rows): loss + gradient,
                # clip, AdamW, image refresh -- the float32 chain + weight-gradient kernels against the bf16 per-layer products
                from pocomc_amd.train import AdamW
                opt5 = AdamW(f5, 1e-3)
                acc5 = torch.zeros(1, dtype=torch.float32, device="cuda")
                opt5.epoch(x5, None, None, 512, 1.0, acc5)
                e0.record()
                for _ in range(3):
                    opt5.epoch(x5, None, None, 512, 1.0, acc5)
                e1.record()
                torch.cuda.synchronize()
                us_step = e0.elapsed_time(e1) / 30 * 1e3
                dense = 3 * sp5.flops_forward_dense() * 5000 / 10 / us_step / 1e6
                executed = dense if prec == "bf16" else 3 * 2 * sp5.macs_masked() * 5000 / 10 / us_step / 1e6
                flow_cfg5[prec]["fit"] = {"us_per_step_of_512_rows": us_step, "rows_per_s": 5000 / 10 / us_step * 1e6,
                                          "dense_equivalent_tflops": dense, "executed_tflops": executed,
                                          "frac": executed / peak,
                                          "engine": "csrc/maf_train_bf16.hip (dense products, masked weights are zeros)"
                                          if prec == "bf16" else "csrc/maf_train.hip (masked tiles skipped)"}
                del f5, opt5
            flow_cfg5["speedup_bf16"] = flow_cfg5["f32"]["us_per_call"] / flow_cfg5["bf16"]["us_per_call"]
            flow_cfg5["fit_speedup_bf16"] = (flow_cfg5["f32"]["fit"]["us_per_step_of_512_rows"]
                                             / flow_cfg5["bf16"]["fit"]["us_per_step_of_512_rows"])
        except Exception as exc:                                          # (a sub-metric must not take the bench down)
            flow_cfg5 = {"error": repr(exc)}
    ms_per_step = dt / args.steps * 1e3
    value = (n * world * args.steps / dt) / 1e4
    # 16-bit helper products: how far the sweep this run timed is from the float32 sweep ON THE TRAINED FLOW, at the current
    # walkers' theta (the opt-in precision's cost in accuracy, measured where it is used)
    lane16_check = None
    if args.precision != "f32" and bool(flow._desc.lane16):
        th = (leng.lanes[0].theta32 if leng is not None else eng.theta32)[:4096].clone()
        keep = flow.inverse_algo
        flow.inverse_algo = 0
        x16, l16 = flow.inverse(th)
        flow.inverse_algo = 8                               # PMC_INVERSE_TRIANGULAR_LANE: the float32 helpers
        x32, l32 = flow.inverse(th)
        flow.inverse_algo = keep
        okr = torch.isfinite(x32).all(dim=1) & torch.isfinite(x16).all(dim=1)
        ex = ((x16 - x32).abs().max(dim=1).values / x32.abs().max(dim=1).values.clamp_min(1e-30))[okr]
        el = (l16 - l32).abs()[okr]
        lane16_check = {"rows": int(okr.sum().item()), "x_rel_err_max": float(ex.max().item()), "x_rel_err_median": float(ex.median().item()),
                        "ladj_abs_err_max": float(el.max().item()), "ladj_abs_err_median": float(el.median().item()),
                        "note": f"{args.precision} left-looking products against the float32 sweep of the same (trained) flow, per walker"}
    out = {"metric": "preconditioned MCMC steps/sec (1e4 particles, 32-D)", "value": value,
           "unit": ("steps/s per 1e4 walkers" if world == 1 else "steps/s × global_walkers/1e4 (weak)"), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": ("f32 flow (MFMA) + f64 step" if flow.inverse_precision_active == "f32" else
                     f"{flow.inverse_precision_active} left-looking products + f32 chain of the flow inverse (MFMA, f32 accumulation) + f64 step"),
           "data": "synthetic",
           "config": {"workload": f"{D}-D {TARGET_NAMES[args.target]}, U({p_lo:g},{p_hi:g})^{D} prior, {n} walkers/GPU x {world} GPU = {n * world} walkers, {args.flow} "
                                  f"(H={spec.hidden}), beta={beta}, tpCN kernel, host numpy likelihood in the loop",
                      "walkers_per_gpu": n, "global_walkers": n * world, "n_dim": D, "flow": args.flow,
                      "flow_trained_50_epochs": flow_trained, "flow_fit_epochs": (flow_fit or {}).get("epochs"),
                      "flow_fit_rule": ("50 epochs" if D <= 64 else "the Sampler's: patience = D, best validation state restored at the early stop"), "parallelism": f"walker-sharded x{world}",
                      "lanes": len(leng.lanes) if leng is not None else 1,
                      "lane_rows": [int(e_.n) for e_ in leng.lanes] if leng is not None else [n],
                      "pipelined_device_adaptation": bool(pipelined and leng is not None),
                      "timed_region": ("closed: K steps between two barriers with every launch of the K steps enqueued behind t0 -- the "
                                       "pre-steps (proposal + flow inverse + scaler/prior epilogue + hand-over of x') of step 1 are "
                                       "issued right after t0 and run exposed, steps 1..K-1 enqueue their successor's, the K-th "
                                       "enqueues none: K proposal+sweep launches, K likelihoods, K accepts (one preconditioned_pcn "
                                       "call of K steps, pocomc_amd/mcmc.py::_run)"),
                      "inverse_algo": args.inverse, "inverse_precision": flow.inverse_precision_active,
                      "inverse_precision_requested": args.precision, "inverse_guard": flow.inverse_guard, "host_threads": args.host_threads, "host_prefetch_threads": args.host_prefetch, "host_x_order": args.x_order, "prior_on_device": bool(device_prior),
                      "accept_rate": float((ad_l if leng is not None else ad).mean_alpha),
                      "backend": (dist.get_backend() if world > 1 else None),
                      "collectives": (None if world == 1 else
                                      ("pmc_comm mailboxes inside the C pipeline (" + comm_kind_name() + ", one D+4-double sum per step; "
                                       "the process group only exchanges the handles at start-up)"
                                       if (laned_host or {}).get("pipeline") else
                                       ("RCCL (torch.distributed nccl backend on ROCm)" if dist.get_backend() == "nccl"
                                        else "gloo (functional check on a shared GPU)"))),
                      "ranks_reported_by_backend": (dist.get_world_size() if world > 1 else 1),
                      "shared_gpu": bool(os.environ.get("PMC_BENCH_SHARE_GPU"))},
           "roofline": roofline,
           "roofline_sweeps": sweeps,
           "flow_fit": flow_fit,
           "inverse_16bit_vs_f32": lane16_check,
           "flow_config5": flow_cfg5,
           "device_only_steps_per_s": 1e6 / (us["propose"] + us["maf_inverse"] + us["scaler_inverse"]
                                             + us["accept_reduce"]),
           "breakdown_us_per_step": dict(us, host_prior_likelihood=t_host[0] / n_inst * 1e6,
                                         device_kernels=us["propose"] + us["maf_inverse"] + us["scaler_inverse"]
                                         + us["accept_reduce"], wall=ms_per_step * 1e3,
                                         instrumented_pass_wall=dt_inst / n_inst * 1e6),
           "timed_region_host_us_per_step": seg_timed,
           "laned_path_host_us_per_step": laned_host,
           "composite_path_host_us_per_step": composite_host,
           "host_us_per_step": {**{k: v / n_inst * 1e6 for k, v in eng.host_timers.items()},
                                **{k: v / n_inst * 1e6 for k, v in t_seg.items()}}}
    out["config"]["driver_pinned_to_core"] = pinned_core
    out["config"]["pin_calibration"] = pin_probe
    if per_rank is not None:
        out["per_rank_us_per_step"] = per_rank
    if dt_ss is not None:
        out["steady_state"] = {"steps": n_ss, "value": (n * world * n_ss / dt_ss) / 1e4, "ms_per_step": dt_ss / n_ss * 1e3,
                               "note": "the same closed region over more steps (the exposed first pre-step is 1/K of it)"}
    if lane16_check is not None and not (lane16_check["x_rel_err_max"] <= LANE16_BOUND):
        # a 16-bit sweep that is not an inverse of the trained flow within the stated bound is not a measurement of this metric
        out["value_refused"] = {"value_measured": value, "reason": f"16-bit sweep: max per-walker relative error on x "
                                f"{lane16_check['x_rel_err_max']:.3g} > {LANE16_BOUND:g} against the float32 sweep of the trained flow"}
        out["value"] = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(D, n, beta, flow.params.cpu().numpy(), spec, x, u, geo, sigma0, seed=0,
                                           target=target, bounds=(p_lo, p_hi))
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
