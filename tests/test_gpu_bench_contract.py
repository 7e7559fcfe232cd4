"""bench.py prints ONE JSON line with the fields the driver's contract names (metric / value / unit / n_gpus / steps /
warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config + roofline + cpu_baseline)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [[], ["--lanes", "1", "--no-pipeline"], ["--flow", "nsf3"]])
def test_bench_line_follows_the_contract(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--particles", "1024", "--steps", "6", "--warmup", "2"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["ms_per_step"] - 1024 / 1e4 / d["value"] * 1e3) < 1e-6 * d["ms_per_step"] + 1e-9
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 * max(1.0, r["frac"])
    assert r["traffic"] is None or r["traffic"] > 0          # (PMC bytes only for the profiled headline launch size)
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1


def test_two_ranks_on_a_shared_gpu_report_a_two_rank_line():
    """``python bench.py --gpus 2`` on a one-GPU box: the bench launches its own two ranks (``torch.distributed.run``;
    with fewer GPUs than ranks they share device 0 over gloo -- the multi-rank code path, flagged ``shared_gpu``): rc 0,
    one JSON line from rank 0, ``n_gpus == 2``, 2 x 10000 walkers, a backend that reports two ranks, no CPU baseline at
    N > 1, roofline present -- so that the first run on a real 8-GPU node cannot die on plumbing."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-flow-bench"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak"
    c = d["config"]
    assert c["global_walkers"] == 20000 and c["walkers_per_gpu"] == 10000
    assert c["ranks_reported_by_backend"] == 2 and c["backend"] in ("nccl", "gloo") and c["collectives"]
    import torch
    assert c["shared_gpu"] == (torch.cuda.device_count() < 2)
    assert d["value"] > 0 and abs(d["ms_per_step"] - 20000 / 1e4 / d["value"] * 1e3) < 1e-6 * d["ms_per_step"] + 1e-9
    assert d["cpu_baseline"] is None and d["roofline"]["frac"] > 0
    # the sharded step runs behind the C ABI (pmc_pipeline_next + the pmc_comm mailboxes): no process-group call per step
    assert "global_walkers/1e4 (weak)" in d["unit"]
    h = d["laned_path_host_us_per_step"]
    assert h["pipeline"] == "pmc_pipeline_next (C ABI)" and "pmc_comm" in c["collectives"]
    print("two-rank python_overhead us/step:", h["python_overhead"])
    assert h["python_overhead"] <= 12.0, h               # (measured 9.3 on a shared GPU, 20 timed steps; 2 us of margin)


def test_eight_ranks_on_a_shared_gpu_report_an_eight_rank_line():
    """``python bench.py --gpus 8`` on a one-GPU box (BASELINE configs[3]: 80 000 walkers sharded 8 x): eight ranks share
    device 0 over gloo -- a rehearsal of the 8-way plumbing (handle exchange, mailbox indexing, per-rank core choice, the
    data-parallel fit at 64 local rows per batch), not a measurement: rc 0, one line, ``n_gpus == 8``, eight ranks reported
    by the backend, a workload that names 80 000 walkers, eight per-rank clock records on eight distinct cores."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "6", "--warmup", "2", "--no-flow-bench",
           "--no-steady-state"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 6 and d["scaling"] == "weak"
    c = d["config"]
    assert c["global_walkers"] == 80000 and c["walkers_per_gpu"] == 10000 and "80000 walkers" in c["workload"]
    assert c["ranks_reported_by_backend"] == 8 and "pmc_comm" in c["collectives"]
    assert d["value"] > 0 and abs(d["ms_per_step"] - 80000 / 1e4 / d["value"] * 1e3) < 1e-6 * d["ms_per_step"] + 1e-9
    pr = d["per_rank_us_per_step"]
    assert len(pr) == 8 and sorted(r["rank"] for r in pr) == list(range(8))
    cores = [r["pinned_core"] for r in pr]
    assert None not in cores and len(set(cores)) == 8, cores
    assert all(r["wait_x"] is not None and r["likelihood"] > 0 for r in pr)
    assert d["cpu_baseline"] is None and d["roofline"]["frac"] > 0
