"""Summary of the rocprofv3 passes of scripts/collect_profile.sh: per-kernel durations from the kernel trace and the
mean PMC counters per launch; writes <dir>/../<tag>_traffic.json with the HBM bytes per launch of the step's
dominant kernel (FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 counts 64-byte fetches in FETCH_SIZE at half weight,
hence 2 x FETCH_SIZE -- /opt/skills/guides/MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash                               # noqa: E402

out, tag, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
short = lambda n: n.split("(")[0][:64] if not n.startswith("void ") else n[5:].split("(")[0][:64]


def one(pattern):
    f = glob.glob(os.path.join(out, pattern), recursive=True)
    return f[0] if f else None


print(f"# {tag}: rocprofv3 on MI355X of `{cmd}`")
print("# pass 1: --kernel-trace --stats ; passes 2-4: --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_* (separate runs)")
ks = one("kt/**/*kernel_stats.csv")
if ks:
    print("\n## kernel-trace stats")
    print(f"{'kernel':64s} {'calls':>7s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>7s}")
    for r in csv.DictReader(open(ks)):
        print(f"{short(r['Name']):64s} {int(r['Calls']):7d} {float(r['AverageNs']) / 1e3:10.2f} "
              f"{float(r['MinNs']) / 1e3:10.2f} {float(r['MaxNs']) / 1e3:10.2f} {float(r['Percentage']):7.2f}")
kt = one("kt/**/*kernel_trace.csv")
if kt:
    # the step kernels of the timed region by launch size (lanes launch the same kernel on fewer rows)
    by = defaultdict(list)
    for r in csv.DictReader(open(kt)):
        n = short(r["Kernel_Name"])
        if any(w in n for w in ("tri4", "tri5", "tri6", "tri_nsf", "scaler_inverse", "accept_kernel", "rng_fill", "adapt_update")):
            by[(n, r.get("Grid_Size") or r.get("Grid_Size_X", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("\n## step kernels by launch size (grid = work-items)")
    print(f"{'kernel':64s} {'grid':>9s} {'calls':>7s} {'avg_us':>10s}")
    for (n, g), v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print(f"{n:64s} {g:>9s} {len(v):7d} {sum(v) / len(v):10.2f}")
if kt:
    # the trainer's launches in launch order: the fit is the first GPU work of the bench process (50 epochs, ~50 ms), so
    # its first launches run while the device's clocks ramp up from idle -- the mean of a kernel over the whole fit is not
    # its steady duration; median and the means by fifth of the launch order say which is which
    rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r["Start_Timestamp"]))
    tr = defaultdict(list)
    for r in rows:
        n = short(r["Kernel_Name"])
        if any(w in n for w in ("maf_chain", "maf_dw", "adamw_kernel")):
            tr[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if tr:
        print("\n## trainer kernels in launch order (us): median, 10th / 90th percentile, mean of each fifth of the launches")
        print("# (bench.py fits TWO flows with the float32 trainer: the step benchmark's own -- maf3 at D = 32 by default, 50 epochs,")
        print("#  the first ~90 % of the launches -- and, unless --no-flow-bench, the config-5 flow of its flow_config5 sub-metric")
        print("#  (D = 128, 8 transforms, H = 512: the same kernel template and grid, ~6 x the time): the kernel-trace MEAN above mixes")
        print("#  the two; the median is the step benchmark's flow)")
        for n, v in tr.items():
            v_ = sorted(v)
            q = lambda f: v_[min(len(v_) - 1, int(f * len(v_)))]
            fifths = [sum(v[i * len(v) // 5:(i + 1) * len(v) // 5]) / max(1, (i + 1) * len(v) // 5 - i * len(v) // 5) for i in range(5)]
            print(f"{n:48s} n={len(v):5d} median {q(0.5):8.2f}  p10 {q(0.1):8.2f}  p90 {q(0.9):8.2f}  fifths " + " ".join(f"{f:8.2f}" for f in fifths))
pmc = defaultdict(lambda: defaultdict(list))
grids = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    f = one(f"{sub}/**/*counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        key = (short(r["Kernel_Name"]), r.get("Grid_Size", "?"))
        pmc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("\n## PMC counters, mean per launch (kernel, grid)")
want = ("tri4", "tri5", "tri6", "tri_nsf", "scaler_inverse", "accept_kernel", "maf_chain", "maf_dw", "adamw", "forward_wg", "wide_phase", "forward_bf16")
traffic = None
for key in sorted(pmc, key=lambda k: -len(pmc[k].get("SQ_WAVE_CYCLES", []))):
    if not any(w in key[0] for w in want):
        continue
    print(f"{key[0]}  grid={key[1]}")
    m = {c: sum(v) / len(v) for c, v in pmc[key].items()}
    for c in sorted(m):
        print(f"    {c:32s} {m[c]:16.1f}   ({len(pmc[key][c])} launches)")
    if any(w in key[0] for w in ("tri6", "tri5", "tri4")) and "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        try:
            g_items = int(key[1])
        except ValueError:
            g_items = 0
        # walkers of the launch from its grid (work-items): 16 walkers per 64-lane wave (tri4), per 128 threads (tri5)
        per16 = 128 if "tri5" in key[0] else 64
        walkers = g_items // per16 * 16 if ("tri5" in key[0] or "tri4" in key[0]) else None
        D_, io = 32, None
        if walkers:
            # fused proposal + inverse at D = 32 (bench default): theta (f32) in; theta' (f64), u' (f32), ladj, two
            # quadratic forms out; the weight sections the register-chain sweeps read, once
            io = {"walker_io": walkers * (4 * D_ + 8 * D_ + 4 * D_ + 4 + 16), "weights_once": 3 * 60400 * 4}
            if ", 8>" in key[0] or ", 16>" in key[0] or ", 4>" in key[0]:
                # the fused launch of the step also applies the scaler + prior (epilogue): u', x' (f64) to device memory,
                # x' column-major + finite + logp' to pinned host memory, logdetj / finite / logp' on the device
                io["scaler_epilogue_out"] = walkers * (3 * 8 * D_ + 8 + 4 + 8 + 4 + 8)
            io["total"] = sum(io.values())
        cand = {"kernel": key[0], "grid": key[1], "launches": len(pmc[key]["FETCH_SIZE"]),
                "walkers_per_launch": walkers, "algorithmic_bytes": io,
                "note": "FETCH_SIZE / WRITE_SIZE are per-launch means of separate --pmc passes; the weight image is pulled "
                        "once per XCD L2 (8 x), walker data is touched once; WRITE_SIZE includes the epilogue's stores to pinned host memory",
                "FETCH_SIZE_KB": m["FETCH_SIZE"], "WRITE_SIZE_KB": m["WRITE_SIZE"],
                "hbm_bytes_per_launch": (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0,
                "correction": "2x FETCH_SIZE (gfx950), separate --pmc passes",
                "source": f"profiles/{tag}_summary.txt", "kernel_source_hash": kernel_source_hash()}
        if traffic is None or cand["launches"] > traffic["launches"]:
            traffic = cand
if traffic:
    json.dump(traffic, open(os.path.join(os.path.dirname(out.rstrip('/')), f"{tag}_traffic.json"), "w"), indent=1)
