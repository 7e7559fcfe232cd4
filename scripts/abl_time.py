"""Time the two-wave sweep of every library under scripts/abl/ (timing-only builds, scripts/abl_tri5.sh) next to the product
library: one subprocess per library (PMC_LIBRARY).   python scripts/abl_time.py [n] [D] [flow]"""
import glob, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n, D, flow = (sys.argv[1:] + ["7008", "32", "maf3"][len(sys.argv) - 1:])[:3]
code = f"""
import sys, torch
sys.path.insert(0, {root!r})
import pocomc_amd as pc
f = pc.Flow({D}, {flow!r}, seed=0)
f.inverse_algo = int(__import__("os").environ.get("ABL_ALGO", "7"))
z = torch.randn({n}, {D}, device='cuda')
for _ in range(5): f.inverse(z)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): f.inverse(z)
e1.record(); torch.cuda.synchronize()
print(round(e0.elapsed_time(e1) / 50 * 1e3, 2))
"""
libs = [None] + sorted(glob.glob(os.path.join(root, "scripts", "abl", "lib_*.so")))
for lib in libs:
    env = dict(os.environ)
    if lib:
        env["PMC_LIBRARY"] = lib
        env["PMC_ALLOW_ABLATION"] = "1"                      # (timing-only builds: pocomc_amd._lib refuses them otherwise)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(f"{os.path.basename(lib) if lib else 'product':24s} {out.stdout.strip() or out.stderr.strip()[-200:]} us")
