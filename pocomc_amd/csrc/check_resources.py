"""Build-time check of the lane-per-walker sweep (csrc/maf_inverse_tri6.hip): its register-capped instances -- plain
inverse (FM = 0) with one subset, or two subsets and 16-bit helpers: __launch_bounds__(64 NW, 2) -- exist because
they run without AGPR copies and without scratch; a toolchain that spills them would be a silent regression of the hot
kernel of BASELINE configs 3 and 5.  Reads the compiler's -Rpass-analysis=kernel-resource-usage remarks and fails the
build if a capped instance uses scratch or accumulation registers.   python3 check_resources.py <remarks.txt>"""
import re
import sys

txt = open(sys.argv[1]).read()
name = re.compile(r"Function Name: (\S+)")
inst = re.compile(r"maf_inverse_tri6_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E")
blocks = name.split(txt)[1:]
bad, seen = [], 0
for fn, body in zip(blocks[0::2], blocks[1::2]):
    m = inst.search(fn)
    if not m:
        continue
    ns, fm, nw, hb = (int(v) for v in m.groups())
    capped = fm == 0 and (ns == 1 or (ns == 2 and hb != 0))
    if not capped:
        if fm == 0 and nw == 5:        # five wavefronts share four SIMDs: 256 registers by construction, reported only
            sc = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", body).group(1))
            if sc:
                print(f"check_resources: note: maf_inverse_tri6_kernel<{ns}, {fm}, {nw}, {hb}> uses {sc} B/lane of scratch")
        continue
    seen += 1
    scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", body).group(1))
    agpr = int(re.search(r"AGPRs: (\d+)", body).group(1))
    vgpr = int(re.search(r"\bVGPRs: (\d+)", body).group(1))
    if scratch or agpr or vgpr > 256:
        bad.append(f"maf_inverse_tri6_kernel<{ns}, {fm}, {nw}, {hb}>: VGPRs {vgpr}, AGPRs {agpr}, scratch {scratch} B/lane")
if not seen:
    sys.exit("check_resources: no capped instance of maf_inverse_tri6_kernel found in the remarks")
if bad:
    sys.exit("check_resources: register-capped instances spill:\n  " + "\n  ".join(bad))
print(f"check_resources: {seen} capped instances of maf_inverse_tri6_kernel: no scratch, no AGPRs")
