#!/usr/bin/env python3
"""Developer probe: one line per kernel of csrc/bsm_kernels.hip with its registers, scratch, occupancy and LDS
(hipcc -Rpass-analysis=kernel-resource-usage).  usage: kres.py [substring ...] [-D...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "blocksparsematrices.jl_amd", "csrc")
defs = [a for a in sys.argv[1:] if a.startswith("-D")]
pats = [a for a in sys.argv[1:] if not a.startswith("-D")]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics",
       "-Rpass-analysis=kernel-resource-usage", "-c", "bsm_kernels.hip", "-o", "/dev/null"] + defs
out = subprocess.run(cmd, cwd=src, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: (.*?): (.*?) \[-Rpass", line)
    if not m:
        if "error" in line: print(line)
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    n = r["name"].replace("bsm::", "").replace("void ", "")
    if pats and not any(p in n for p in pats): continue
    print(f"{n:70s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>3s} sgpr {r.get('TotalSGPRs','?'):>4s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} occ {r.get('Occupancy [waves/SIMD]','?'):>2s} lds {r.get('LDS Size [bytes/block]','?'):>6s}")
