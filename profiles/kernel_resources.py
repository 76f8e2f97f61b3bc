"""Register / LDS / spill table of the kernels of one translation unit (compiles it with -Rpass-analysis=kernel-resource-usage).
    python profiles/kernel_resources.py animatablegaussians_amd/csrc/ag_conv.hip [name filter] [extra hipcc flags...]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
       "-ffp-contract=fast", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_kr.o"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    if flt and flt not in r["name"]:
        continue
    print(f"VGPR {r.get('VGPRs', -1):4d} AGPR {r.get('AGPRs', -1):3d} spill {r.get('VGPRs Spill', -1):3d} scratch {r.get('ScratchSize [bytes/lane]', -1):4d} "
          f"occ {r.get('Occupancy [waves/SIMD]', -1)} LDS {r.get('LDS Size [bytes/block]', -1):6d}  {r['name'][:150]}")
