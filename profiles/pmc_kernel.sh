#!/bin/bash
# usage: profiles/pmc_kernel.sh <kernel-name-fragment> <python script + args...>   -- prints per-launch averages of SQ counters
frag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pmck_$i -o p --output-format csv -- "$@" > /dev/null 2>&1
  python - "$frag" /tmp/pmck_$i <<'PY'
import csv, glob, collections, sys
frag, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no output in", d); raise SystemExit
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if frag in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:32s} {sum(v)/len(v):16.1f}  ({len(v)} launches)")
PY
done
