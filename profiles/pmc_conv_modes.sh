#!/bin/bash
# SQ / memory-path counters of the gather convolution 256 -> 256 @256^2 (77 GFLOP) in the three-product and the one-product fp16 forms:
# what does a K step wait for once the matrix work is a third?     bash profiles/pmc_conv_modes.sh <out dir>
OUT="${1:-gpurun_out/pmc_conv_modes}"
mkdir -p "$OUT"
R=${GRAFT_REPO_ROOT:-$PWD}
for mode in split_f16 f16; do
  AG_CONV_MATH=$mode bash "$R/profiles/pmc_kernel.sh" gather_conv_split python "$R/profiles/conv_one.py" 256 256 256 256 3 1 1 10 fwd > "$OUT/sq_$mode.txt" 2>&1
  ( cd /tmp && export TMPDIR=/tmp
    for set in "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
      rm -rf /tmp/pmcm; AG_CONV_MATH=$mode rocprofv3 --pmc $set --kernel-trace -d /tmp/pmcm -o p --output-format csv -- python "$R/profiles/conv_one.py" 256 256 256 256 3 1 1 10 fwd > /dev/null 2>&1
      python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pmcm/**/*counter_collection.csv", recursive=True)
if f:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "gather_conv_split" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:36s} {sum(v)/len(v):16.1f}  ({len(v)} launches)")
else:
    print("no output")
PY
    done ) > "$OUT/mem_$mode.txt" 2>&1
done
rocprofv3 -L 2>/dev/null | grep -o "\bT[AC][A-Z_]*[A-Za-z0-9_]*" | sort -u | head -150 > "$OUT/counter_names.txt"
