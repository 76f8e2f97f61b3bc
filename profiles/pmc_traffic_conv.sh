#!/bin/bash
# HBM traffic of the fp16 gather convolution on one layer: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), per-launch averages.
#   bash profiles/pmc_traffic_conv.sh Cin Cout H W
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcc_$c
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmcc_$c -o p --output-format csv -- python $R/profiles/conv_one.py "$@" 3 1 1 10 fwd > /dev/null 2>&1
  python - $c /tmp/pmcc_$c <<'PY'
import csv, glob, collections, sys
c, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "ag::" in k:
        print(f"{c:11s} {k:62s} {sum(v)/len(v):14.1f} KB raw per launch ({len(v)} launches)")
PY
done
