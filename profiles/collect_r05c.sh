#!/bin/bash
# Round-5 (third session) artefacts for profiles/ from ONE box:  bash profiles/collect_r05c.sh   (run through gpurun; outputs under gpurun_out/r05z)
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r05z
mkdir -p $O
# 1. the driver's default command, and the driver's round-end command (--steps 20 --warmup 5)
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -3 $O/bench.time
python bench.py --steps 20 --warmup 5 --no-full-step --no-cpu-baseline --no-stress > $O/bench_driver_cmd.json 2> /dev/null
# 2. rocprofv3 kernel summary of the raster legs of the same command
rm -rf /tmp/prof_bench
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o b -- python $R/bench.py --no-full-step --no-cpu-baseline --no-stress > $O/bench_profiled.json 2> /dev/null )
db=$(find /tmp/prof_bench -name "*.db" | head -1)
python profiles/summarize_rocprof.py "$db" $O/bench_kernel_stats.csv | head -10 | cut -c1-150
# 3. HBM traffic per launch: two PMC passes (one counter each), calibrated on a 256-MiB copy of the same run; raster + gather / LBS kernels
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  ( cd /tmp && rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o p --output-format csv -- python $R/profiles/traffic_probe.py > /dev/null 2>&1 )
done
python profiles/traffic_summarize.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/traffic.json > /dev/null 2>&1; python - <<PY
import json
t=json.load(open("$O/traffic.json"))
print({k.split("::")[-1][:28]: v["hbm_bytes"] for k,v in t["kernels"].items()})
PY
# 4. whole training steps (1 view per step, the grouped chain) under rocprofv3
rm -rf /tmp/prof_fs
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_fs -o fs -- python $R/profiles/fullstep_prof.py 8 > /dev/null 2>&1 )
db=$(find /tmp/prof_fs -name "*.db" | head -1)
python profiles/summarize_rocprof.py "$db" $O/fullstep_kernel_stats.csv | head -8 | cut -c1-150
# 5. steps per view count; per-view breakdown; glue sources; the full GPU suite
python profiles/views_scaling.py 1 2 4 8 16 2>&1 | grep -v amdgpu.ids | tee $O/views_scaling.txt
python profiles/per_view_breakdown.py 1 3 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > $O/per_view_breakdown.txt; head -3 $O/per_view_breakdown.txt
python profiles/step_glue_sources.py 1 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > $O/glue_sources.txt; head -3 $O/glue_sources.txt
( time python -m pytest tests -m gpu -q 2>&1 | tail -4 ) > $O/gputests_head.txt 2>&1; cat $O/gputests_head.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
