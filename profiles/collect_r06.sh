#!/bin/bash
# Round-6 artefacts for profiles/ from ONE box:  bash profiles/collect_r06.sh   (run through gpurun; outputs under gpurun_out/r06z)
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r06z
mkdir -p $O
# 1. the driver's round-end command with every leg (its last 2000 characters = what the driver's record keeps), and the default command
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
tail -c 2000 $O/bench_driver_cmd.json > $O/bench_driver_cmd_tail2000.txt
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -3 $O/bench.time
# 2. rocprofv3 kernel summary of the raster legs of the same command
rm -rf /tmp/prof_bench
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o b -- python $R/bench.py --no-full-step --no-cpu-baseline --no-stress > $O/bench_profiled.json 2> /dev/null )
db=$(find /tmp/prof_bench -name "*.db" | head -1)
python profiles/summarize_rocprof.py "$db" $O/bench_kernel_stats.csv | head -10 | cut -c1-150
# 3. HBM traffic per launch: two PMC passes (one counter each), calibrated on a 256-MiB copy of the same run
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  ( cd /tmp && rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o p --output-format csv -- python $R/profiles/traffic_probe.py > /dev/null 2>&1 )
done
python profiles/traffic_summarize.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/traffic.json > /dev/null 2>&1; python - <<PY
import json
t=json.load(open("$O/traffic.json"))
print({k.split("::")[-1][:28]: v["hbm_bytes"] for k,v in t["kernels"].items()})
PY
# 4. SQ counters of the headline kernel (VALU busy for the `valu` roofline's cross-check)
bash profiles/pmc_kernel.sh blend_backward_wave python $R/profiles/one_view.py 0 > $O/pmc_blend_backward.txt 2>&1; grep -E "SQ_INSTS_VALU |SQ_ACTIVE_INST_VALU|GRBM_GUI_ACTIVE|SQ_WAVE_CYCLES" $O/pmc_blend_backward.txt
bash profiles/pmc_kernel.sh blend_forward python $R/profiles/one_view.py 0 > $O/pmc_blend_forward.txt 2>&1
# 5. whole training steps (1 view per step, the grouped chain) under rocprofv3
rm -rf /tmp/prof_fs
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_fs -o fs -- python $R/profiles/fullstep_prof.py 8 > /dev/null 2>&1 )
db=$(find /tmp/prof_fs -name "*.db" | head -1)
python profiles/summarize_rocprof.py "$db" $O/fullstep_kernel_stats.csv | head -8 | cut -c1-150
# 6. steps per view count; the full GPU suite with the parity reports; smoke
python profiles/views_scaling.py 1 2 4 8 16 2>&1 | grep -v amdgpu.ids | tee $O/views_scaling.txt
mkdir -p $O/reports
( time AG_TEST_REPORT_DIR=$O/reports python -m pytest tests -m gpu -q -s 2>&1 | grep -E "parity\]|passed|failed|error" ) > $O/gputests_head.txt 2>&1; tail -4 $O/gputests_head.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
