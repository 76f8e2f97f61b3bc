# per-kernel time of a training step with the DMA loader on / off (rocprofv3 --kernel-trace --stats)
out=$PWD/gpurun_out/$1; mkdir -p $out
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  AG_CONV_DMA=$v rocprofv3 --kernel-trace --stats -d /tmp/prof_dma$v -o p --output-format csv -- python $R/bench_avatar.py --steps 6 --warmup 2 > /dev/null 2>&1
  f=$(find /tmp/prof_dma$v -name "*kernel_stats.csv" | head -1)
  cp $f $out/kernel_stats_dma$v.csv
  python - $f $v <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"AG_CONV_DMA={sys.argv[2]} total kernel time {tot/1e6:.2f} ms")
for r in rows[:14]:
    print(f'  {r["Name"][:90]:90s} calls {int(r["Calls"]):6d} total {float(r["TotalDurationNs"])/1e6:9.3f} ms avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
done 2>&1 | tee $out/summary.txt
