set -x
mkdir -p gpurun_out/r06a
timeout 900 python -m pytest tests/test_raster_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06a/raster_tests.txt
timeout 120 python profiles/bwd_px_stats.py 0 2 5 > gpurun_out/r06a/px_stats.txt 2>&1
for k in pixel wave pixel wave; do
  AG_BWD_KERNEL=$k timeout 200 python bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-full-step --no-stress > gpurun_out/r06a/bench_$k.json 2> gpurun_out/r06a/bench_$k.err
  python - $k <<'PY'
import json, sys
d=json.loads(open(f"gpurun_out/r06a/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:8s} value {d['value']:8.1f}  seq {d['sequential']['views_per_s']:7.1f}  bwd us (1 stream) {d['sequential']['blend_backward_avg_launch_us']:6.1f}  overlapped {d['roofline']['avg_launch_us']:6.1f}")
PY
done > gpurun_out/r06a/ab.txt 2>&1
cat gpurun_out/r06a/raster_tests.txt gpurun_out/r06a/px_stats.txt gpurun_out/r06a/ab.txt
