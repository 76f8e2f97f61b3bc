# same-box A/B of the fused scan (scan as block 0 of the scatter launch) against the separate tile_scan_kernel
mkdir -p gpurun_out/r03k
timeout 600 python -m pytest tests/test_raster_gpu.py -q 2>&1 | tail -3
run() { # label env...
  label=$1; shift
  env "$@" timeout 200 python bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-full-step --no-stress --breakdown > gpurun_out/r03k/bench_$label.json 2> gpurun_out/r03k/bench_$label.err
  python - "$label" <<'PY'
import json, sys
d=json.loads(open(f"gpurun_out/r03k/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
k=d["kernels_us"]
print(f"{sys.argv[1]:10s} value {d['value']:8.1f}  seq {d['sequential']['views_per_s']:7.1f}  op {d['operator_path']['views_per_s']:7.1f}  pre {k.get('preprocess_kernel')} scan {k.get('tile_scan_kernel')} scatter {k.get('scatter_kernel')} sort {k.get('tile_sort_kernel')} fwd {k.get('blend_forward_kernel')} bwd {k.get('blend_backward_kernel')}")
PY
}
run separate AG_FUSED_SCAN=0
run fused AG_FUSED_SCAN=1
run separate2 AG_FUSED_SCAN=0
run fused2 AG_FUSED_SCAN=1
