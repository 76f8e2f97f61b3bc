# same-box A/B of the blend-forward variants: parity tests on the wave kernel, then bench legs (one-stream kernel times from --breakdown)
mkdir -p gpurun_out/r03j
AG_FWD_KERNEL=1 timeout 600 python -m pytest tests/test_raster_gpu.py -q 2>&1 | tail -4
run() { # label env...
  label=$1; shift
  env "$@" timeout 200 python bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-full-step --no-stress --breakdown > gpurun_out/r03j/bench_$label.json 2> gpurun_out/r03j/bench_$label.err
  python - "$label" <<'PY'
import json, sys
d=json.loads(open(f"gpurun_out/r03j/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
k=d["kernels_us"]
print(f"{sys.argv[1]:10s} value {d['value']:8.1f}  seq {d['sequential']['views_per_s']:7.1f}  fwd {k.get('blend_forward_kernel')}  bwd {k.get('blend_backward_kernel')}  pre {k.get('preprocess_kernel')} scan {k.get('tile_scan_kernel')} scatter {k.get('scatter_kernel')} sort {k.get('tile_sort_kernel')} prebwd {k.get('preprocess_backward_kernel')}")
PY
}
run region AG_FWD_KERNEL=0
run wave AG_FWD_KERNEL=1
run region2 AG_FWD_KERNEL=0
run wave2 AG_FWD_KERNEL=1
