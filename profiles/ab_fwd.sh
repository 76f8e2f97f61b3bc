# same-box A/B of blend-forward build variants (profiles/ub/build_variant.sh <name> ag_blend_forward ...): parity on HEAD, bench legs per variant
mkdir -p gpurun_out/r03s
timeout 600 python -m pytest tests/test_raster_gpu.py -q 2>&1 | tail -3
run() { # label env...
  label=$1; shift
  env "$@" timeout 200 python bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-full-step --no-stress --breakdown > gpurun_out/r03s/bench_$label.json 2> gpurun_out/r03s/bench_$label.err
  python - "$label" <<'PY'
import json, sys
d=json.loads(open(f"gpurun_out/r03s/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
k=d["kernels_us"]
print(f"{sys.argv[1]:10s} value {d['value']:8.1f}  seq {d['sequential']['views_per_s']:7.1f}  fwd {k.get('blend_forward_kernel')}  bwd {k.get('blend_backward_kernel')}")
PY
}
run head X=1
for v in "$@"; do run $v AG_LIB_PATH=$PWD/profiles/ub/ko/libag_$v.so; done
run head2 X=1
