# same-box A/B of blend-backward build variants (profiles/ub/build_variant.sh <name> ag_blend_backward -D...): bench legs per variant
mkdir -p gpurun_out/r03r
run() { # label env...
  label=$1; shift
  env "$@" timeout 200 python bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-full-step --no-stress > gpurun_out/r03r/bench_$label.json 2> gpurun_out/r03r/bench_$label.err
  python - "$label" <<'PY'
import json, sys
d=json.loads(open(f"gpurun_out/r03r/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:12s} value {d['value']:8.1f}  seq {d['sequential']['views_per_s']:7.1f}  bwd us (1 stream) {d['sequential']['blend_backward_avg_launch_us']:6.1f}  overlapped {d['roofline']['avg_launch_us']:6.1f}")
PY
}
run head X=1
for v in "$@"; do run $v AG_LIB_PATH=$PWD/profiles/ub/ko/libag_$v.so; done
run head_again X=1
