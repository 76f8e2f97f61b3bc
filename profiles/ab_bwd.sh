# same-box A/B of blend-backward variants: parity tests on the wave kernel, then bench legs
mkdir -p gpurun_out/r03g
AG_BWD_KERNEL=1 timeout 600 python -m pytest tests/test_raster_gpu.py -q 2>&1 | tail -4
run() { # label env...
  label=$1; shift
  env "$@" timeout 200 python bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-full-step > gpurun_out/r03g/bench_$label.json 2> gpurun_out/r03g/bench_$label.err
  python - "$label" <<'PY'
import json, sys
d=json.loads(open(f"gpurun_out/r03g/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:12s} value {d['value']:8.1f}  seq {d['sequential']['views_per_s']:7.1f}  bwd us (1 stream) {d['sequential']['blend_backward_avg_launch_us']:6.1f}  overlapped {d['roofline']['avg_launch_us']:6.1f}")
PY
}
run region AG_BWD_KERNEL=0
run wave6 AG_BWD_KERNEL=1
for o in 5 7 8; do run wave$o AG_BWD_KERNEL=1 AG_LIB_PATH=$PWD/profiles/ub/ko/libag_occ$o.so; done
run wave6_again AG_BWD_KERNEL=1
