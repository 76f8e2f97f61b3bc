# same-box A/B of library variants, per-kernel one-stream times (bench.py's roofline_raster_kernels leg): bash profiles/ab_kernels_r05b.sh <variant>...
mkdir -p gpurun_out/r05b
run() { # label env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-full-step --no-stress > gpurun_out/r05b/bench_$label.json 2> gpurun_out/r05b/bench_$label.err
  python - "$label" <<'PY'
import json, sys
d=json.loads(open(f"gpurun_out/r05b/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
k=d['roofline_raster_kernels']
names=[n for n in k if n.endswith('_kernel')]
print(f"{sys.argv[1]:12s} value {d['value']:7.1f} seq {d['sequential']['views_per_s']:7.1f} sum {k['one_stream_sum_us']:6.1f} | " + " ".join(f"{n.replace('_kernel','')[:12]} {k[n]['avg_launch_us']:5.1f}" for n in names))
PY
}
run head X=1
for v in "$@"; do run $v AG_LIB_PATH=$PWD/profiles/ub/ko/libag_$v.so; done
run head_again X=1
