out=gpurun_out/$1; mkdir -p $out
run() { label=$1; shift
  env "$@" timeout 200 python bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-full-step --no-stress > $out/bench_$label.json 2> $out/bench_$label.err
  python - "$out" "$label" <<'PY'
import json, sys
d=json.loads(open(f"{sys.argv[1]}/bench_{sys.argv[2]}.json").read().strip().splitlines()[-1])
rk=d["roofline_raster_kernels"]
print(f"{sys.argv[2]:10s} value {d['value']:8.1f} seq {d['sequential']['views_per_s']:7.1f} fwd us {rk['blend_forward_kernel']['avg_launch_us']:6.2f} bwd us {rk['blend_backward_kernel']['avg_launch_us']:6.2f} sum {rk['one_stream_sum_us']:6.1f}")
PY
}
run head X=1 2>&1
for v in "$@"; do run $v AG_LIB_PATH=$PWD/profiles/ub/ko/libag_$v.so; done 2>&1
run head2 X=1
