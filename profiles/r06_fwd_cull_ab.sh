out=gpurun_out/$1; mkdir -p $out
timeout 900 python -m pytest tests/test_raster_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $out/raster_tests.txt
python profiles/fwd_step_stats.py 0 2 5 > $out/fwd_stats_tight.txt 2>&1; AG_LIB_PATH=$PWD/profiles/ub/ko/libag_fstatsdisc.so python profiles/fwd_step_stats.py 0 2 5 > $out/fwd_stats_disc.txt 2>&1
cat $out/fwd_stats_tight.txt $out/fwd_stats_disc.txt
run() { label=$1; shift
  env "$@" timeout 200 python bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-full-step --no-stress > $out/bench_$label.json 2> $out/bench_$label.err
  python - "$out" "$label" <<'PY'
import json, sys
d=json.loads(open(f"{sys.argv[1]}/bench_{sys.argv[2]}.json").read().strip().splitlines()[-1])
rk=d["roofline_raster_kernels"]
print(f"{sys.argv[2]:10s} value {d['value']:8.1f} seq {d['sequential']['views_per_s']:7.1f} fwd us {rk['blend_forward_kernel']['avg_launch_us']:6.2f} bwd us {rk['blend_backward_kernel']['avg_launch_us']:6.2f} sum {rk['one_stream_sum_us']:6.1f}")
PY
}
for i in 1 2; do run tight X=1; run disc AG_LIB_PATH=$PWD/profiles/ub/ko/libag_fwddisc.so; done 2>&1 | tee $out/ab.txt
