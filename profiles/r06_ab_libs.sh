# same-box A/B of blend-backward build variants: bash profiles/r06_ab_libs.sh <outdir> <variant> ... ; variants = profiles/ub/ko/libag_<variant>.so ("head" = in-tree, "wave" = AG_BWD_KERNEL=wave)
out=gpurun_out/$1; shift
mkdir -p $out
run() { # label env...
  label=$1; shift
  env "$@" timeout 200 python bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-full-step --no-stress > $out/bench_$label.json 2> $out/bench_$label.err
  python - "$out" "$label" <<'PY'
import json, sys
d=json.loads(open(f"{sys.argv[1]}/bench_{sys.argv[2]}.json").read().strip().splitlines()[-1])
print(f"{sys.argv[2]:12s} value {d['value']:8.1f}  seq {d['sequential']['views_per_s']:7.1f}  bwd us (1 stream) {d['sequential']['blend_backward_avg_launch_us']:6.1f}  overlapped {d['roofline']['avg_launch_us']:6.1f}")
PY
}
for v in "$@"; do
  case $v in
    head) run head X=1;;
    wave) run wave AG_BWD_KERNEL=wave;;
    *) run $v AG_LIB_PATH=$PWD/profiles/ub/ko/libag_$v.so;;
  esac
done 2>&1 | tee $out/ab.txt
