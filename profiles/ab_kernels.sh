# same-box A/B of build variants (profiles/ub/build_variant.sh <name> <unit> -D...): one-stream per-kernel breakdown + headline per variant
#   bash profiles/ab_kernels.sh <variant> [<variant> ...]
mkdir -p gpurun_out/r03k
run() { # label env...
  label=$1; shift
  env "$@" timeout 200 python bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-full-step --no-stress --breakdown > gpurun_out/r03k/bench_$label.json 2> gpurun_out/r03k/bench_$label.err
  python - "$label" <<'PY'
import json, sys
d=json.loads(open(f"gpurun_out/r03k/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
k=d["kernels_us"]
print(f"{sys.argv[1]:10s} value {d['value']:8.1f}  seq {d['sequential']['views_per_s']:7.1f}  " + "  ".join(f"{n.replace('_kernel','')} {v}" for n, v in k.items() if v))
PY
}
run head X=1
for v in "$@"; do run $v AG_LIB_PATH=$PWD/profiles/ub/ko/libag_$v.so; done
run head2 X=1
