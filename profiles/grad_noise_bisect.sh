#!/bin/bash
# Which switch of the grouped chain moves the gradient deviations of tests/test_styleunet_net.py against the reference's fp64 golden?
# (round-4 review: convs2.5.noise.weight 7.9e-3 one network at a time vs 4.5e-2 on the grouped chain, pose gradient 5.5e-3 vs 1.7e-2)
#   bash profiles/grad_noise_bisect.sh <out dir>
# One pytest process per configuration (the switches are read at import), reports written by the test itself (AG_TEST_REPORT_DIR).
OUT="${1:-gpurun_out/grad_bisect}"
mkdir -p "$OUT"
run() { # name, env...
  local name="$1"; shift
  mkdir -p "$OUT/$name"
  env AG_TEST_REPORT_DIR="$OUT/$name" "$@" python -m pytest tests/test_styleunet_net.py -q -x -k "grouped_chain_forward_backward or (dual_styleunet_forward_backward and split_f16)" > "$OUT/$name/pytest.log" 2>&1
  tail -1 "$OUT/$name/pytest.log"
}
run default
run comb_split_off AG_COMB_SPLIT=0
run fused_act_off AG_FUSED_ACT=0
run both_off AG_COMB_SPLIT=0 AG_FUSED_ACT=0
run fp32_math AG_CONV_MATH=fp32
python - "$OUT" <<'PY'
import glob, os, re, sys
out = sys.argv[1]
names = ["pose", "convs2.5.noise.weight", "convs1.5.noise.weight", "p50", "p90", "p99", "p100"]
print(f"{'configuration':34s} " + " ".join(f"{n[:22]:>22s}" for n in names))
for d in sorted(glob.glob(os.path.join(out, "*"))):
    for f in sorted(glob.glob(os.path.join(d, "styleunet_grad_report_*.txt"))):
        vals = {}
        for line in open(f):
            m = re.match(r"(p\d+): ours (\S+) ref32 (\S+)", line)
            if m:
                vals[m.group(1)] = f"{float(m.group(2)):.2e}/{float(m.group(3)):.1e}"
            m = re.match(r"ours (\S+) ref32 (\S+) (\S+)", line)
            if m and m.group(3) in names:
                vals[m.group(3)] = f"{float(m.group(1)):.2e}/{float(m.group(2)):.1e}"
        tag = os.path.basename(d) + ":" + os.path.basename(f).replace("styleunet_grad_report_", "").replace(".txt", "")
        print(f"{tag:34s} " + " ".join(f"{vals.get(n, '-'):>22s}" for n in names))
PY
