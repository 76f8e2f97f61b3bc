# conv DMA loader: correctness tests, then same-box A/B of the full step (AG_CONV_DMA=0 / 1)
out=gpurun_out/$1; mkdir -p $out
./profiles/ub/glds_test > $out/glds_test.txt 2>&1
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q -m gpu 2>&1 | tail -8 > $out/conv_tests.txt
cat $out/glds_test.txt $out/conv_tests.txt
for v in 1 0 1 0; do
  AG_CONV_DMA=$v timeout 300 python bench_avatar.py --steps 8 --warmup 3 > $out/avatar_dma$v.json 2> $out/avatar_dma$v.err
  python - $out/avatar_dma$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("AG_CONV_DMA=" + sys.argv[2], {k: d.get(k) for k in ("value", "ms_per_step")})
PY
done 2>&1 | tee $out/ab.txt
