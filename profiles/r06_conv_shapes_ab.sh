# one-convolution A/B over the decoder's shapes: old register-staged loader / DMA 128-wide tiles / DMA big tiles
out=$PWD/gpurun_out/$1; mkdir -p $out; R=$PWD
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $out/conv_tests.txt
cd /tmp
for shape in "512 512 64 64" "256 256 128 128" "256 256 256 256" "128 128 256 256" "64 64 512 512" "512 256 128 128"; do
 for mode in fwd dgrad; do
  for cfg in "AG_CONV_DMA=0" "AG_CONV_DMA=1 AG_CONV_BIG_TILES=0" "AG_CONV_DMA=1 AG_CONV_BIG_TILES=1"; do
    rm -rf /tmp/kk; env $cfg rocprofv3 --kernel-trace --stats -d /tmp/kk -o p --output-format csv -- python $R/profiles/conv_one.py $shape 3 1 1 12 $mode > /dev/null 2>&1
    f=$(find /tmp/kk -name "*kernel_stats.csv" | head -1)
    python - "$f" "$shape $mode" "$cfg" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
g = [r for r in rows if "gather_conv" in r["Name"]]
ps = [r for r in rows if "presplit" in r["Name"]]
print(f'{sys.argv[2]:26s} {sys.argv[3]:38s} gather {float(g[0]["AverageNs"])/1e3:8.1f} us  {g[0]["Name"][10:50]}  presplit {float(ps[0]["AverageNs"])/1e3 if ps else 0:6.1f} us')
PY
  done
 done
done 2>&1 | tee $out/shapes.txt
