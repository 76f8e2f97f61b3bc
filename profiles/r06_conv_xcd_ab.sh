# per-shape A/B of the XCD-aware tile order (AG_CONV_XCD=1, round 4) on single convolutions and on grouped (G = 6) launches through the layer API
out=$PWD/gpurun_out/$1; mkdir -p $out; R=$PWD
cd /tmp
for shape in "64 64 512 512" "128 128 256 256" "256 256 128 128" "512 512 64 64" "256 256 256 256"; do
 for mode in fwd dgrad wgrad; do
  for cfg in "AG_CONV_XCD=0" "AG_CONV_XCD=1"; do
    rm -rf /tmp/kk; env $cfg rocprofv3 --kernel-trace --stats -d /tmp/kk -o p --output-format csv -- python $R/profiles/conv_one.py $shape 3 1 1 12 $mode > /dev/null 2>&1
    f=$(find /tmp/kk -name "*kernel_stats.csv" | head -1)
    python - "$f" "$shape $mode" "$cfg" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
g = [r for r in rows if ("gather_conv" in r["Name"] or "wgrad_split" in r["Name"])]
g.sort(key=lambda r: -float(r["TotalDurationNs"]))
print(f'{sys.argv[2]:26s} {sys.argv[3]:16s} {float(g[0]["AverageNs"])/1e3:8.1f} us  {g[0]["Name"][10:52]}')
PY
  done
 done
done 2>&1 | tee $out/xcd_shapes.txt
