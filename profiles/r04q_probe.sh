mkdir -p gpurun_out/r04q; O=gpurun_out/r04q; export TMPDIR=/tmp
R=$PWD
for t in b c; do AG_F16_TILE=$t python -m pytest tests/test_conv_gpu.py -m gpu -q -p no:cacheprovider -x -k "split_f16" 2>&1 | tail -2; done
for t in a b c; do
rm -rf /tmp/prof_fs; ( cd /tmp && AG_F16_TILE=$t AG_CONV_MATH=split_f16 rocprofv3 --kernel-trace --stats -d /tmp/prof_fs -o fs -- python $R/profiles/fullstep_prof.py 8 > /dev/null 2>&1 )
db=$(find /tmp/prof_fs -name "*.db" | head -1); echo "--- tile $t"; python profiles/summarize_rocprof.py "$db" $O/fullstep_kernel_stats_f16_$t.csv | grep "gather_conv\|wgrad" | cut -c1-130
done
