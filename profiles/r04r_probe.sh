mkdir -p gpurun_out/r04r; O=gpurun_out/r04r; export TMPDIR=/tmp
R=$PWD
for t in 0 1; do
rm -rf /tmp/prof_fs; ( cd /tmp && AG_CONV_ONE_WG=$t AG_CONV_MATH=split_bf16x3 rocprofv3 --kernel-trace --stats -d /tmp/prof_fs -o fs -- python $R/profiles/fullstep_prof.py 8 > /dev/null 2>&1 )
db=$(find /tmp/prof_fs -name "*.db" | head -1); echo "--- one_wg $t"; python profiles/summarize_rocprof.py "$db" $O/fullstep_kernel_stats_x3_onewg$t.csv | grep "gather_conv" | cut -c1-130
done
