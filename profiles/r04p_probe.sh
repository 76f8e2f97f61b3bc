mkdir -p gpurun_out/r04p; O=gpurun_out/r04p; export TMPDIR=/tmp
R=$PWD
for m in split_f16 split_bf16x3; do
rm -rf /tmp/prof_fs; ( cd /tmp && AG_CONV_MATH=$m rocprofv3 --kernel-trace --stats -d /tmp/prof_fs -o fs -- python $R/profiles/fullstep_prof.py 8 > /dev/null 2>&1 )
db=$(find /tmp/prof_fs -name "*.db" | head -1); python profiles/summarize_rocprof.py "$db" $O/fullstep_kernel_stats_$m.csv | head -12 | cut -c1-150
done
