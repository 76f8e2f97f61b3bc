mkdir -p gpurun_out/r04l; O=gpurun_out/r04l; export TMPDIR=/tmp
R=$PWD; rm -rf /tmp/prof_fs; ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_fs -o fs -- python $R/profiles/fullstep_prof.py 8 > /dev/null 2>&1 )
db=$(find /tmp/prof_fs -name "*.db" | head -1); python profiles/summarize_rocprof.py "$db" $O/fullstep_kernel_stats.csv | head -3 | cut -c1-100
