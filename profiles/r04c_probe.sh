mkdir -p gpurun_out/r04c; O=gpurun_out/r04c; export TMPDIR=/tmp
python -m pytest tests/test_grouped_gpu.py tests/test_avatar_net_gpu.py tests/test_zz_guards_gpu.py -m gpu -q -p no:cacheprovider -s > $O/tests.txt 2>&1; tail -15 $O/tests.txt
echo "--- grouped"; python profiles/host_vs_gpu.py 2>&1 | grep -v amdgpu.ids | tee $O/host_vs_gpu_grouped.txt
echo "--- one by one"; AG_GROUPED=0 python profiles/host_vs_gpu.py 2>&1 | grep -v amdgpu.ids | tee $O/host_vs_gpu_onebyone.txt
R=$PWD; rm -rf /tmp/prof_fs; ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_fs -o fs -- python $R/profiles/fullstep_prof.py 8 > /dev/null 2>&1 )
db=$(find /tmp/prof_fs -name "*.db" | head -1); python profiles/summarize_rocprof.py "$db" $O/fullstep_kernel_stats.csv | head -40 | cut -c1-170
