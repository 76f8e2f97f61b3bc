# round 5, second session: exact (quadratic-form) cull of the blend backward against the disc cull it replaces -- parity tests, same-box A/B, work figures
mkdir -p gpurun_out/r05b
timeout 900 python -m pytest tests/test_raster_gpu.py -x -q -m gpu > gpurun_out/r05b/raster_tests.txt 2>&1; tail -3 gpurun_out/r05b/raster_tests.txt
bash profiles/ab_bwd.sh disc 2>&1 | tee gpurun_out/r05b/ab_tight_cull.txt
timeout 300 python profiles/bwd_step_stats.py 0 2 5 2>&1 | tail -4 | tee gpurun_out/r05b/bwd_step_stats_tight.txt
