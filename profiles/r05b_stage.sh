mkdir -p gpurun_out/r05b
timeout 900 python -m pytest tests/test_raster_gpu.py -x -q -m gpu > gpurun_out/r05b/raster_tests_stage.txt 2>&1; tail -3 gpurun_out/r05b/raster_tests_stage.txt
bash profiles/ab_kernels_r05b.sh pbold prenostage 2>&1 | tee gpurun_out/r05b/ab_stage.txt
