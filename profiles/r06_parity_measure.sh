out=gpurun_out/$1; mkdir -p $out
export AG_TEST_REPORT_DIR=$PWD/$out
AG_TEST_EXEMPT_CAP=1e9 timeout 1200 python -m pytest tests/test_raster_gpu.py -x -q -m gpu -s -k "backward or avatar_config2" 2>&1 | grep -E "parity\]|passed|failed|Error" | tee $out/raster_exempt.txt
AG_TEST_FULL_MULT=1e9 AG_TEST_FULL_CAP_SUM=1e9 AG_TEST_FULL_CAP_BLK=1e9 AG_TEST_FULL_CAP_SQ=1e9 timeout 1200 python -m pytest tests/test_styleunet_net.py -x -q -m gpu -s -k "golden" 2>&1 | grep -E "parity\]|passed|failed|Error" | tee $out/styleunet_full.txt
head -5 $out/styleunet_grad_fullstats_*.txt
