mkdir -p gpurun_out/r04u; O=gpurun_out/r04u
AG_TEST_REPORT_DIR=$O AG_CONV_MATH=split_f16 python -m pytest tests/test_styleunet_net.py -m gpu -q -p no:cacheprovider -k "golden" 2>&1 | grep -v "^$" | tail -12
head -9 $O/styleunet_grad_report_split_bf16_grouped.txt; echo; head -9 $O/styleunet_grad_report_split_f16.txt; echo;  head -9 $O/styleunet_grad_report_split_bf16.txt
