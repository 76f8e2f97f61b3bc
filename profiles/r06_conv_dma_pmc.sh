out=$PWD/gpurun_out/$1; mkdir -p $out; R=$PWD
bash profiles/pmc_kernel.sh gather_conv_dma python $R/profiles/conv_one.py 256 256 256 256 3 1 1 10 fwd > $out/sq_dma.txt 2>&1
AG_CONV_DMA=0 bash profiles/pmc_kernel.sh gather_conv_split python $R/profiles/conv_one.py 256 256 256 256 3 1 1 10 fwd > $out/sq_old.txt 2>&1
paste $out/sq_dma.txt $out/sq_old.txt
cd /tmp; for v in 1 0; do AG_CONV_DMA=$v rocprofv3 --kernel-trace --stats -d /tmp/k$v -o p --output-format csv -- python $R/profiles/conv_one.py 256 256 256 256 3 1 1 20 fwd > /dev/null 2>&1; f=$(find /tmp/k$v -name "*kernel_stats.csv" | head -1); echo "AG_CONV_DMA=$v"; head -6 $f | cut -c1-200; done | tee $out/times.txt
