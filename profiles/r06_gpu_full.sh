# full GPU suite + smoke + the driver's bench command; outputs under gpurun_out/$1
out=gpurun_out/$1; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $out/gputests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err
tail -c 2000 $out/bench_driver_cmd.json > $out/bench_driver_cmd_tail2000.txt
cat $out/gputests.txt; tail -3 $out/smoke.txt; cat $out/bench_driver_cmd_tail2000.txt
