mkdir -p gpurun_out/r05b
( time python -m pytest tests -m gpu -q -x 2>&1 | tail -6 ) > gpurun_out/r05b/gputests.txt 2>&1; cat gpurun_out/r05b/gputests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
