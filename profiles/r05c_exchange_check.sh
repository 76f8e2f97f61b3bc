cd /root/repo
O=gpurun_out/r05c; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-full-step --no-cpu-baseline --no-stress 2>$O/n1.err | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("N1", d["value"], d["config"]["parallelism"], d["config"]["exchange_every_steps"])'
for E in 0 2 1; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 30 --warmup 10 --exchange-every $E > $O/n2_e$E.json 2> $O/n2_e$E.err
python -c 'import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0]); print("N2 E", sys.argv[2], d["value"], d["value_blocks"]["views_per_s"], d["exchange_check"], d["config"]["exchange_every_steps"])' $O/n2_e$E.json $E || tail -20 $O/n2_e$E.err
done
python -m pytest tests/test_zz_bench_contract_gpu.py -q -k two_ranks 2>&1 | tail -3
