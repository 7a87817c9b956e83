cd $GRAFT_REPO_ROOT; O=gpurun_out/r04r; mkdir -p $O
BRIE_BENCH_SINGLE_DEVICE=1 BRIE_BENCH_STRICT=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 2 > $O/bench_n2_c3.json 2> $O/err.log
tail -n 1 $O/bench_n2_c3.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['genes_per_rank']); print(d['allgather'])"
tail -n 2 $O/err.log
