cd $GRAFT_REPO_ROOT; O=gpurun_out/r02w; mkdir -p $O
BRIE_BENCH_FORCE_GATHER=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --config c2 --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-psi-check --no-f32-leg --no-e2e > $O/bench_n1_nccl.json 2> $O/err.log
tail -n 1 $O/bench_n1_nccl.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step']); print(json.dumps(d.get('allgather'), indent=1))"
tail -n 4 $O/err.log
