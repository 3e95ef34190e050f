cd $GRAFT_REPO_ROOT
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>&1 | tail -3 | cut -c1-600
