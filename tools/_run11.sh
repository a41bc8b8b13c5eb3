cd $GRAFT_REPO_ROOT
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
MRC_BENCH_BACKEND=gloo timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --c4-frames 65536 > gpurun_out/r04_n2_gloo_rehearsal.json 2> gpurun_out/r04_n2_gloo_rehearsal.err; echo "n2 rc $?"
tail -c 600 gpurun_out/r04_n2_gloo_rehearsal.err
python tools/show_bench.py gpurun_out/r04_n2_gloo_rehearsal.json 2>&1 | head -8
