cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_chain.py -m gpu -x -q 2>&1 | tail -3
timeout -k 10 1000 python tools/pac_sweep.py > gpurun_out/r04_pac_sweep.txt 2>&1; echo "pac rc $?"
tail -6 gpurun_out/r04_pac_sweep.txt
