cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_devpack.py tests/test_gpu_decode.py tests/test_gpu_a_ranks.py -m gpu -x -q 2>&1 | tail -15
