cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
bash tools/final_collect.sh r04f
