cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_golden.py -x -q 2>&1 | tail -3 && bash tools/ab_time.sh r04v2 prev
