cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_golden.py tests/test_gpu_pcm16.py -x -q 2>&1 | tail -2 && bash tools/ab_time.sh r04i && bash tools/phase_counters.sh r04r 131072
