cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04d_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r04d_tests.txt
tail -5 gpurun_out/r04d_tests.txt
