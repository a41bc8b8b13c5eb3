cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04c_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r04c_tests.txt
tail -4 gpurun_out/r04c_tests.txt
timeout -k 10 300 python tools/smr_shapes_bench.py 2>&1 | grep -v amdgpu.ids | tail -12
