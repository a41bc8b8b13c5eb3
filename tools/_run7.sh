cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04b_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r04b_tests.txt
tail -15 gpurun_out/r04b_tests.txt
timeout -k 10 200 python tools/block_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04b_block_probe.txt
