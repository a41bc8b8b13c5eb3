cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sensitivity.py -m gpu -x -q -k "threshold or spread or mono_noise or joint_ms or edge or sensitiv or crafted or ordinary or loose" 2>&1 | tail -3
rm -f gpurun_out/r04c_time.txt
timeout -k 10 600 bash tools/ab_time.sh r04c $1 > /dev/null 2>&1
cat gpurun_out/r04c_time.txt | sed 's/(.mdct_long_kernel., [0-9.]*), //; s/, (.bitalloc.*//'
