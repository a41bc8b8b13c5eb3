set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_golden.py -m gpu -x -q > gpurun_out/r04a_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r04a_tests.txt
tail -5 gpurun_out/r04a_tests.txt
MRC_HIP_LIBRARY=$PWD/mrcaudiocodec_amd/libmrc_hip_nodestats.so timeout -k 10 300 python tools/node_stats.py 1024 > gpurun_out/r04a_node_stats.txt 2>&1
cat gpurun_out/r04a_node_stats.txt
rm -f gpurun_out/r04a_time.txt
timeout -k 10 600 bash tools/ab_time.sh r04a nodesoff
