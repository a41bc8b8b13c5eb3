cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python bench.py > gpurun_out/r04b_bench.json 2> gpurun_out/r04b_bench.err; echo "rc $?"
tail -c 3000 gpurun_out/r04b_bench.err | tail -5
python tools/show_bench.py gpurun_out/r04b_bench.json 2>&1 | head -120
