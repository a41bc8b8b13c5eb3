cd $GRAFT_REPO_ROOT
timeout -k 10 1150 python tools/parity_sweep.py --mono 163840 --joint 65536 --varied 16384 --chunk 2048 > gpurun_out/r04_parity_sweep_nodes_full.txt 2>&1
grep "RESULT\|MISMATCH" gpurun_out/r04_parity_sweep_nodes_full.txt
