cd $GRAFT_REPO_ROOT
timeout -k 10 1150 python tools/parity_sweep.py $@ > gpurun_out/r04_parity_sweep_final_$TAG.txt 2>&1
grep "RESULT\|MISMATCH" gpurun_out/r04_parity_sweep_final_$TAG.txt
