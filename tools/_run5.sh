cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r04d_time.txt
timeout -k 10 800 bash tools/ab_time.sh r04d stop0 stop1 stop2 stop3 stop4 stop5 nosweep > /dev/null 2>&1
cat gpurun_out/r04d_time.txt | sed 's/(.mdct_long_kernel., [0-9.]*), //; s/, (.bitalloc.*//'
