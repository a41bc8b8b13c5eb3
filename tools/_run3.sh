cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r04b_time.txt
timeout -k 10 900 bash tools/ab_time.sh r04b ns1 ns2 ns4 ns7 > /dev/null 2>&1
cat gpurun_out/r04b_time.txt | sed 's/(.mdct_long_kernel., [0-9.]*), //; s/, (.bitalloc.*//'
