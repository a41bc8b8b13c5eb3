#!/bin/bash
# one GPU call: tests, bitwise digests old/new, MDCT shape timings old/new.  Stops after a step that was killed.
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
step() { name=$1; shift; timeout -k 10 "$@"; rc=$?; echo "$name rc $rc"; if [ $rc -ge 124 ]; then echo "stopping after $name"; exit $rc; fi; }
PREV=$PWD/mrcaudiocodec_amd/libmrc_hip_prev.so
step digests_new 200 python tools/mdct_bitwise_shapes.py > $O/r04w_digest_new.jsonl 2> $O/r04w_digest_new.err
MRC_HIP_LIBRARY=$PREV step digests_prev 200 python tools/mdct_bitwise_shapes.py > $O/r04w_digest_prev.jsonl 2> $O/r04w_digest_prev.err
if cmp -s $O/r04w_digest_new.jsonl $O/r04w_digest_prev.jsonl; then echo "DIGESTS EQUAL ($(wc -l < $O/r04w_digest_new.jsonl) cases)"; else echo "DIGESTS DIFFER"; diff $O/r04w_digest_new.jsonl $O/r04w_digest_prev.jsonl | head -5; fi
for i in 1 2; do
MRC_HIP_LIBRARY=$PREV step bench_prev 200 python tools/mdct_bench.py 419430 > $O/r04w_mdct_prev_$i.jsonl 2>> $O/r04w_bench.err
step bench_new 200 python tools/mdct_bench.py 419430 > $O/r04w_mdct_new_$i.jsonl 2>> $O/r04w_bench.err
done
step tests 400 python -m pytest tests -m gpu -x -q > $O/r04w_tests.txt 2>&1
tail -3 $O/r04w_tests.txt
echo done
