cd $GRAFT_REPO_ROOT
MRC_HIP_LIBRARY=$PWD/mrcaudiocodec_amd/libmrc_hip_prof.so timeout -k 10 300 python tools/phase_profile.py 16384 > gpurun_out/r04a_phases.json 2>&1
cat gpurun_out/r04a_phases.json
timeout -k 10 900 bash tools/collect_counters.sh r04a 131072 "" 1 > gpurun_out/r04a_collect.log 2>&1
tail -3 gpurun_out/r04a_collect.log
