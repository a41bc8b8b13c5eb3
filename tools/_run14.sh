cd $GRAFT_REPO_ROOT
MRC_HIP_LIBRARY=$PWD/mrcaudiocodec_amd/libmrc_hip_cprof.so timeout -k 10 300 python tools/chain_profile.py 8192 2>&1 | grep -v amdgpu.ids
