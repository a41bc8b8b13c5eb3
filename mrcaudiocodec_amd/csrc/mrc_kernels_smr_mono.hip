// The mono long-block instantiation of smr_kernel (psychoac.py:134-219 for one channel of a 1024 + 1024 block, the hot
// path's dominant kernel) as a translation unit of its own, so that it can be compiled with LLVM's max-ILP scheduling
// strategy (Makefile: SMR_MONO_SCHED), which this instantiation gains 3 % from and the joint one loses 2 % with.
// The source is mrc_kernels_smr.hip; this unit defines launch_smr_mono_long only.
#define MRC_SMR_TU_MONO 1
#pragma clang diagnostic ignored "-Wunneeded-internal-declaration"    // (the short block's helpers have no user in this unit)
#include "mrc_kernels_smr.hip"
