import ctypes, torch
torch.zeros(1, device="cuda")
lib = ctypes.CDLL("/root/repo/mrcaudiocodec_amd/libmrc_hip_occ.so")
out = (ctypes.c_int * 2)()
print(lib.mrc_debug_smr_occupancy(out), list(out))
