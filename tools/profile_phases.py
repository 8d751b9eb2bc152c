"""Diagnostic: per-phase cycle shares of the CTU kernel (needs gpurun_out/libhm355_prof.so built with -DHM355_PROFILE)."""
import ctypes as C, sys, time, numpy as np
sys.path[:0] = ['hm-16.2_amd']
import hm355, synth
lib = hm355.load_library(sys.argv[1])
w, h, F = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
enc = hm355.Encoder(w, h, 10, 1, F, lib=lib)
planes = synth.frame(w, h, 10, 0, 1234)
for i in range(F): enc.upload(i, planes)
t = time.time(); ms, l = enc.run(F, 32); dt = time.time() - t
NP = 44
out = (C.c_ulonglong * (2 * NP))()
lib.hm355_read_profile.argtypes = [C.c_void_p, C.c_void_p]
lib.hm355_read_profile(enc.h_, out)
names = ["RDOQ", "BITS", "ADI", "PRED", "FWD", "INV", "SATD35", "RDOQ4z", "SAVE", "CHROMA", "LUMA", "ENCCU", "TOTAL", "RDOQ4nz", "RDOQ8", "RDOQ16"]   # ids 16.. are the inter phases (tools/inter_timing.py)
tot = out[12]
print(f"{w}x{h} F={F}: {dt:.2f}s, {enc.num_ctus*F/dt:.1f} CTU/s, per-step {ms/l:.1f} ms")
names += [None] * 16 + ["S4LUMA", "S4CHROMA", "CU64", "CU32", "CU16", "CU8_2Nx2N", "CU8_NxN", "S8LUMA", "S4LEAF", "S4CLEAF", "S8CHROMA", "RDOQ32"]
for i, n in enumerate(names):
    if n is None: continue
    print(f"{n:8s} {100.0*out[i]/tot:6.2f}%  calls {out[NP+i]:9d}  cyc/call {out[i]/max(1,out[NP+i]):10.0f}")
