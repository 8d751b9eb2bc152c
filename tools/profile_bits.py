"""Diagnostic: cycle shares inside the bitstream pass (needs tools/libhm355_prof.so built with -DHM355_PROFILE).
usage: profile_bits.py <lib> <w> <h> <pictures>"""
import ctypes as C, sys, numpy as np
sys.path[:0] = ['hm-16.2_amd']
import hm355, synth
lib = hm355.load_library(sys.argv[1])
w, h, F = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
enc = hm355.Encoder(w, h, 10, 1, F, lib=lib)
planes = synth.frame(w, h, 10, 0, 1234)
for i in range(F): enc.upload(i, planes)
enc.run(F, 32)
lib.hm355_reset_profile.argtypes = [C.c_void_p]; lib.hm355_read_profile.argtypes = [C.c_void_p, C.c_void_p]
lib.hm355_reset_profile(enc.h_)
res = enc.encode_slices_run([dict(slice_type=2, qp=32)] * F)
k, l = C.c_double(), C.c_int(); lib.hm355_last_run_info(enc.h_, C.byref(k), C.byref(l))
NP = 32
out = (C.c_ulonglong * (2 * NP))(); lib.hm355_read_profile(enc.h_, out)
names = {12: "CTU total", 11: "encode_ctu", 4: "code_coeff_nxn", 2: "  staging", 5: "  last position", 6: "  per-group preparation", 7: "  significance flags",
         9: "  levels / signs / remaining", 8: "intra dir syntax"}
tot = out[12]
print(f"{w}x{h} F={F}: kernel {k.value:.1f} ms, {sum(r[2] for r in res)} bins, {sum(sum(len(x) for x in r[0]) for r in res)} bytes")
for i, n in names.items():
    print(f"{n:24s} {100.0 * out[i] / max(1, tot):6.2f}%  calls {out[NP + i]:9d}  cyc/call {out[i] / max(1, out[NP + i]):10.0f}")
