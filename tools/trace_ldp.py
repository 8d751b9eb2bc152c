"""Diagnostic: RD-evaluation trace of the HIP P-slice path (needs tools/libhm355_trace.so built with -DHM355_TRACE).
usage: trace_ldp.py <case> <out.txt> [max P slices]"""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "hm-16.2_amd"), os.path.join(ROOT, "tests")]
import common, hm355, synth
lib = hm355.load_library(os.path.join(ROOT, "tools", "libhm355_trace.so"))
lib.hm355_read_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
lib.hm355_read_trace.restype = C.c_longlong
name, out = sys.argv[1], sys.argv[2]
maxp = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg, slices, finals = common.load_ldp_case(name)
enc = hm355.Encoder(cfg["width"], cfg["height"], cfg["bit_depth"], 0, 1, lib=lib)
buf = np.zeros((1 << 21, 3), np.uint64)
with open(out, "w") as f:
    n_p = 0
    for r in slices:
        if int(r["slice_type"]) != 1:
            continue
        planes = synth.frame(cfg["width"], cfg["height"], cfg["bit_depth"], int(r["poc"]), cfg["seed"])
        sp, refs = common.ldp_slice_inputs(r, finals)
        enc.compress_inter(planes, sp, refs)
        n = lib.hm355_read_trace(enc.h_, buf.ctypes.data, len(buf))
        for i in range(n):
            w0, w1, w2 = int(buf[i, 0]), int(buf[i, 1]), buf[i, 2:3].view(np.float64)[0]
            f.write("%d %d %u %u %.3f\n" % (w0 >> 32, w0 & 0xffffffff, w1 >> 32, w1 & 0xffffffff, w2))
        n_p += 1
        if n_p >= maxp:
            break
print("records written")
