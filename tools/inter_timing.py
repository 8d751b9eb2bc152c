"""Diagnostic: time one P slice on the GPU (reference picture = the HIP path's own I-slice reconstruction of frame 0).
usage: inter_timing.py <w> <h> <wpp> [qp] [nref]"""
import math, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "hm-16.2_amd")]
import hm355, synth
w, h, wpp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
qp = int(sys.argv[4]) if len(sys.argv) > 4 else 32
nref = int(sys.argv[5]) if len(sys.argv) > 5 else 1
bd = 8
lib = hm355.load_library(os.environ["HM355_LIB"]) if os.environ.get("HM355_LIB") else None
enc = hm355.Encoder(w, h, bd, wpp, 1, lib=lib)
n = enc.num_ctus
refs = {}
for i in range(nref):
    (rec, ctus, _), = enc.compress([synth.frame(w, h, bd, i, 7)], qp)
    refs[i] = dict(slice_type=2, rec=rec, pred_mode=np.ones((n, 256), np.uint8), mv=[np.zeros((n, 256, 2), np.int16)] * 2,
                   ref_idx=[np.full((n, 256), -1, np.int8)] * 2, num_ref_idx=(0, 0), ref_poc=np.zeros((2, 16), np.int32),
                   ref_long_term=np.zeros((2, 16), np.int32))
lam = 0.4624 * 2.0 ** ((qp + 3 - 12) / 3.0) * min(4.0, max(2.0, (qp + 3 - 12) / 6.0))
ref_poc = np.zeros((2, 16), np.int32); ref_poc[0, :nref] = list(range(nref - 1, -1, -1))
sp = dict(qp=qp + 3, chroma_weight=hm355.intra_lambda(qp + 3)[1], poc=nref, cabac_init_type=1, num_ref_idx=(nref, 0), ref_poc=ref_poc,
          col_from_l0=1, col_ref_idx=0, tmvp=1, mvd_l1_zero=0, max_merge_cand=5, check_ldc=1,
          lambda_motion_sad=int(math.floor(65536.0 * math.sqrt(lam))), lambda_motion_sse=int(math.floor(65536.0 * lam)))
sp["lambda"] = lam
t0 = time.time()
rec, ctus, ictus, st = enc.compress_inter(synth.frame(w, h, bd, nref, 7), sp, refs)
dt = time.time() - t0
ms, _ = hm355.C.c_double(), None
k = hm355.C.c_double(); l = hm355.C.c_int(); enc.lib.hm355_last_run_info(enc.h_, hm355.C.byref(k), hm355.C.byref(l))
print(f"{w}x{h} wpp={wpp} refs={nref}: {n} CTUs, kernel {k.value:.1f} ms ({n / k.value * 1000:.2f} CTU/s), wall {dt * 1000:.0f} ms; "
      f"skip {float((ictus['skip'] != 0).mean()):.2f} merge {float((ictus['merge_flag'] != 0).mean()):.2f} intra {float((ctus['pred_mode'] == 1).mean()):.2f} bits {st[0]}")
if lib is not None and hasattr(lib, "hm355_read_profile"):
    NP = 44
    out = (hm355.C.c_ulonglong * (2 * NP))()
    lib.hm355_read_profile.argtypes = [hm355.C.c_void_p, hm355.C.c_void_p]
    lib.hm355_read_profile(enc.h_, out)
    names = {0: "RDOQ", 1: "BITS", 2: "ADI", 3: "PRED", 4: "FWD", 5: "INV", 6: "SATD35", 8: "SAVE", 9: "CHROMA", 10: "LUMA", 11: "ENCCU", 12: "TOTAL",
             16: "ME_INT", 17: "ME_FRAC", 18: "AMVP", 19: "MRG_EST", 20: "MC", 21: "IRQ", 22: "IRES", 23: "MRG2N", 24: "INTERCU", 25: "INTRA_IN_P", 26: "IQ_FULL", 27: "IQ_FWD", 28: "IQ_RDOQ", 29: "IQ_BITS", 30: "IQ_INV", 31: "IQ_ENC"}
    tot = out[12]
    for i, nm in names.items():
        print(f"{nm:10s} {100.0 * out[i] / tot:6.2f}%  calls {out[NP + i]:9d}  cyc/call {out[i] / max(1, out[NP + i]):10.0f}")
