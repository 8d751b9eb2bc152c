"""Diagnostic: P-slice throughput with n independent 1080p streams in one call (config 3 shape: low-delay P, 8-bit).
usage: inter_batch_timing.py <w> <h> <wpp> <streams> [nref]"""
import math, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "hm-16.2_amd")]
import hm355, synth
w, h, wpp, S = (int(v) for v in sys.argv[1:5])
nref = int(sys.argv[5]) if len(sys.argv) > 5 else 4
qp, bd = 32, 8
enc = hm355.Encoder(w, h, bd, wpp, S)
n = enc.num_ctus
jobs = []
lam = 0.4624 * 2.0 ** ((qp + 3 - 12) / 3.0) * min(4.0, max(2.0, (qp + 3 - 12) / 6.0))
ref_poc = np.zeros((2, 16), np.int32); ref_poc[0, :nref] = list(range(nref - 1, -1, -1))
sp = dict(qp=qp + 3, chroma_weight=hm355.intra_lambda(qp + 3)[1], poc=nref, cabac_init_type=1, num_ref_idx=(nref, 0), ref_poc=ref_poc,
          col_from_l0=1, col_ref_idx=0, tmvp=1, mvd_l1_zero=0, max_merge_cand=5, check_ldc=1,
          lambda_motion_sad=int(math.floor(65536.0 * math.sqrt(lam))), lambda_motion_sse=int(math.floor(65536.0 * lam)))
sp["lambda"] = lam
t0 = time.time()
for s in range(S):
    frames = [synth.frame(w, h, bd, i, 100 + s) for i in range(nref + 1)]
    res = enc.compress(frames[:nref], qp) if nref <= S else [enc.compress([f], qp)[0] for f in frames[:nref]]
    refs = {}
    for i in range(nref):
        refs[i] = dict(slice_type=2, rec=res[i][0], pred_mode=np.ones((n, 256), np.uint8), mv=[np.zeros((n, 256, 2), np.int16)] * 2,
                       ref_idx=[np.full((n, 256), -1, np.int8)] * 2, num_ref_idx=(0, 0), ref_poc=np.zeros((2, 16), np.int32),
                       ref_long_term=np.zeros((2, 16), np.int32))
    jobs.append((frames[nref], sp, refs))
print(f"prepared {S} streams in {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
out = enc.compress_inter_batch(jobs)
dt = time.time() - t0
k = hm355.C.c_double(); l = hm355.C.c_int(); enc.lib.hm355_last_run_info(enc.h_, hm355.C.byref(k), hm355.C.byref(l))
print(f"{w}x{h} wpp={wpp} refs={nref} streams={S}: {n * S} CTUs, kernel {k.value:.1f} ms ({n * S / k.value * 1000:.1f} CTU/s), call {dt:.2f} s; "
      f"skip {np.mean([float((o[2]['skip'] != 0).mean()) for o in out]):.2f}")
