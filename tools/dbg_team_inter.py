"""Diagnostic: one P / B fixture through the team search (HM355_TEAM=1) with the library named by HM355_LIB; prints what differs from the fixture.
usage: dbg_team_inter.py <fixture name>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "hm-16.2_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
os.environ["HM355_TEAM"] = os.environ.get("HM355_TEAM", "1")
import hm355, synth, common
name = sys.argv[1]
lib = hm355.load_library(os.environ["HM355_LIB"]) if os.environ.get("HM355_LIB") else None
cfg, slices, finals = common.load_ldp_case(name)
enc = hm355.Encoder(cfg["width"], cfg["height"], cfg["bit_depth"], cfg["wpp"], max_batch=1, lib=lib)
bad = 0
for r in slices:
    if int(r["slice_type"]) == 2:
        continue
    planes = synth.frame(cfg["width"], cfg["height"], cfg["bit_depth"], int(r["poc"]), cfg["seed"])
    sp, refs = common.ldp_slice_inputs(r, finals)
    rec, ctus, ictus, stats = enc.compress_inter(planes, sp, refs)
    want = r["ctus"]
    for f in ("total_bits", "total_dist", "depth", "part_size", "pred_mode", "cbf", "coeff_y"):
        d = (ctus[f] != want[f]).reshape(len(ctus), -1)
        if d.any():
            bad += 1
            for a in np.nonzero(d.any(axis=1))[0][:4]:
                z = np.nonzero(d[a])[0]
                print(f"POC {int(r['poc'])} {f}: CTU {a} first z {z[:6]} got {ctus[f][a].reshape(-1)[z[:6]]} want {want[f][a].reshape(-1)[z[:6]]}")
    for f in [x for x in ("skip", "merge_flag", "merge_idx", "inter_dir", "mv", "ref_idx", "mvd", "mvp_idx") if x in ictus.dtype.names and x in want.dtype.names]:
        d = (ictus[f] != want[f]).reshape(len(ctus), -1)
        if d.any():
            bad += 1
            for a in np.nonzero(d.any(axis=1))[0][:4]:
                z = np.nonzero(d[a])[0]
                print(f"POC {int(r['poc'])} {f}: CTU {a} first idx {z[:6]} got {ictus[f][a].reshape(-1)[z[:6]]} want {want[f][a].reshape(-1)[z[:6]]}")
print("differences:", bad)
enc.close()
