"""Diagnostic: the drop-in entry with host buffers in and out (hm355_compress_slices through Encoder.compress: upload, search, download of every
picture's reconstruction / decisions / coefficients) on a batch of 4K pictures -- the PCIe-inclusive rate quoted in DESIGN section 7 (never bench.py's value).
usage: pcie_inclusive.py [pictures=192]"""
import json, os, sys, time
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'hm-16.2_amd')]
import hm355, synth
F = int(sys.argv[1]) if len(sys.argv) > 1 else 192
w, h, bd = 3840, 2160, 10
frames = [synth.frame(w, h, bd, f, 1234) for f in range(4)]
enc = hm355.Encoder(w, h, bd, 1, F)
enc.compress(frames[:2], 32)                       # first call: allocations
t = time.time(); res = enc.compress([frames[i % 4] for i in range(F)], 32); dt = time.time() - t
t = time.time()
for i in range(F): enc.upload(i, frames[i % 4])
up = time.time() - t
ms, _ = enc.run(F, 32)
t = time.time()
for i in range(F): enc.download(i)
down = time.time() - t
n = enc.num_ctus * F
print(json.dumps({"pictures": F, "ctus": n, "compress_call_s": dt, "pcie_inclusive_ctu_per_s": n / dt, "upload_s": up, "search_kernel_s": ms / 1e3, "download_and_repack_s": down,
                  "hbm_resident_ctu_per_s": n / (ms / 1e3), "build_id": enc.lib.hm355_build_id().decode()}))
