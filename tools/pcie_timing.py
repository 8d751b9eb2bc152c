"""Host<->device transfer cost of the boundary (hm355_upload / hm355_download) per 4K picture, to quote the
PCIe-inclusive rate next to the HBM-resident rate bench.py reports."""
import os, sys, time
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'hm-16.2_amd')]
import hm355, synth
w, h, bd, F = 3840, 2160, 10, 8
enc = hm355.Encoder(w, h, bd, 1, F)
planes = synth.frame(w, h, bd, 0, 1234)
enc.upload(0, planes)
t = time.time()
for i in range(F): enc.upload(i, planes)
up = (time.time() - t) / F
ms, _ = enc.run(F, 32)
enc.download(0)
t = time.time()
for i in range(F): enc.download(i)
down = (time.time() - t) / F
print(f"per 4K picture: upload {up*1e3:.1f} ms, download {down*1e3:.1f} ms (incl. host-side repacking), search at the bench rate {2040/6316*1e3:.0f} ms")
