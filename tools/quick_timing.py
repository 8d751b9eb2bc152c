"""Diagnostic: time of one search step over F copies of a picture.  usage: quick_timing.py <w> <h> <F> [lib.so]  (a second library for A/B runs on the same box)"""
import sys, time, numpy as np
import os; sys.path[:0]=[os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'hm-16.2_amd')]
import hm355, synth
w,h,bd,F=int(sys.argv[1]),int(sys.argv[2]),10,int(sys.argv[3])
lib=hm355.load_library(sys.argv[4]) if len(sys.argv)>4 else None
enc=hm355.Encoder(w,h,bd,1,F,lib=lib)
frames=[synth.frame(w,h,bd,f,1234) for f in range(min(F,4))]
for i in range(F): enc.upload(i,frames[i%len(frames)])
for rep in range(int(os.environ.get("REPS","1"))):
    t=time.time(); ms,l=enc.run(F,32); dt=time.time()-t
    n=enc.num_ctus*F
    print(f"{sys.argv[4] if len(sys.argv)>4 else 'libhm355.so'} {w}x{h} F={F}: {dt:.3f}s wall, kernel {ms:.1f} ms, {n/dt:.1f} CTU/s", flush=True)
