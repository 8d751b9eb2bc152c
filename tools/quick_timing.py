import sys, time, numpy as np
import os; sys.path[:0]=[os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'hm-16.2_amd')]
import hm355, synth
w,h,bd,F=int(sys.argv[1]),int(sys.argv[2]),10,int(sys.argv[3])
enc=hm355.Encoder(w,h,bd,1,F)
planes=synth.frame(w,h,bd,0,1234)
for i in range(F): enc.upload(i,planes)
t=time.time(); ms,l=enc.run(F,32); dt=time.time()-t
n=enc.num_ctus*F
print(f"{w}x{h} F={F}: {dt:.3f}s wall, kernel {ms:.1f} ms, {l} launches, {n/dt:.1f} CTU/s, per-step {ms/l:.2f} ms")
