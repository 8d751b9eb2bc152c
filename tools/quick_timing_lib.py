import sys, time, os
sys.path[:0]=[os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'hm-16.2_amd')]
import hm355, synth
lib = hm355.load_library(sys.argv[1])
w,h,bd,F=int(sys.argv[2]),int(sys.argv[3]),10,int(sys.argv[4])
enc=hm355.Encoder(w,h,bd,1,F,lib=lib)
planes=synth.frame(w,h,bd,0,1234)
for i in range(F): enc.upload(i,planes)
t=time.time(); ms,l=enc.run(F,32); dt=time.time()-t
print(f"{sys.argv[1]} {w}x{h} F={F}: kernel {ms:.1f} ms, {enc.num_ctus*F/dt:.1f} CTU/s")
