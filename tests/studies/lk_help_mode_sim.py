#!/usr/bin/env python3
"""lk_help_mode_sim.py -- what "idle groups help the active pairs" would save in the LK kernel's iteration loop (CPU only).

Replays the oracle's per-pair iteration counts (pco_set_lk_iter_trace) through the kernel's wavefront mapping.  Model:
an issued iteration costs O + P instructions (O = everything around the pixel loop, P = the pixel loop); when at most
8 / k pairs of each half-wave are still active, k groups could share a pair: P / k, plus E instructions of exchange per
iteration and S per change of mode.  The numbers behind DESIGN.md section 4 "What is left" (a).

    python tests/studies/lk_help_mode_sim.py
"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle
from polychase_amd import synth
W,H,F=1280,720,100
skips=(-8,-4,-2,-1,1,2,4,8)
clip=synth.NoiseClip(W,H,300)
g={t:oracle.rgb2gray(clip.frame(t)) for t in [F+s for s in (0,)+skips]}
kps=oracle.gftt(g[F]); p0=oracle.Pyramid(g[F]); L=oracle.lib()
L.pco_set_lk_iter_trace.argtypes=[C.c_void_p,C.c_int]
its=np.zeros((8,len(kps),4),np.int32)
for k,s in enumerate(skips):
    buf=np.zeros((len(kps),4),np.uint8); L.pco_set_lk_iter_trace(buf.ctypes.data,4)
    oracle.lk(p0,oracle.Pyramid(g[F+s]),kps); L.pco_set_lk_iter_trace(None,0); its[k]=buf
T,N,LV=its.shape
tile=(kps[:,1].astype(int)//64)*64+(kps[:,0].astype(int)//64)
I=its[:,np.argsort(tile,kind="stable"),:]
n2=N//2*2
A=I[:,:n2,:].reshape(T,n2//2,2,LV).transpose(1,2,0,3)   # wave, half, target, level
O,P,E,S=65.0,94.0,12.0,30.0
def pow2ceil(n):
    p=1
    while p<n: p*=2
    return p
cur=0.0; new=0.0; new_half_indep=0.0
for w in range(A.shape[0]):
    for l in range(LV):
        it=A[w,:,:,l]       # 2 x 8
        mx=int(it.max())
        cur+=mx*(O+P)
        prevk=1
        for j in range(mx):
            ks=[]
            for h in range(2):
                n=int((it[h]>j).sum())
                if n>0: ks.append(8//pow2ceil(n))
            k=min(ks)
            c=O+P/k+(E if k>1 else 0)
            if k!=prevk: c+=S; prevk=k
            new+=c
print("waves",A.shape[0],"current",cur/A.shape[0],"help",new/A.shape[0],"ratio",new/cur)

def run(max_k, levels, E=12.0, S=30.0, P=90.0, O=56.0):
    cur=0.0; new=0.0
    for w in range(A.shape[0]):
        for l in range(LV):
            it=A[w,:,:,l]; mx=int(it.max()); cur+=mx*(O+P)
            prevk=1
            for j in range(mx):
                if l not in levels:
                    new+=O+P; continue
                ks=[]
                for h in range(2):
                    n=int((it[h]>j).sum())
                    if n>0: ks.append(min(max_k, 8//pow2ceil(n)))
                k=min(ks); c=O+P/k+(E if k>1 else 0)
                if k!=prevk: c+=S; prevk=k
                new+=c
    return new/cur
for mk in (2,4,8):
    for lv in ((0,),(0,1),(0,1,2,3)):
        print("max_k",mk,"levels",lv,"ratio",round(run(mk,lv),4))
print("E=20,S=40 max_k 2 all", round(run(2,(0,1,2,3),E=20.0,S=40.0),4), " max_k 8 all", round(run(8,(0,1,2,3),E=20.0,S=40.0),4))
