#!/usr/bin/env python3
"""Geometry of the tile pairs by system size (CPU, numpy, ~1 min): for the bench family (rho = 0.8, rc = 0.49 L, jittered
lattice, k-d tiles of 64) the share of all tile pairs the cutoff box test keeps, the share of the kept ones with at least one
axis that has no common periodic image (straddle / general forms of the pair loop), and the share that needs no cutoff test
(INNER).  The numbers behind profiles/r04_unit_sweep.txt: why a system of 16 384 particles evaluates twice the pairs that
lie inside the cutoff and one of 262 144 only 1.43x.  Measurement tool, not product code."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ljmd_amd
from ljmd_amd import synthetic
for n in (4096, 8192, 16384, 32768, 65536, 262144):
    p,r,v=synthetic.make_config(n)
    L=p.box_length; rc=p.rc
    r=np.mod(r,L)
    def kd(idx, depth, leaf):
        if len(idx)<=leaf: return [idx]
        # longest-extent axis
        ext=[r[a,idx].max()-r[a,idx].min() for a in range(3)]
        ax=int(np.argmax(ext))
        o=idx[np.argsort(r[ax,idx],kind='stable')]
        h=len(o)//2
        return kd(o[:h],depth+1,leaf)+kd(o[h:],depth+1,leaf)
    order=np.concatenate(kd(np.arange(n),0,64))
    T=n//64
    tiles=r[:,order].reshape(3,T,64)
    lo=tiles.min(axis=2); hi=tiles.max(axis=2)
    rng=np.random.default_rng(1)
    rows=rng.choice(T,size=min(T,32),replace=False)
    kept=0; tot=0; strad=0; inner=0
    for I in rows:
        d2=np.zeros(T); f2=np.zeros(T); st=np.zeros(T,int)
        for k in range(3):
            a=lo[k,I]-hi[k]; b=hi[k,I]-lo[k]
            g=np.full(T,np.inf); zero=np.zeros(T,bool)
            for m in (-1,0,1):
                c=m*L
                zero|=(a<=c)&(c<=b)
                g=np.minimum(g,np.minimum(np.abs(a-c),np.abs(b-c)))
            g[zero]=0
            d2+=g*g
            # uniform image?
            tlo=a/L; thi=b/L; nn=np.round(0.5*(tlo+thi))
            uni=(tlo>nn-0.5)&(thi<nn+0.5)
            st+=(~uni).astype(int)
            far=np.maximum(np.abs(a-nn*L),np.abs(b-nn*L))
            f2+=far*far
        keep=d2<=rc*rc
        kept+=keep.sum(); tot+=T
        strad+=(keep&(st>0)).sum()
        inner+=(keep&(st==0)&(f2<rc*rc)).sum()
    print(f"n={n:7d} L={L:.1f} tiles={T:5d} kept tile pairs {kept/tot:.3f} (inside cutoff 0.489)  kept with >=1 straddle axis {strad/kept:.3f}  INNER {inner/kept:.3f}")
