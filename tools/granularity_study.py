#!/usr/bin/env python3
"""Which fraction of all unordered pairs of the bench configuration (n = 262144, rho = 0.8, rc = 0.49 L, jittered
lattice) lies in a (row unit x column unit) pair of exact bounding boxes that cannot be proven outside the cutoff --
i.e. what a pair kernel has to evaluate when it can skip at that granularity.  Units = leaves of the k-d order the
engine uses (median splits along x, y, z, ...), extended to 8-particle leaves.  64 x 64 is the present kernel (its
measured figure, 69.7 %, is reproduced); 49.3 % of the pairs are really inside the cutoff.  CPU only, ~35 min (numpy).
Output: profiles/r02_granularity_study.txt.  Measurement tool, not product code."""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import ljmd_amd
from ljmd_amd import synthetic
n=262144
p,r,v=synthetic.make_config(n)
L=p.box_length; rc=p.rc
r=np.mod(r,L)
def kd(idx, depth, leaf):
    # returns ordering with leaves of size leaf
    if len(idx)<=leaf: return [idx]
    ax=depth%3
    o=idx[np.argsort(r[ax,idx],kind='stable')]
    h=len(o)//2
    return kd(o[:h],depth+1,leaf)+kd(o[h:],depth+1,leaf)
leaves=kd(np.arange(n),0,8)
order=np.concatenate(leaves)
rs=r[:,order]
def boxes(sz):
    q=rs.reshape(3,n//sz,sz)
    return q.min(axis=2), q.max(axis=2)
def gap2(lo1,hi1,lo2,hi2):
    # lo1.. shape (3,A), lo2 (3,B) -> (A,B) lower bound of squared MIC distance
    d2=0
    for k in range(3):
        lo=lo1[k][:,None]-hi2[k][None,:]   # min of xi-xj
        hi=hi1[k][:,None]-lo2[k][None,:]
        g=np.full(lo.shape,np.inf)
        zero=np.zeros(lo.shape,bool)
        for m in (-1,0,1):
            c=m*L
            zero|=(lo<=c)&(c<=hi)
            g=np.minimum(g,np.minimum(np.abs(lo-c),np.abs(hi-c)))
        g[zero]=0
        d2=d2+g*g
    return d2
def frac(szA,szB,chunk=512):
    loA,hiA=boxes(szA); loB,hiB=boxes(szB)
    A=loA.shape[1]; B=loB.shape[1]
    cnt=0
    for a0 in range(0,A,chunk):
        d2=gap2(loA[:,a0:a0+chunk],hiA[:,a0:a0+chunk],loB,hiB)
        cnt+=np.count_nonzero(d2<=rc*rc)
    return cnt/(A*B)
for a,b in ((64,64),(256,64),(64,16),(64,8),(16,16),(8,8),(32,32)):
    print(a,b,frac(a,b,chunk=max(16,4096*64//(a*64))),flush=True)
