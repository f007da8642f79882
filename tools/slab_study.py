#!/usr/bin/env python3
"""How many pairs does a boundary pass of the Newton-3 kernel have to evaluate when the COLUMN tile can be skipped in
units smaller than 64 particles?  Bench configuration (n = 262144, jittered lattice), a sample of row tiles against all
column tiles: 4 column units of 16 as (a) k-d leaves with their exact boxes, (b) slabs of the tile sorted by the
projection on the row->column direction, tested by projection (exact row maximum / support function of the row tile's
box).  CPU only (numpy), ~2 min.  Result (profiles/r03_cluster_passes.txt): 64x64 69.8 %, k-d leaves 65.8 %,
projection slabs 62.1 % (62.2 % with the box support), inside the cutoff 48.9 %.  Measurement tool, not product code."""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import ljmd_amd
from ljmd_amd import synthetic
n=262144
p,r,v=synthetic.make_config(n)
L=p.box_length; rc=p.rc
r=np.mod(r,L)
def kd(idx, depth, leaf):
    if len(idx)<=leaf: return [idx]
    ax=depth%3
    o=idx[np.argsort(r[ax,idx],kind='stable')]
    h=len(o)//2
    return kd(o[:h],depth+1,leaf)+kd(o[h:],depth+1,leaf)
leaves=kd(np.arange(n),0,16)
order=np.concatenate(leaves)
rs=r[:,order]                       # k-d order with 16-leaves inside 64-tiles
T=n//64
tiles=rs.reshape(3,T,64)
lo=tiles.min(axis=2); hi=tiles.max(axis=2); cen=0.5*(lo+hi)
rng=np.random.default_rng(1)
rows=rng.choice(T,size=24,replace=False)
tot=0; kept64=0; inner64=0
res={}
def mic(d): return d-L*np.round(d/L)
for I in rows:
    xi=tiles[:,I,:]                                   # (3,64)
    # tile-level test (AABB, MIC) against all column tiles
    d2=np.zeros(T)
    far2=np.zeros(T)
    for k in range(3):
        a=lo[k,I]-hi[k]; b=hi[k,I]-lo[k]              # range of xi-xj
        g=np.full(T,np.inf); zero=np.zeros(T,bool); f=np.zeros(T)
        for m in (-1,0,1):
            c=m*L
            zero|=(a<=c)&(c<=b)
            g=np.minimum(g,np.minimum(np.abs(a-c),np.abs(b-c)))
        g[zero]=0
        d2+=g*g
    keep=d2<=rc*rc
    keep[I]=False
    J=np.nonzero(keep)[0]
    tot+=T
    kept64+=len(J)
    # exact per-pair distances for kept tiles
    xj=tiles[:,J,:]                                    # (3,nJ,64)
    # shift column tile into the image nearest to row tile centre
    sh=np.round((cen[:,J]-cen[:,I][:,None])/L)*L       # (3,nJ)
    xjs=xj-sh[:,:,None]
    dc=(cen[:,J]-sh)-cen[:,I][:,None]                  # centre displacement (3,nJ)
    nh=dc/np.linalg.norm(dc,axis=0)
    # pair distances (nJ,64,64)
    dd=xi[:,None,:,None]-xjs[:,:,None,:]
    r2=(dd*dd).sum(axis=0)
    inside=r2<rc*rc
    allin=inside.all(axis=(1,2))
    inner64+=allin.sum()
    # projections
    pi=(xi[:,None,:]*nh[:,:,None]).sum(axis=0)         # (nJ,64) row particle projection on each pass's nhat
    pj=(xjs*nh[:,:,None]).sum(axis=0)                  # (nJ,64)
    pimax=pi.max(axis=1)
    supp=(np.maximum(nh*lo[:,I][:,None],nh*hi[:,I][:,None])).sum(axis=0)
    for K in (4,):
        sz=64//K
        # (a) k-d sub-leaves (contiguous slots) with exact AABB test
        cnt_kd=0; cnt_proj=0; cnt_projbox=0
        # kd leaves
        sub=xjs.reshape(3,len(J),K,sz)
        slo=sub.min(axis=3); shi=sub.max(axis=3)       # (3,nJ,K)
        g2=np.zeros((len(J),K))
        for k in range(3):
            a=lo[k,I]-shi[k]; b=hi[k,I]-slo[k]
            g=np.where((a<=0)&(0<=b),0.0,np.minimum(np.abs(a),np.abs(b)))
            g2+=g*g
        kd_keep=(g2<=rc*rc)
        cnt_kd=kd_keep.sum()*sz
        # (b) projection-sorted slabs
        o=np.argsort(pj,axis=1)
        pjs=np.take_along_axis(pj,o,axis=1).reshape(len(J),K,sz)
        pmin=pjs.min(axis=2)
        proj_keep=(pmin-pimax[:,None])<=rc             # cannot prove outside by projection
        supp_keep=(pmin-supp[:,None])<=rc
        # plus AABB of the slab
        xs=np.take_along_axis(xjs,o[None,:,:].repeat(3,0),axis=2).reshape(3,len(J),K,sz)
        slo=xs.min(axis=3); shi=xs.max(axis=3)
        g2=np.zeros((len(J),K))
        for k in range(3):
            a=lo[k,I]-shi[k]; b=hi[k,I]-slo[k]
            g=np.where((a<=0)&(0<=b),0.0,np.minimum(np.abs(a),np.abs(b)))
            g2+=g*g
        box_keep=(g2<=rc*rc)
        # exact: slab has any pair inside
        ins=np.take_along_axis(inside,o[:,None,:].repeat(64,1),axis=2).reshape(len(J),64,K,sz)
        exact_keep=ins.any(axis=(1,3))
        d=res.setdefault(K,np.zeros(6))
        d+= [cnt_kd, (proj_keep).sum()*sz, (proj_keep&box_keep).sum()*sz, exact_keep.sum()*sz, inside.sum()/64.0, supp_keep.sum()*sz]
den=tot*64.0
print("64x64 kept",kept64/tot,"inner share of kept",inner64/kept64)
for K,d in res.items():
    print("K=%d slabs of %d: kd-leaf boxes %.4f  proj-sorted (proj test) %.4f  proj-sorted (proj&box) %.4f  exact-any %.4f   inside %.4f  proj with box support %.4f"%(K,64//K,d[0]/den,d[1]/den,d[2]/den,d[3]/den,d[4]/den,d[5]/den))
