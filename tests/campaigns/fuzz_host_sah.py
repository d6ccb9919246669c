#!/usr/bin/env python3
"""CPU campaign: libtrc_host.so's SAH builder against oracle/oracle_sah.cpp, record for record, on random leaf sets of 2..9 000 boxes -- uniform, on a
grid (equal centroids), flat, clustered, denormal-sized, zero-extent.        python3 tests/campaigns/fuzz_host_sah.py <a> <b>"""
import os, sys, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as po
from tracer_amd import abi, host
def rec(nodes,n):
    a=(C.c_uint32*(16*n)).from_address(C.addressof(nodes)); return np.frombuffer(a,dtype=np.uint32).reshape(-1,16).copy()
bad=0; tot=0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng=np.random.default_rng(seed)
    n=int(rng.integers(2,400)) if seed%7 else int(rng.integers(4000,9000))
    mode=seed%5
    c=rng.uniform(-100,100,(n,3)).astype(np.float32)
    if mode==1: c=np.round(c/10)*10            # grid: many equal centroids
    if mode==2: c[:,rng.integers(3)]=np.float32(3.0)   # flat
    if mode==3: c=(rng.normal(0,1e-3,(n,3))+rng.integers(0,3,(n,1))*1000).astype(np.float32)
    if mode==4: c=c*np.float32(1e-20)          # tiny: denormal extents
    e=rng.uniform(0,5,(n,3)).astype(np.float32)*(0 if seed%11==0 else 1)
    e=(e*np.float32(1e-20)).astype(np.float32) if mode==4 else e
    leaves=[host.build_node(tuple(p-x),tuple(p+x),abi.PRIM_SPHERE,i) for i,(p,x) in enumerate(zip(c,e))]
    arr=(abi.BVH*n)(*leaves)
    want=rec(po.sah_build(arr,n),2*n-1); got=rec(host.build_tree(leaves),2*n-1)
    tot+=1
    if not np.array_equal(want,got):
        bad+=1; print("MISMATCH seed",seed,"n",n,"mode",mode, "first", np.nonzero((want!=got).any(axis=1))[0][:5]); 
print("done",tot,"sets, mismatches",bad)
