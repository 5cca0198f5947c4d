import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import importlib
from oracle import wmf_oracle as orc
import proto_whiten as P  # runs its prints; ignore
f32 = np.float32
def variant(Y, C, lam, bias, v64, x64, row64=False):
    Yt = Y.copy(); b=None
    if bias: b = Yt[:,0].copy(); Yt[:,0]=1
    G = Yt.astype(np.float64).T @ Yt.astype(np.float64) + lam*np.eye(Y.shape[1])
    L = np.linalg.cholesky(G); Linv = np.linalg.inv(L)
    V = (Yt.astype(np.float64) @ Linv.T).astype(f32) if v64 else (Yt @ Linv.T.astype(f32)).astype(f32)
    n,f = C.shape[0], Y.shape[1]; Gout = np.zeros((n,f), np.float64 if row64 else f32)
    dt = np.float64 if row64 else f32
    for r in range(n):
        lo,hi = C.indptr[r], C.indptr[r+1]
        if hi==lo: continue
        idx = C.indices[lo:hi]; w = C.data[lo:hi].astype(f32)
        if bias: w = (w-b[idx]).astype(f32)
        Vu = V[idx].astype(dt); w=w.astype(dt); d=len(idx); p=(w+1)
        S = Vu@Vu.T; M = np.eye(d,dtype=dt)+w[:,None]*S
        c = np.linalg.solve(M,p); Gout[r] = Vu.T@c
    X = (Gout.astype(np.float64)@Linv) if x64 else (Gout.astype(f32)@Linv.astype(f32))
    return X.astype(f32)
n,m,k,bias = 3000,1500,128,True
C = P.synth(n,m,12,3); C64=C.astype(np.float64)
Y = orc.init_items(m,k,bias)
Xref = orc.recompute_factors_bias(Y, C64, 0.1, out_dtype='float64')
for v64,x64,row64 in ((0,0,0),(1,0,0),(0,1,0),(1,1,0),(1,1,1),(0,0,1)):
    e = P.relerr(variant(Y,C,0.1,bias,v64,x64,row64), Xref)
    print(f"v64={v64} x64={x64} row64={row64}: max={e.max():.2e} med={np.median(e):.2e}")
