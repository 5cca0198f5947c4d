"""Numerical prototype (CPU, NumPy): accumulated tile B = I + X^T X and the solved row with f32 MFMA order, the split-bf16 (three parts, six products) and the split-f16 (two parts, three products, operands scaled by S) emulations, over small / ordinary / large operands and ill-conditioned rows.  Not shipped; run by hand: python tests/scale/proto_f16.py"""
import numpy as np
f32=np.float32
rng=np.random.default_rng(0)
def split16(x):
    hi=x.astype(np.float16).astype(f32); lo=(x-hi).astype(f32).astype(np.float16).astype(f32); return hi,lo
def bf16(x):
    u = np.asarray(x, dtype=f32).view(np.uint32).astype(np.uint64)
    u = ((u + ((u >> 16) & 1) + 0x7fff) >> 16) << 16
    return u.astype(np.uint32).view(f32)
def acc_f16x3(X,S):
    Xs=(X*f32(S)).astype(f32); hi,lo=split16(Xs)
    acc=np.zeros((X.shape[1],X.shape[1]),f32)
    # accumulate in chunks of 32 entries in f32 like MFMA (products exact)
    for c in range(0,X.shape[0],32):
        h,l=hi[c:c+32].astype(np.float64),lo[c:c+32].astype(np.float64)
        acc=(acc+(l.T@h).astype(f32)).astype(f32); acc=(acc+(h.T@l).astype(f32)).astype(f32); acc=(acc+(h.T@h).astype(f32)).astype(f32)
    return acc/f32(S*S)
def acc_bf16x6(X):
    a1=bf16(X); r=(X-a1).astype(f32); a2=bf16(r); a3=bf16((r-a2).astype(f32))
    acc=np.zeros((X.shape[1],X.shape[1]),f32)
    for c in range(0,X.shape[0],32):
        p=[a.astype(np.float64)[c:c+32] for a in (a1,a2,a3)]
        for i,j in ((2,0),(1,1),(0,2),(1,0),(0,1),(0,0)):
            acc=(acc+(p[i].T@p[j]).astype(f32)).astype(f32)
    return acc
def acc_f32(X):
    acc=np.zeros((X.shape[1],X.shape[1]),f32)
    for c in range(0,X.shape[0],4):
        acc=(acc+(X[c:c+4].astype(np.float64).T@X[c:c+4].astype(np.float64)).astype(f32)).astype(f32)
    return acc
f=128
for sigma,d in ((1.2e-3,100),(0.05,100),(0.5,100),(0.5,1000),(3.0,100),(1.2e-3,3000)):
    V=(rng.standard_normal((d,f))*sigma).astype(f32)
    w=(10*np.log(1+rng.integers(1,6,d))).astype(f32)
    X=(V*np.sqrt(w)[:,None]).astype(f32)
    B64=np.eye(f)+X.astype(np.float64).T@X.astype(np.float64)
    y=(X.astype(np.float64).T@((w+1)/np.sqrt(w)))
    g64=np.linalg.solve(B64,y)
    out=f"sigma={sigma} d={d} cond={np.linalg.cond(B64):.1e}"
    for name,B in (("f32",acc_f32(X)),("bf16x6",acc_bf16x6(X)),("f16x3 S=1",acc_f16x3(X,1.0)),("f16x3 S=2^10",acc_f16x3(X,1024.0)),("f16x3 S=4",acc_f16x3(X,4.0))):
        Bm=np.eye(f)+B.astype(np.float64)
        g=np.linalg.solve(Bm,y)
        out+=f" | {name}: B {np.abs(Bm-B64).max()/np.abs(B64).max():.1e} g {np.linalg.norm(g-g64)/np.linalg.norm(g64):.1e}"
    print(out)
