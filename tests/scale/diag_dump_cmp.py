import numpy as np
a=np.load('gpurun_out/d1_alone.npy'); b=np.load('gpurun_out/d1_noalone.npy')
print("alone: run0==run1", np.array_equal(a[0],a[1]), "run1==run2", np.array_equal(a[1],a[2]))
print("noalone: run0==run1", np.array_equal(b[0],b[1]), "run1==run2", np.array_equal(b[1],b[2]))
for i in range(3):
    for j in range(3):
        d=(a[i]!=b[j]).any(axis=1).sum()
        print(f"alone run{i} vs noalone run{j}: rows that differ {d}")
d = np.abs(a[0]-b[0]); r = np.argmax(d.max(axis=1)); print("worst row", r, "cols differing", np.flatnonzero(a[0][r]!=b[0][r])[:40], "max abs", d.max(), "rel", d.max()/np.abs(a[0]).max())
