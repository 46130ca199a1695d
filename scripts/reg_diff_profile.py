import numpy as np, sys
a=np.load(sys.argv[1])['reg']; b=np.load(sys.argv[2])['reg']
d=np.abs(a.astype(np.float64)-b); print('shape', a.shape, 'max', d.max(), 'scale', np.abs(b).max())
for ax,name in ((1,'z'),(2,'y'),(3,'x')):
    m=d.max(axis=tuple(i for i in range(4) if i!=ax)); print(name, np.array2string(m, precision=1, max_line_width=250))
print('per view', d.max(axis=(1,2,3)))
