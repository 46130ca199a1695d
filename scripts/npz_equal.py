import numpy as np,sys
a=dict(np.load(sys.argv[1])); b=dict(np.load(sys.argv[2]))
print({k:(bool(np.array_equal(a[k],b[k])), float(np.abs(a[k].astype(np.float64)-b[k]).max()) if a[k].dtype.kind=='f' else None) for k in a if k!='kernel'})
