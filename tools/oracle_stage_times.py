import sys, numpy as np
sys.path.insert(0,'.')
from oracle import oracle
m=10_000_000
rng=np.random.default_rng(2)
pts=rng.uniform(-np.pi,np.pi,(m,2)).astype(np.float32)
c=(rng.uniform(-.5,.5,m)+1j*rng.uniform(-.5,.5,m)).astype(np.complex64)
for nt in (16,32,64,128):
  oracle.time_nufft(c,pts,[1024,1024],nthreads=nt,repeats=3)
