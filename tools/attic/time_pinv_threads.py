import time, numpy as np, os
from threadpoolctl import threadpool_limits, threadpool_info
print([ (d['internal_api'], d['num_threads']) for d in threadpool_info()])
rng=np.random.default_rng(0)
G=rng.normal(size=(5000,532)); H=G.T@G+0.1*np.eye(532); Y=rng.normal(size=(532,524))
def run():
    t=time.perf_counter(); P=np.linalg.pinv(H); M=P@Y; return time.perf_counter()-t
for lim in (None,1,2,4,8,16,32):
    if lim is None:
        ts=[run() for _ in range(4)]
    else:
        with threadpool_limits(limits=lim):
            ts=[run() for _ in range(4)]
    print(lim, ["%.1f ms"%(t*1e3) for t in ts])
t=time.perf_counter()
with threadpool_limits(limits=8): pass
print("ctx overhead us", (time.perf_counter()-t)*1e6)
