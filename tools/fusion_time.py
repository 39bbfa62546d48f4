import time, numpy as np, sys
sys.path.insert(0,'.')
import myscaledb_amd.host as host
rng=np.random.default_rng(1)
nq,kv,kt=64,100,100
vd=np.sort(rng.random((nq,kv)).astype(np.float32),axis=1)
vi=np.stack([rng.permutation(10_000_000)[:kv] for _ in range(nq)]).astype(np.int64)
td=-np.sort(-rng.random((nq,kt)).astype(np.float32)*10,axis=1)
ti=np.stack([rng.permutation(10_000_000)[:kt] for _ in range(nq)]).astype(np.int64)
z=np.zeros(100,np.uint64)
for rep in range(3):
    t=time.perf_counter()
    for _ in range(5): host.hybrid_search_batch("rrf",vd,vi,td,ti,10)
    print("batch", (time.perf_counter()-t)/5*1e3,"ms")
    t=time.perf_counter()
    for _ in range(5):
        for q in range(nq): host.hybrid_search("rrf",(vd[q],z,vi[q].astype(np.uint64)),(td[q],z,ti[q].astype(np.uint64)),10)
    print("loop", (time.perf_counter()-t)/5*1e3,"ms")
