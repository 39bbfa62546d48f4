import sys, time, numpy as np, torch
sys.path.insert(0,'.')
import myscaledb_amd.capi as capi
from bench import make_data, make_queries
dev=torch.device('cuda',0)
n,d,nlist=1_000_000,768,1024
t=time.time(); model,x=make_data(n,d,1234,dev); torch.cuda.synchronize(); print("gen",time.time()-t)
ix=capi.Index(capi.INDEX_IVFFLAT,capi.METRIC_L2,d,"ncentroids=1024,kmeans_iters=10,train_sample=65536")
t=time.time(); ix.train(x.data_ptr(),n=n,mem=capi.MEM_DEVICE); torch.cuda.synchronize(); print("train",time.time()-t)
t=time.time(); ix.add(x.data_ptr(),n=n,mem=capi.MEM_DEVICE); torch.cuda.synchronize(); print("add",time.time()-t)
t=time.time(); ix.build(); torch.cuda.synchronize(); print("build",time.time()-t)
import ctypes as C
off=np.empty(nlist+1,np.int64)
capi._check(capi.lib().msvs_index_export(ix._h,None,off.ctypes.data_as(C.c_void_p),None,None))
L=np.diff(off); print("list len min/mean/max", L.min(), L.mean(), L.max(), "pcts", np.percentile(L,[1,10,50,90,99]).tolist(), "empty", (L==0).sum())
q=make_queries(model,64,4321,dev).cpu().numpy()
print("scanned rows/query nprobe32:", ix.scanned_rows(q,32)/64)
