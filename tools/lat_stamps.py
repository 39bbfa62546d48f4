import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.getcwd())
import myscaledb_amd.capi as capi
from bench import data_model, ivf_params
dev = torch.device("cuda", 0)
n, d, nlist, nprobe, k = 1_000_000, 768, 1024, 32, 10
x, qd, _ = data_model(os.environ.get("LAT_DATA", "blobs03"), n, 256, d, dev)
ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, ivf_params(nlist, n))
ix.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE); ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE); ix.build()
qh = qd.cpu().numpy()
out = (C.c_ulonglong * 16)()
capi.lib().msvs_lat_debug(out)
acc = np.zeros(9)
for i in range(120):
    ix.search(qh[i:i+1], k, "nprobe=32")
    capi.lib().msvs_lat_debug(out)
    v = np.array(list(out)[:9], dtype=np.float64)
    if i >= 20:
        acc += np.array([v[1]-v[0], v[2]-v[1], v[3]-v[2], v[4]-v[0], v[5]-v[4], v[6]-v[5], v[7]-v[6], v[8]-v[7], v[8]-v[0]])
acc /= 100 * 100.0  # 100 MHz -> us
print("coarse last block: scan %.1f | publish+arrive %.1f | merge %.1f   ||  scan kernel last block start - coarse start %.1f" % tuple(acc[:4]))
print("scan last block: cut %.1f | stage+scan+blockmerge %.1f | publish+arrive %.1f | final merge+store %.1f  || coarse start -> end %.1f" % tuple(acc[4:]))
