import numpy as np, sys
sys.path.insert(0,'.')
import myscaledb_amd.capi as capi
from oracle import oracle as o
rng = np.random.default_rng(3)
x = rng.standard_normal((300, 77), dtype=np.float32) * 5
x[5] = 0; x[6] = 1e-5
a = capi.normalize(x); b = o.normalize_rows(x)
bad = np.argwhere(a.view(np.uint32) != b.view(np.uint32))
print("mismatch", len(bad), "of", a.size, "rows", sorted(set(bad[:,0].tolist()))[:20])
for r,c in bad[:8]:
    print(r,c, x[r,c], a[r,c], b[r,c], a[r,c].view(np.uint32)-b[r,c].view(np.uint32))
# per-row: is the norm differing?
s64 = np.sqrt((x.astype(np.float64)**2).sum(1))
r=bad[0][0]; print("row", r, "x/a", (x[r,:4]/a[r,:4]), "x/b", x[r,:4]/b[r,:4], s64[r])
