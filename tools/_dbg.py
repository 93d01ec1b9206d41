import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import inputs as I
from tests.test_gpu_knn_binned import raster, dev, both
from oracle import oracle as O
r = I.rng(4500)
H, Wd, N, B = 36, 60, 1024, 2
pts = I.pixel_cloud(r, B, N, H, Wd)
pts[0, 5], pts[0, 17, 0], pts[1, 0, 1], pts[1, 1000] = np.nan, np.inf, -np.inf, np.nan
qry = raster(B, H, Wd)
qry[0, 70], qry[1, 128:192, 0], qry[1, 300, 1] = np.nan, np.inf, -np.inf
(bi, bd), (si, sd) = both(dev(pts), dev(qry))
w = np.argwhere(bi != si)
print(w)
for b, q, _ in w:
    print("query", qry[b, q], "binned", bi[b, q], bd[b, q], pts[b, bi[b, q, 0]], "sweep", si[b, q], sd[b, q], pts[b, si[b, q, 0]])
    oi, od = O.k_nearest_neighbor(pts[b:b+1], qry[b:b+1, q:q+1], 1, return_dists=True)
    print("oracle", oi, od)
