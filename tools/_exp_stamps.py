import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from rpeflow_amd import pointconv as PC, _lib
from rpeflow_amd.csrc import k_nearest_neighbor
dev="cuda:0"; torch.manual_seed(0)
lib=ctypes.CDLL(_lib.LIB_PATH)
for N in (256, 4096):
    xyz=torch.randn(4,3,N,device=dev); knn=k_nearest_neighbor(xyz,xyz,16)
    m=PC.PointConvNoSampling(195,128).to(dev).eval()
    with torch.no_grad():
        packed=PC.pack_rows(xyz, torch.randn(4,195,N,device=dev))
        for _ in range(20): m(xyz,packed,knn)
    out=np.zeros(256,np.uint64); lib.rpe_debug_pc_stamps(out.ctypes.data_as(ctypes.c_void_p))
    st=out.reshape(-1,2).astype(np.int64); st=st[st[:,0]>0]
    t=st[:,0]-st[0,0]; rt=(st[:,1]-st[0,1])*10  # ns
    print("N",N,"stamps",len(st),"clock GHz", (t[-1])/max(rt[-1],1))
    print("cycles:", t.tolist())
