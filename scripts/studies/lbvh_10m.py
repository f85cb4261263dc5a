import sys, numpy as np
sys.path.insert(0,'/root/repo')
import torch, ntrace_amd as nt
from ntrace_amd import scenes
dev=torch.device("cuda:0")
def up(a): return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
tri,pos,cam=scenes.courtyard(); n=tri.shape[0]
capn,capw,capi=nt.lbvh_capacity(n); d_tri,d_pos=up(tri),up(pos)
dn=torch.zeros(capn,dtype=torch.uint8,device=dev); dw=torch.zeros(capw,dtype=torch.uint8,device=dev); di=torch.zeros(capi,dtype=torch.uint8,device=dev)
best=None
for _ in range(5):
    r=nt.lbvh_build(n,d_tri.data_ptr(),pos.shape[0],d_pos.data_ptr(),pos.min(0),pos.max(0),8,0.001,dn.data_ptr(),capn,dw.data_ptr(),capw,di.data_ptr(),capi)
    best=r if best is None or r.seconds<best.seconds else best
print(best.as_dict())
