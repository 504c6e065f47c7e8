"""Which entries of the 4x4x4 covariance differ from numpy.cov (found the missing hazard pad before the first asm MFMA of a tile, round 3)."""
import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from srcfinder_amd import _ffi
L=_ffi.lib(); P=_ffi.ptr; st=_ffi.stream_ptr()
rng=np.random.default_rng(0)
n,p,C=512,72,3
x=rng.standard_normal((C,n,p)).astype(np.float32)+3
mask=np.ones((C,n),np.uint8); mask[:,5]=0; x[:,5,:]=np.nan
xt=torch.as_tensor(x).cuda(); mk=torch.as_tensor(mask).cuda()
nuse=torch.empty(C,dtype=torch.int32,device='cuda'); mu=torch.empty((C,p),dtype=torch.float64,device='cuda'); S=torch.empty((C,p,p),dtype=torch.float64,device='cuda')
ws=torch.empty(L.sf_cmf_workspace_bytes(n,p,C,201),dtype=torch.uint8,device='cuda')
_ffi.check(L.sf_cmf_column_mean(P(xt),0,P(mk),n,p,C,P(nuse),P(mu),P(ws),st),"mean")
_ffi.check(L.sf_cmf_covariance(P(xt),0,P(mk),P(nuse),P(mu),n,p,C,P(S),P(ws),st),"cov")
torch.cuda.synchronize()
Sn=S.cpu().numpy()
for c in range(C):
    xv=x[c][mask[c]!=0].astype(np.float64); ref=np.cov(xv.T)
    bad=np.argwhere(np.abs(Sn[c]-ref)>1e-9*np.abs(ref).max())
    print(c,len(bad),bad[:20].tolist())
