import sys, ctypes as C, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import lbfgspp_amd as A
from lbfgspp_amd import _lib as L
core,_=A.load()
n=int(sys.argv[1]) if len(sys.argv)>1 else 5000
m=10; npairs=10
h=C.c_void_p()
L.check(core.lbfgsx_create(C.byref(h), 0, n, m, 0, 1))
rng=np.random.default_rng(0)
S=rng.standard_normal((npairs,n)); Y=S*(1+rng.random((npairs,n)))
vp=C.c_void_p
for k in range(npairs):
    L.check(core.lbfgsx_bfgs_add_correction_host(h, S[k].ctypes.data_as(vp), Y[k].ctypes.data_as(vp)))
d=rng.standard_normal(n)
L.check(core.lbfgsx_upload(h, L.VEC_D, d.ctypes.data_as(vp)))
t=2*npairs
g1=np.zeros((t,t)); g2=np.zeros((t,t)); w1=np.zeros(t); w2=np.zeros(t)
f=core.lbfgsx_b_gram; f.restype=C.c_int; f.argtypes=[vp,C.c_int,vp]
L.check(f(h,0,g1.ctypes.data_as(vp)))
f=core.lbfgsx_b_wtv; f.restype=C.c_int; f.argtypes=[vp,C.c_int,C.c_int,vp,vp]
L.check(f(h,0,0,w1.ctypes.data_as(vp),None))
f=core.lbfgsx_b_gram_fused; f.restype=C.c_int; f.argtypes=[vp,C.c_int,C.c_int,vp,vp]
rc=f(h,0,0,g2.ctypes.data_as(vp),w2.ctypes.data_as(vp)); print("fused rc",rc, core.lbfgsx_last_error())
W=np.concatenate([Y,S],0)
gref=W@W.T
print("dd vs numpy  max rel", np.abs(g1-gref).max()/np.abs(gref).max())
print("mfma vs dd   max rel", np.abs(g2-g1).max()/np.abs(g1).max(), "max abs", np.abs(g2-g1).max())
rel=np.abs(g2-g1)/np.abs(g1).max()
print("entries with rel>1e-13:", np.argwhere(rel>1e-13)[:20].tolist())
print("wtv mfma vs dd", np.abs(w2-w1).max()/np.abs(w1).max())
print(np.round(rel[:4,:4]*1e16,1))
