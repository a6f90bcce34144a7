import sys, os, importlib, numpy as np, torch
sys.path.insert(0,'.')
from __graft_entry__ import load_package
pkg=load_package(); synth=importlib.import_module('fpv_amd.synth')
XS,YS=1280,800; n=1000
base=synth.plasma_frames(64,XS,YS,bits=12)
pred=pkg.Predictor(XS,YS,4,False,max_batch=n); pred.set_delta_raw(base[0])
d=torch.from_numpy(base.reshape(64,-1).view(np.int16)).cuda()
fr=d[torch.arange(n,device='cuda')%64].contiguous().view(torch.uint16)
out=pred.alloc_outputs(n)
for wd in (True, False, True, False):
    pred.time_predict(fr,out,3,with_delta=wd)
    ms,km=pred.time_predict(fr,out,10,with_delta=wd,per_kernel=True)
    print('with_delta',wd, [round(k,4) for k in km], np.unique(out[3].cpu().numpy(),return_counts=True))
