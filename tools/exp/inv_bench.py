import sys, os, importlib, numpy as np, torch, time
sys.path.insert(0,'.')
from __graft_entry__ import load_package
pkg=load_package(); synth=importlib.import_module('fpv_amd.synth')
from oracle.oracle import Oracle
orc=Oracle()
XS,YS=1280,800
for n in (64, 1024, 4096):
    base=synth.plasma_frames(min(n,8),XS,YS,bits=12)
    pred=pkg.Predictor(XS,YS,4,False,max_batch=n); pred.set_delta_raw(base[0])
    d=torch.from_numpy(base.reshape(base.shape[0],-1).view(np.int16)).cuda()
    fr=d[torch.arange(n,device='cuda')%base.shape[0]].contiguous().view(torch.uint16)
    high,low,pv,flags=pred.predict(fr)
    dh,dl,_=orc.split(base[0],4,False); pred.set_delta_decoded((dh.astype(np.uint16)<<8)|dl)
    out=torch.empty((n,YS,XS),dtype=torch.uint16,device='cuda')
    pred.inverse(high,low,flags,out=out); torch.cuda.synchronize()
    ok=bool((out.view(torch.int16)==(fr.view(torch.int16).view(n,YS,XS)<<4)).all())
    ms=pred.time_inverse(high,low,flags,out,3)/3
    print(f"n={n} ok={ok} flags={np.unique(flags.cpu().numpy())} inverse {ms:.3f} ms/batch = {ms/n:.4f} ms/frame, {n/ms*1e3:.0f} fps, {n*4*XS*YS/ms/1e6:.1f} GB/s")
    pred.close()
