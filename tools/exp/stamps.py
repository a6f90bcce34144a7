import sys, os, importlib, ctypes, numpy as np, torch
sys.path.insert(0,'.')
from __graft_entry__ import load_package
pkg=load_package(); synth=importlib.import_module('fpv_amd.synth')
XS,YS=1280,800; n=1000
base=synth.plasma_frames(64,XS,YS,bits=12)
pred=pkg.Predictor(XS,YS,4,False,max_batch=n); pred.set_delta_raw(base[0])
d=torch.from_numpy(base.reshape(64,-1).view(np.int16)).cuda()
fr=d[torch.arange(n,device='cuda')%64].contiguous().view(torch.uint16)
out=pred.alloc_outputs(n)
pred.predict(fr,out=out); pred.predict(fr,out=out); torch.cuda.synchronize()
ms=pred.time_predict(fr,out,1)
buf=(ctypes.c_uint32*64)()
lib=pkg.hip_lib(); lib.fpvhip_debug_counters.argtypes=[ctypes.c_void_p,ctypes.c_void_p,ctypes.c_uint32]
assert lib.fpvhip_debug_counters(pred.ctx,buf,64)==0
c=np.frombuffer(buf,np.uint32)
seg=c[16:28].view(np.uint64).astype(np.float64)
names=["loop+prefetch issue","wait loads+split","preview+hist_d","delta+low store+ring write","ring reads+CG+high store","CG sample+bookkeeping+flush"]
tot=seg.sum(); print("ms",ms,"tasks",c[6],"total stamped cycles/wave-step:",tot/ (n*YS*XS/1024))
for nme,v in zip(names,seg): print(f"{nme:34s} {v/tot*100:5.1f}%  {v/(n*YS*XS/1024):8.1f} cycles/step")
