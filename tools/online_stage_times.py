import sys, importlib, numpy as np, torch
sys.path.insert(0,'/root/repo')
import __graft_entry__ as e
pkg=e.load_package(); synth=importlib.import_module(e.PKG_NAME+'.synth')
dev=torch.device('cuda',0)
seq=synth.StereoSequence(width=1241,height=376,n_frames=12,seed=20200710,device=dev)
fr=[seq.render(t) for t in range(12)]
P1,P2=seq.proj()
c=pkg.Context(1241,376,device=0,max_batch=1,P1=P1,P2=P2)
c.enable_timing(True)
acc={}
for t,(l,r) in enumerate(fr):
    rc,res=c.add_frame(l,r)
    tm=dict(c.get_timing())
    if t>=4:
        for k,v in tm.items(): acc.setdefault(k,[]).append(v)
print({k:round(float(np.mean(v)),4) for k,v in acc.items()}, int(res['n_tracked']), int(res['ransac_iters']), int(res['lm_iters']))
