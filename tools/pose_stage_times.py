import sys, importlib, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg=e.load_package(); synth=importlib.import_module(e.PKG_NAME+'.synth')
dev=torch.device('cuda',0)
seq=synth.StereoSequence(width=1241,height=376,n_frames=34,seed=20200710,device=dev)
fr=[seq.render(t) for t in range(34)]
P1,P2=seq.proj()
for mode in ("lk","orb"):
    kw=dict(P1=P1,P2=P2)
    if mode=="orb": kw.update(track_mode=pkg.MODE_ORB,min_move2=0.05**2,max_move2=100.0)
    c=pkg.Context(1241,376,device=0,max_batch=1,**kw)
    c.enable_timing(True)
    acc={}; wall=[]
    for t,(l,r) in enumerate(fr):
        torch.cuda.synchronize(); t0=time.perf_counter()
        rc,res=c.add_frame(l,r)
        t1=time.perf_counter()
        tm=dict(c.get_timing())
        if t>=4:
            wall.append((t1-t0)*1e3)
            for k,v in tm.items(): acc.setdefault(k,[]).append(v)
    print(mode,"online ms/pair median %.3f"%np.median(wall),{k:round(float(np.mean(v)),4) for k,v in acc.items()}, int(res['n_tracked']), int(res['ransac_iters']), int(res['lm_iters']))
    c.close()
    # batched pose-stage time, no overlap
    B=32
    c=pkg.Context(1241,376,device=0,max_batch=B,**kw)
    L=torch.stack([f[0] for f in fr[:B+1]]); R=torch.stack([f[1] for f in fr[:B+1]])
    c.enable_timing(True)
    for _ in range(3): c.track_batch(L,R)
    c.get_timing()
    for _ in range(5): c.track_batch(L,R)
    print(mode,"batch32 stage ms",{k:round(v,4) for k,v in c.get_timing()})
    c.close()
