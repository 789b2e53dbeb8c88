import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trajtrack_mpcndqn_rlboost_amd import MpcConfig, BatchSolver, scenes
cfg=MpcConfig(N_hor=20); N=20; B=8192
for fam in ("benchmark","avoidance"):
    sc=scenes.make_family(cfg,B,fam,seed=1236)
    bs=BatchSolver(cfg, order="as_given")
    res=bs.solve(sc["p"]); n_psi,_=bs.last_eval_counts(B)
    ms=res.solve_time_ms
    print(fam,"kernel ms",bs.last_timing()["solve_ms"],"per-problem ms pct 10/50/90/99/max",np.percentile(ms,[10,50,90,99,100]).round(1),"evals pct",np.percentile(n_psi,[10,50,90,100]))
    p=sc["p"]; off=cfg.offsets()
    ref=p[:,off["r"]:off["r"]+3*N].reshape(B,N,3)
    dyn=p[:,off["od"]:off["qstc"]].reshape(B,cfg.Ndynobs,N,6)
    ex=ref[:,None,:,0]-dyn[...,0]; ey=ref[:,None,:,1]-dyn[...,1]
    act=dyn[...,2]>0
    inside=(1-ex**2/(dyn[...,2]+1e-6)**2-ey**2/(dyn[...,3]+1e-6)**2)
    f1=((inside>0)&act).sum(axis=(1,2))           # ref points inside hard ellipses
    f2=np.where(act,np.maximum(inside,0),0).sum(axis=(1,2))
    stc=p[:,off["os"]:off["od"]].reshape(B,cfg.Nstcobs,12)
    h=stc[:,:,None,0:4]-ref[:,None,:,0:1]*stc[:,:,None,4:8]-ref[:,None,:,1:2]*stc[:,:,None,8:12]
    f3=((h.min(axis=3)>0)&(np.abs(stc).sum(axis=2)[:,:,None]>0)).sum(axis=(1,2))
    f4=np.abs(p[:,6]-1.2)
    for name,f in (("ref pts in hard ellipses",f1),("sum hard indicator",f2),("ref pts in polygons",f3),("|v_init - v_ref|",f4)):
        print("   corr(evals,",name,") =",round(float(np.corrcoef(n_psi,f)[0,1]),3))
    print("   status hist",np.bincount(res.status,minlength=3), "evals mean by status", [int(n_psi[res.status==s].mean()) if (res.status==s).any() else 0 for s in (0,1)])
    bs.close()
