import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd import ops
from tce_rl_amd.mp import ProDMP
from oracle.prodmp_oracle import ProDMPOracle
from oracle import tce_oracle as O
cfg=dict(num_dof=7, num_basis=8, tau=2.0, alpha_phase=3, alpha=10, dt=0.02, basis_bandwidth_factor=3, weights_scale=0.3, goal_scale=0.3, relative_goal=False)
dt=torch.float64
mp=ProDMP(dtype=dt, device='cuda', **cfg); o=ProDMPOracle(dtype=dt, **cfg)
N,T,K=5,100,63
g=torch.Generator().manual_seed(0)
w=torch.randn(N,K,generator=g,dtype=dt); y0=torch.randn(N,7,generator=g,dtype=dt); v0=torch.randn(N,7,generator=g,dtype=dt); t0=torch.zeros(N,dtype=dt)
times=O.get_times(t0,0.02,T)
pos,vel=o.traj(times,w,t0,y0,v0)
tg=ops.times(t0.cuda(),0.02,T)
print("times diff", (tg.cpu()-times).abs().max().item())
out=ops.prodmp_traj(mp,tg,w.cuda(),t0.cuda(),y0.cuda(),v0.cuda()).cpu()
print("pos diff", (out[...,:7]-pos).abs().max().item(), "vel diff", (out[...,7:]-vel).abs().max().item())
B=mp._ws[0].cpu()
xi1,xi2,xi3,xi4,Hp,Hv,_,_=o.basis_terms(times[:1],t0[:1])
print("B c0", (B[:T,0]-xi1[0]).abs().max().item(), "c1", (B[:T,1]-xi2[0]*2.0).abs().max().item(), "Hp", (B[:T,4:13]-Hp[0]).abs().max().item(), "Hv", (B[:T,13:22]-Hv[0]/2.0).abs().max().item())
# zero params: only linear part
out0=ops.prodmp_traj(mp,tg,torch.zeros_like(w).cuda(),t0.cuda(),y0.cuda(),v0.cuda()).cpu()
p0,v0_=o.traj(times,torch.zeros_like(w),t0,y0,v0)
print("zero-param pos diff", (out0[...,:7]-p0).abs().max().item())
outg=ops.prodmp_traj(mp,tg.clone(),w.cuda(),t0.cuda(),y0.cuda(),v0.cuda()).cpu()
print("general path pos diff", (outg[...,:7]-pos).abs().max().item())
