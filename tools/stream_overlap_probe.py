import sys, time, json, torch, numpy as np
sys.path.insert(0,'/root/repo')
import irbpp_amd
from bench import make_workload
from irbpp_amd.vec_env import GpuPackingEnv
shapes,seqs,kw=make_workload('blockout')
for nsplit in (1,2,4,8):
    bins=4096; per=bins//nsplit
    envs=[GpuPackingEnv(shapes,seqs,per,device='cuda:0',global_offset=i*per,global_bins=bins,**kw) for i in range(nsplit)]
    streams=[torch.cuda.Stream() for _ in range(nsplit)]
    obs=[]; nxt=[]; act=[]
    for e,s in zip(envs,streams):
        with torch.cuda.stream(s):
            o=e.reset(); obs.append(o); nxt.append(torch.empty_like(o)); act.append(torch.empty((per,),dtype=torch.int32,device='cuda:0'))
    def step_all():
        for i,(e,s) in enumerate(zip(envs,streams)):
            with torch.cuda.stream(s):
                e.policy_minz(obs[i],actions_out=act[i]); e.step(act[i],obs_out=nxt[i])
                obs[i],nxt[i]=nxt[i],obs[i]
    for _ in range(100): step_all()
    torch.cuda.synchronize(); t=time.perf_counter()
    K=100
    for _ in range(K): step_all()
    torch.cuda.synchronize(); dt=time.perf_counter()-t
    print(nsplit, 'streams:', bins*K/dt/1e6, 'M steps/s', dt/K*1e3,'ms/step')
    for e in envs: e.close()
