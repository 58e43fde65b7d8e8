import sys, time; sys.path.insert(0,'/root/repo')
import torch, bpp_amd
def run(size, E, rot, mask, n=100):
    pool=bpp_amd.sequences.cut2_pool(size,512,seed=0)
    env=bpp_amd.BppVecEnv(E,size,enable_rotation=rot,pool=pool,compute_mask=True); env.reset()
    # collect a realistic action tape first
    acts=[]
    for t in range(30):
        a=env.sample_feasible(1,t); acts.append(a.clone()); env.step_tensors(a)
    env2=bpp_amd.BppVecEnv(E,size,enable_rotation=rot,pool=pool,compute_mask=mask); env2.reset()
    ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for rep in range(3):
        env2.reset()
        for t in range(30):
            ev[t][0].record(); env2.step_tensors(acts[t]); ev[t][1].record()
    torch.cuda.synchronize()
    ts=sorted(a.elapsed_time(b)*1e3 for a,b in ev[5:])
    return sum(ts)/len(ts)
for size,E,rot in (((10,10,10),65536,False),((10,10,10),65536,True),((20,20,20),32768,False)):
    print(size,rot,"with mask %.1f us   without mask phases %.1f us"%(run(size,E,rot,True),run(size,E,rot,False)))
