import sys, os; sys.path.insert(0,'/root/repo')
import torch, bpp_amd
size=(10,10,10); E=65536
pool=bpp_amd.sequences.cut2_pool(size,512,seed=0)
env=bpp_amd.BppVecEnv(E,size,pool=pool); env.reset()
acts=[]
for t in range(20):
    a=env.sample_feasible(1,t); acts.append(a.clone()); env.step_tensors(a)
torch.cuda.synchronize()
for abl in (0,128,16,15,47,111):
    os.environ["BPP_ABLATE"]=str(abl)
    for rep in range(2):
        for t in range(20): env.step_tensors(acts[t])
    torch.cuda.synchronize()
