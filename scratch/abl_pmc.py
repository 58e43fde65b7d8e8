import sys, os; sys.path.insert(0,'/root/repo')
import torch, bpp_amd
size=(10,10,10); E=65536
rot = len(sys.argv) > 1 and sys.argv[1] == "rot"
if len(sys.argv) > 1 and sys.argv[1] == "20": size=(20,20,20); E=32768
pool=bpp_amd.sequences.cut2_pool(size,256,seed=0)
env=bpp_amd.BppVecEnv(E,size,enable_rotation=rot,pool=pool); env.reset()
acts=[]
for t in range(12):
    a=env.sample_feasible(1,t); acts.append(a.clone()); env.step_tensors(a)
torch.cuda.synchronize()
for abl in (0,1,2,4,8,32,64,128):
    os.environ["BPP_ABLATE"]=str(abl)
    for t in range(12): env.step_tensors(acts[t])
    torch.cuda.synchronize()
os.environ["BPP_ABLATE"]="0"
nxt=torch.empty_like(acts[0])
for t in range(12): env.step_tensors(acts[t], sample=(1,t,nxt))
torch.cuda.synchronize()
