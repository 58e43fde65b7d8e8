import sys, os; sys.path.insert(0,'/root/repo')
import torch, bpp_amd
size=(10,10,10); E=65536
pool=bpp_amd.sequences.cut2_pool(size,512,seed=0)
env=bpp_amd.BppVecEnv(E,size,pool=pool); env.reset()
acts=[]
for t in range(30):
    a=env.sample_feasible(1,t); acts.append(a.clone()); env.step_tensors(a)
def run(abl):
    os.environ["BPP_ABLATE"]=str(abl)
    ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for rep in range(3):
        for t in range(30):
            ev[t][0].record(); env.step_tensors(acts[t]); ev[t][1].record()
    torch.cuda.synchronize()
    ts=sorted(a.elapsed_time(b)*1e3 for a,b in ev[5:])
    return sum(ts)/len(ts)
for name,abl in (("full",0),("alloff(15)",15),("15+nophase2(32)",47),("15+nophase1(64)",79),("15+32+64",111),("empty(16)",16)):
    print(name, "%.1f us"%run(abl))
for epw in (2,4,8):
    os.environ["BPP_EPW"]=str(epw)
    print("EPW",epw,"empty %.1f"%run(16), "alloff %.1f"%run(15), "full %.1f"%run(0))
