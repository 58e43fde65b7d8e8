import sys, os; sys.path.insert(0,'/root/repo')
import torch, bpp_amd
def run(size, E, rot, abl):
    os.environ["BPP_ABLATE"]="0"
    pool=bpp_amd.sequences.cut2_pool(size,512,seed=0)
    env=bpp_amd.BppVecEnv(E,size,enable_rotation=rot,pool=pool); env.reset()
    acts=[]
    for t in range(30):
        a=env.sample_feasible(1,t); acts.append(a.clone()); env.step_tensors(a)
    env2=bpp_amd.BppVecEnv(E,size,enable_rotation=rot,pool=pool); env2.reset()
    os.environ["BPP_ABLATE"]=str(abl)
    ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for rep in range(3):
        for t in range(30):
            ev[t][0].record(); env2.step_tensors(acts[t]); ev[t][1].record()
    torch.cuda.synchronize()
    ts=sorted(a.elapsed_time(b)*1e3 for a,b in ev[5:])
    return sum(ts)/len(ts)
for size,E,rot in (((10,10,10),65536,False),((10,10,10),65536,True),((20,20,20),32768,False)):
    base=run(size,E,rot,0)
    print(size,rot,"full %.1f | -build %.1f | -eval %.1f | -maskwrite %.1f | -obswrite %.1f | -build-eval-maskwrite %.1f | all off %.1f"%(
        base, run(size,E,rot,1), run(size,E,rot,2), run(size,E,rot,4), run(size,E,rot,8), run(size,E,rot,7), run(size,E,rot,15)))
