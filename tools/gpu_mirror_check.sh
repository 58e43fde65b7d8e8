#!/bin/bash
# The kernel-written host mirrors of reward / done (ABI v10): a guarded first run, then the drop-in tests and timing.
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-mirror}; mkdir -p $O
timeout 120 python - > $O/mirror_first.txt 2>&1 <<'PY'
import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch, bpp_amd
size = (10, 10, 10)
pool = bpp_amd.sequences.cut2_pool(size, 64, seed=0)
for fresh in (False, True):
    env = bpp_amd.BppVecEnv(4099, size, enable_rotation=True, pool=pool, fresh_outputs=fresh)
    env.reset()
    for t in range(30):
        a = env.sample_feasible(seed=2, step=t)
        if t % 5 == 2:
            a[::3] = -1
        obs, rew, done, infos = env.step(a)
        r = env._res
        assert np.array_equal(rew.numpy()[:, 0], r.reward.cpu().numpy()[:, 0]), t
        assert np.array_equal(done, r.done.cpu().numpy().astype(bool)), t
    print("mirror ok fresh_outputs=%s, %d episodes ended in the last step" % (fresh, int(done.sum())))
PY
cat $O/mirror_first.txt | grep -v amdgpu.ids
if grep -q "mirror ok fresh_outputs=True" $O/mirror_first.txt; then
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_lookahead.py -m gpu -q -k "dropin or late or lookahead" > $O/pytest_dropin.log 2>&1; tail -2 $O/pytest_dropin.log
  python tools/bench_dropin_step.py > $O/dropin_step.json 2> $O/err.txt; cat $O/dropin_step.json
  python bench.py --no-cpu-baseline --no-past-l3 > $O/bench_quick.json 2>> $O/err.txt; python -c "import json; d=json.load(open('$O/bench_quick.json')); print(d['value']/1e6, d['roofline']['launch_us'])"
fi
