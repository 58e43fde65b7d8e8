cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r3o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "dropin or late" > $O/pytest_dropin.log 2>&1; tail -2 $O/pytest_dropin.log
python tools/bench_dropin_step.py > $O/dropin_step.json 2> $O/err.txt; cat $O/dropin_step.json
python bench.py > $O/bench.json 2>> $O/err.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20.json 2>> $O/err.txt
python bench.py --no-cpu-baseline --rotation > $O/bench_rotation.json 2>> $O/err.txt
python bench.py --no-cpu-baseline --size 20 20 20 --envs 32768 --pool 2048 > $O/bench_20x20x20.json 2>> $O/err.txt
python bench.py --no-cpu-baseline --pool-file tests/golden/cut2_dataset_10.npz > $O/bench_primary_pool_cut2_dataset.json 2>> $O/err.txt
for f in bench bench_steps20 bench_rotation bench_20x20x20 bench_primary_pool_cut2_dataset; do python -c "
import json; d=json.load(open('$O/$f.json')); r=d['roofline']; print('$f', round(d['value']/1e6,1), round(r['launch_us'],2), round(r['frac'],3), round(r['launch_us_past_l3'],2), round(r['frac_past_l3'],3), r['traffic_source'], r['limiter'])"; done
