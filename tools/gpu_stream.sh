cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r3h; mkdir -p $O
python bench.py --no-cpu-baseline --stream > $O/bench_stream_d32_r14.json 2> $O/err.txt
python bench.py --no-cpu-baseline --stream --stream-depth 64 --stream-refill 30 > $O/bench_stream_d64_r30.json 2>> $O/err.txt
python bench.py --no-cpu-baseline --stream --size 20 20 20 --envs 32768 > $O/bench_stream_20_d32_r14.json 2>> $O/err.txt
BPP_STREAM_OVERLAP=0 python bench.py --no-cpu-baseline --stream > $O/bench_stream_d32_r14_serial.json 2>> $O/err.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_stream -o run -- python /root/repo/bench.py --no-cpu-baseline --stream > /dev/null 2>&1)
cp $O/prof_stream/run_kernel_stats.csv $O/kernel_stats_stream_d32_r14.csv; rm -rf $O/prof_stream
for f in $O/bench_stream_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', round(d['value']/1e6,1), 'M', round(d['ms_per_step']*1e3,2),'us/step', 'step kernel pairs', round(d['roofline']['launch_us'],2))"; done
head -8 $O/kernel_stats_stream_d32_r14.csv | cut -c1-160
tail -3 $O/err.txt
