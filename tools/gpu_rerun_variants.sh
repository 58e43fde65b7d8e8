cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r3r; mkdir -p $O
R=/root/repo
BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_legacystats.so timeout 600 python tools/stress_stats.py --launches 20000 > $O/stress_stats_legacy_atomics.json 2> $O/stress.err
cat $O/stress_stats_legacy_atomics.json
for cfg in "10:" "10rot:--rotation" "20:--size 20 20 20 --envs 32768"; do
  name=${cfg%%:*}; args=${cfg#*:}
  BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_abl.so python tools/phase_timeline.py $args > $O/timeline_$name.json 2>> $O/err.txt
done
tools/lds_conflicts_by_phase.sh r3r > /dev/null 2>&1
cat $O/lds_conflicts_by_phase.txt | grep -v "SQ_BUSY\|SQ_INSTS" | head -80
tail -3 $O/err.txt
