#!/bin/bash
export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/${1:-q}; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_stream_supply.py -m gpu -x -q 2>&1 | tail -2
python tools/bench_stream_refill.py --needs 1 2 4 2>/dev/null | cut -c330-
for cfg in "8 5" "16 6" "32 14"; do
  set -- $cfg
  timeout 300 python bench.py --no-cpu-baseline --stream --stream-depth $1 --stream-refill $2 --steps 600 --warmup 100 > $O/b.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$O/b.json"))
print("depth $1 refill $2: %.1f M steps/s %.2f us/step" % (d["value"]/1e6, d["ms_per_step"]*1e3))
PY
done
