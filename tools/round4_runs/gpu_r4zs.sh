#!/bin/bash
# the freshly rebuilt library: smoke() and the parity / stream tests that touched this round's last changes
set -u
export TMPDIR=/tmp
TAG=${1:-r4zs}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python -m pytest tests/test_stream_supply.py tests/test_gpu_parity.py -m gpu -x -q -k "row_cache or config or stats or golden" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
