#!/bin/bash
# store-bandwidth calibration, more forms: 8-byte, nontemporal 16-byte, one contiguous chunk per workgroup
set -u
TAG=${1:-r4zk}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
tools/ubench calib > $O/calib.jsonl 2> $O/err.log; cat $O/calib.jsonl
