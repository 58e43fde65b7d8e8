#!/usr/bin/env python3
"""Average every PMC counter per kernel name from rocprofv3 --pmc CSV output directories."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if ("bpp_kernel<" in k or "bpp_fast_kernel<" in k) and k.split(">(")[0].endswith(", 0"):
            k = "step"      # MODE == kStep is the last template argument
        elif "bpp_tile_kernel<" in k and k.split("bpp_tile_kernel<")[1].split(">(")[0].split(",")[4].strip() == "0":
            k = "step"      # bpp_tile_kernel<W, L, K, ROT, MODE, EPW, NIT>
        elif "stream_" in k and "_kernel" in k:
            k = k.split("stream_")[1].split("_kernel")[0]      # scan / cut / sort / refill / init
        else:
            k = "sample" if "sample_kernel" in k else ("stats" if "stats_kernel" in k else None)
        if k is None:
            continue
        a = acc[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"])
        a[1] += 1
with open(os.path.join(root, "summary.txt"), "w") as out:
    for k in sorted(acc):
        for c in sorted(acc[k]):
            s, n = acc[k][c]
            line = "%-8s %-24s avg %16.1f  (n=%d)" % (k, c, s / n, n)
            print(line)
            out.write(line + "\n")
