#!/usr/bin/env python3
"""Fold rocprofv3 PMC summaries (tools/profile_pmc.sh -> summary.txt) into profiles/hbm_traffic.json, the file
bench.py reads `roofline.traffic` / `frac_moved` / `limiter` from.

    python tools/pmc_to_json.py [--tracked-prefix profiles/r6_] <key>=<summary.txt> [...]     e.g. 10x10x10_rot0_E65536=gpurun_out/x/pmc_summary_10.txt

--tracked-prefix P: every summary is first COPIED to <P>pmc_summary_<10|10rot|20...>.txt (a tracked file under profiles/) and that
copy is what the JSON names as its `source` (the source used to be a scratch path under gpurun_out/).

Counter handling (MI355X_MICROARCH.md "HBM", calibrated on this box with tools/ubench calib, profiles/archive/r02a_counter_calibration.txt):
  * FETCH_SIZE and WRITE_SIZE are in KiB and come from separate passes;
  * FETCH_SIZE reports exactly HALF of the bytes read, for 4-byte-per-lane and 16-byte-per-lane loads alike
    (1 GiB read -> 524 298 KiB) -> doubled;
  * WRITE_SIZE is exact for 4- and 16-byte-per-lane stores (1 GiB written -> 1 048 576 KiB) -> taken as is;
  * round 5 (profiles/r5i_counter_calibration_narrow_reads.json): the factor 2 holds for EVERY read width the step kernel uses -- 4, 8, 16
    bytes per lane, 48-byte records read by every lane or by every fourth lane; 100-byte tiles read as dwords: 1.92 -- and a
    SCATTERED dword that misses the L2 costs one whole 128-byte line at the fabric.  `fetch_attribution` below prices the
    kernel's contiguous reads (tile + 48-byte record + action + the statistics row prefetched for every bin, per bin) and
    books what is left to the three speculative pool look-aheads per bin (4 bytes each, scattered over the pool): the pool
    is a few MB, but the 140 MB a lock-step writes keep flushing it out of the 4 MB L2s, and every miss fetches a line.
VALU utilisation = SQ_ACTIVE_INST_VALU (quad-cycles, summed over all SIMDs) * 4 / (1024 SIMDs * kernel cycles),
kernel cycles = SQ_BUSY_CYCLES / 32 shader engines.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path):
    d = {}
    for line in open(path):
        p = line.split()
        if len(p) >= 4 and p[0] == "step":
            d[p[1]] = float(p[3])
    return d


def main():
    out_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        out = json.load(open(out_path))
    except Exception:
        out = {}
    out["_comment"] = ("per bpp_step launch, from rocprofv3 PMC passes (tools/profile_pmc.sh, folded by tools/pmc_to_json.py): "
                       "traffic_bytes = WRITE_SIZE KiB * 1024 + 2 * FETCH_SIZE KiB * 1024 (FETCH_SIZE reports half of the bytes "
                       "read on gfx950, WRITE_SIZE is exact; calibrated with tools/ubench calib); valu_utilisation = "
                       "SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * SQ_BUSY_CYCLES / 32)")
    args = sys.argv[1:]
    prefix = None
    if args and args[0] == "--tracked-prefix":
        prefix, args = args[1], args[2:]
    for arg in args:
        key, path = arg.split("=", 1)
        dims, rot, envs = key.split("_")
        if prefix:
            import shutil
            side = dims.split("x")[0]
            tracked = os.path.join(ROOT, "%spmc_summary_%s%s.txt" % (prefix, side, "rot" if rot == "rot1" else ""))
            if os.path.abspath(path) != os.path.abspath(tracked):
                shutil.copyfile(path, tracked)
            path = tracked
        c = parse(path)
        W, L, H = (int(v) for v in dims.split("x"))
        E = int(envs[1:])
        contiguous = E * (W * L + 48 + 8 + 32)       # byte tile, state record, action, ep_acc row (prefetched for every bin)
        fetched = int(2 * c["FETCH_SIZE"] * 1024)
        extra = max(0, fetched - contiguous)
        cyc = c["SQ_BUSY_CYCLES"] / 32.0
        util = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc)
        out[key] = {
            "traffic_bytes": int(c["WRITE_SIZE"] * 1024 + 2 * c["FETCH_SIZE"] * 1024),
            "write_bytes": int(c["WRITE_SIZE"] * 1024), "fetch_bytes_corrected": int(2 * c["FETCH_SIZE"] * 1024),
            "valu_instructions": int(c["SQ_INSTS_VALU"]), "salu_instructions": int(c["SQ_INSTS_SALU"]),
            "lds_instructions": int(c["SQ_INSTS_LDS"]), "waves": int(c["SQ_WAVES"]),
            "lds_bank_conflict_frac": c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0),
            "valu_utilisation": round(util, 3),
            "fetch_attribution": {"contiguous_reads_bytes": contiguous, "left_over_bytes": extra,
                                  "left_over_as_128B_lines": extra // 128, "speculative_pool_loads_per_launch": 3 * E,
                                  "share_of_pool_loads_that_would_miss_L2": round(extra / 128.0 / (3 * E), 3),
                                  "reading": "the left-over is what the scattered 4-byte pool look-aheads fetch when they miss the L2 (one 128-byte "
                                             "line each, calibrated); not over-fetch of the streaming tensors"},
            "source": os.path.relpath(os.path.abspath(path), ROOT),
        }
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
