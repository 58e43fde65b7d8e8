#!/usr/bin/env python3
"""Item sequences produced by the UNMODIFIED reference creators under fixed seeds -> tests/golden/sequences_reference.npz.

Run in the build container only (needs /root/reference):   python tests/golden/make_sequences_golden.py

  * CUT-2: envs.bpp0.mdCreator.MDlayerBoxCreator(size, [lo, hi]) after random.seed(s); reset()
           (envs/bpp0/mdCreator.py:147-166; the stored trailing [10,10,10] is dropped)
  * CUT-1: envs.bpp0.cutCreator.CuttingBoxCreator(size, box_range, rotation) after random.seed(s);
           np.random.seed(s); reset(); generate_box_size() until the candidates run out
           (envs/bpp0/cutCreator.py:32-128)
  * RS:    envs.bpp0.binCreator.RandomBoxCreator(box_set) after np.random.seed(s): the first n items
           (envs/bpp0/binCreator.py:24-40)
Nothing stored here is computed by this repository's code.  Stored per case: `<name>_pool` uint8 [n][T][3]
(zero padded), `<name>_len` int32 [n], `<name>_meta` (size, bounds/range, rotation, first seed)."""
import contextlib
import io
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_shims  # noqa: E402

ref_shims.install()

from envs.bpp0.binCreator import RandomBoxCreator  # noqa: E402
from envs.bpp0.cutCreator import CuttingBoxCreator  # noqa: E402
from envs.bpp0.mdCreator import MDlayerBoxCreator  # noqa: E402


def pack(seqs):
    T = max(len(s) for s in seqs)
    pool = np.zeros((len(seqs), T, 3), np.uint8)
    for k, s in enumerate(seqs):
        pool[k, :len(s)] = np.asarray(s, np.int64).reshape(-1, 3)
    return pool, np.array([len(s) for s in seqs], np.int32)


def cut2(size, bound, seed0, n):
    out = []
    with contextlib.redirect_stdout(io.StringIO()):
        cr = MDlayerBoxCreator(size, list(bound))
        for k in range(n):
            random.seed(seed0 + k)
            cr.reset()
            out.append([tuple(b) for b in cr.box_set[:-1]])
    return out


def cut1(size, box_range, rotation, seed0, n):
    out = []
    for k in range(n):
        random.seed(seed0 + k)
        np.random.seed(seed0 + k)
        cr = CuttingBoxCreator(size, list(box_range), rotation)   # __init__ cuts once ...
        random.seed(seed0 + k)
        np.random.seed(seed0 + k)
        cr.reset()                                                 # ... reset() cuts again from the same stream
        seq = []
        while True:
            cr.generate_box_size()
            if tuple(cr.box_list[-1]) == tuple(size) and len(cr.candidates) == 0:
                break
            seq.append(tuple(int(v) for v in cr.box_list[-1]))
        out.append(seq)
    return out


def rs(box_set, seed0, n, length):
    out = []
    with contextlib.redirect_stdout(io.StringIO()):
        for k in range(n):
            np.random.seed(seed0 + k)
            cr = RandomBoxCreator(box_set)
            cr.reset()
            out.append([tuple(int(v) for v in b) for b in cr.preview(length)])
    return out


def main():
    rec = {}

    def put(name, seqs, meta):
        rec[name + "_pool"], rec[name + "_len"] = pack(seqs)
        rec[name + "_meta"] = np.array(meta, np.int64)
        print("%-18s n=%d  lengths %d..%d" % (name, len(seqs), min(map(len, seqs)), max(map(len, seqs))))

    put("cut2_10", cut2((10, 10, 10), (2, 5), 1000, 24), [10, 10, 10, 2, 5, 1000])
    put("cut2_20", cut2((20, 20, 20), (2, 5), 100, 4), [20, 20, 20, 2, 5, 100])
    put("cut2_20x20x10", cut2((20, 20, 10), (2, 5), 50, 6), [20, 20, 10, 2, 5, 50])
    put("cut2_12_b36", cut2((12, 12, 12), (3, 6), 7, 6), [12, 12, 12, 3, 6, 7])
    put("cut1_10", cut1((10, 10, 10), (2, 2, 2, 5, 5, 5), False, 77, 16), [10, 10, 10, 2, 2, 2, 5, 5, 5, 0, 77])
    put("cut1_10_rot", cut1((10, 10, 10), (2, 2, 2, 5, 5, 5), True, 77, 12), [10, 10, 10, 2, 2, 2, 5, 5, 5, 1, 77])
    put("cut1_20", cut1((20, 20, 20), (2, 2, 2, 5, 5, 5), False, 5, 2), [20, 20, 20, 2, 2, 2, 5, 5, 5, 0, 5])
    put("cut1_8x12x9_rot", cut1((8, 12, 9), (1, 2, 1, 4, 6, 3), True, 31, 6), [8, 12, 9, 1, 2, 1, 4, 6, 3, 1, 31])
    default = None                                                  # RandomBoxCreator.default_box_set: {2..6}^3
    put("rs_default", rs(default, 11, 6, 64), [0, 11])
    arg_set = [(i, j, k) for i in range(2, 6) for j in range(2, 6) for k in range(2, 6)]   # acktr/arguments.py:122-128
    put("rs_args", rs(arg_set, 3, 6, 64), [1, 3])
    path = os.path.join(HERE, "sequences_reference.npz")
    np.savez_compressed(path, **rec)
    print("->", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
