"""Repack the reference's stored trajectory sets as pool fixtures (uint8 [n][T][4] `pool`, file order, every row ending in
the stored terminator) so that they travel to the GPU box: the files themselves (dataset/*.pt, torch-pickled python lists)
stay in /root/reference.  Run here (build container); commits tests/golden/*.npz.

    python tests/golden/make_dataset_fixtures.py

  cut2_dataset_4bins_20x20x10.npz  = dataset/4bins_cut_2.pt: 2 100 CUT-2 trajectories for the 20x20x10 bin of multi_bin/
                                     (60-167 entries each incl. the stored [20,20,10]; items 2..5 per side, sum of volumes 4000)
(cut2_dataset_10.npz = dataset/cut_2.pt is written by make_golden.py: dataset_fixture.)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = os.environ.get("BPP_REFERENCE_ROOT", "/root/reference")


def main():
    import bpp_amd
    d = torch.load(os.path.join(REF, "dataset", "4bins_cut_2.pt"), weights_only=True)
    term = (20, 20, 10)
    seqs = []
    for s in d:
        assert tuple(s[-1]) == term and sum(x * y * z for x, y, z in s[:-1]) == 20 * 20 * 10
        seqs.append([tuple(int(v) for v in it) for it in s[:-1]])
    pool = bpp_amd.sequences.pad_pool(seqs, term)
    path = os.path.join(HERE, "cut2_dataset_4bins_20x20x10.npz")
    np.savez_compressed(path, pool=pool)
    print("cut2_dataset_4bins_20x20x10  P=%d T=%d  %d KB" % (pool.shape[0], pool.shape[1], os.path.getsize(path) // 1024))


if __name__ == "__main__":
    main()
