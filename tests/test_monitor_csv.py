"""`log_dir` of make_vec_envs (acktr/envs.py:54-58): vec_env.MonitorCsv writes `<rank>.monitor.csv` in the format of the
reference's own ResultsWriter (baselines/bench/monitor.py:99-119) -- checked here against that class itself and read back
with the reference's `load_results`.  CPU only (the writer is host code); the GPU suite checks that step() feeds it."""
import json
import os

import numpy as np
import pytest

from oracle import ref_shims

pytestmark = pytest.mark.skipif(not (ref_shims.available() or ref_shims.copy_available()), reason="no reference tree / oracle/_ref copy")


def _episodes(rng, n, E):
    bins = np.sort(rng.choice(E, size=n, replace=False)).astype("<i4")
    r = np.round(rng.randint(1, 1000, size=n) / 1000.0 * 10.0 + rng.rand(n) * 1e-7, 6)
    return {"bins": bins, "r": r, "l": rng.randint(1, 40, size=n).astype("<i4"), "ratio": r / 10.0, "counter": rng.randint(0, 39, size=n).astype("<i4")}


def test_monitor_csv_matches_the_reference_writer_and_loads_with_load_results(tmp_path):
    ref_shims.install()
    from baselines.bench import monitor as ref_monitor
    import bpp_amd
    from bpp_amd.vec_env import MonitorCsv
    rng = np.random.RandomState(3)
    ours_dir, ref_dir = str(tmp_path / "ours"), str(tmp_path / "ref")
    os.makedirs(ref_dir)
    t0 = 1234.5
    mon = MonitorCsv(ours_dir, rank=3, env_id="Bpp-v0", env_id_base=3 * 64, t_start=t0)
    ref = ref_monitor.ResultsWriter(os.path.join(ref_dir, "3"), header={"t_start": t0, "env_id": "Bpp-v0"}, extra_keys=("bin",))
    assert os.path.basename(mon.path) == "3." + ref_monitor.Monitor.EXT and os.path.basename(ref.f.name) == "3." + ref_monitor.Monitor.EXT
    total = 0
    for step in range(5):
        eps = _episodes(rng, int(rng.randint(0, 9)), 64)
        t_now = t0 + 0.25 * (step + 1) + 1e-7
        mon.write(eps, t_now)
        for b, r, l in zip(eps["bins"], eps["r"], eps["l"]):      # what Monitor.update hands its writer (monitor.py:62-72)
            ref.write_row({"r": round(float(r), 6), "l": int(l), "t": round(t_now - t0, 6), "bin": 3 * 64 + int(b)})
        total += len(eps["bins"])
    mon.close()
    ref.f.close()
    assert mon.rows == total > 0
    assert open(mon.path, newline="").read() == open(ref.f.name, newline="").read()      # byte for byte, header line included
    first = open(mon.path).readline()
    assert first[0] == "#" and json.loads(first[1:]) == {"t_start": t0, "env_id": "Bpp-v0"}
    df = ref_monitor.load_results(ours_dir)            # the reference's reader (pandas)
    assert len(df) == total and list(df.columns[-4:]) == ["r", "l", "t", "bin"] and (df["bin"] >= 3 * 64).all()
