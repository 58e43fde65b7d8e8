#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- recipe that puts the UNMODIFIED reference Python the env side needs into oracle/_ref/.

    python oracle/make_ref.py            # (re)create oracle/_ref/ from /root/reference
    python oracle/make_ref.py --check    # compare oracle/_ref/ byte for byte with /root/reference (exit 1 on a difference)

Why: /root/reference exists in the build container only.  bench.py's `cpu_baseline` leg has to time the reference's
own SubprocVecEnv plumbing on the GPU box's host cores, and the `-m gpu` suite diffs the HIP path against the LIVE
reference there; both need the reference's files on that box.  oracle/_ref/ is git-ignored (reference sources never
enter this repository's history) but NOT gpurun-ignored, so -- like the in-tree .so files -- it travels with the
working tree.  `__graft_entry__.build()` runs this recipe whenever /root/reference is present.

What is copied: exactly the files Python loads when the modules below are imported and exercised (found by tracing
sys.modules, not by a hand-kept list), byte for byte, at their original relative paths, plus dataset/cut_2.pt (the
reference's CUT-2 test set, played through LoadBoxCreator by the parity suite) and the two pretrained checkpoints for
it (pretrained_models/default_cut_2.pt, rotation_cut_2.pt: the competent policies of the deep parity cases).  MANIFEST.json records the sha256 of
every file and of its source; README.txt states provenance.  Nothing under oracle/_ref/ is ever imported by the
product (tests/test_abi_cpu.py::test_product_never_imports_the_oracle covers `_ref` too).
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = "/root/reference"
DST = os.path.join(HERE, "_ref")

# entry modules: the env side (acktr/envs.py:77-118 factory, ShmemVecEnv / DummyVecEnv, Monitor, PackingGame and its
# creators, the mask helpers) and what main.py:100-207's loop needs around it (Policy, RolloutStorage, ACKTR.update)
ENTRY = ["envs.bpp0", "envs.bpp0.bin3D", "envs.bpp0.space", "envs.bpp0.binCreator", "envs.bpp0.cutCreator",
         "envs.bpp0.mdCreator", "acktr.envs", "acktr.utils", "acktr.model", "acktr.storage", "acktr.algo",
         "acktr.distributions", "baselines.bench", "baselines.bench.monitor", "baselines.common.vec_env",
         "baselines.common.vec_env.shmem_vec_env", "baselines.common.vec_env.dummy_vec_env"]
# data files: the reference's CUT-2 test set and the two checkpoints main.py:26-29 -> unified_test.py:29-67 evaluate on it
# (acktr/model_loader.py:19-35 loads them; the parity suite plays them greedily through the reference's own Policy.act)
DATA = ["dataset/cut_2.pt", "pretrained_models/default_cut_2.pt", "pretrained_models/rotation_cut_2.pt"]

TRACE = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
os.environ["BPP_REFERENCE_ROOT"] = %(src)r
from oracle import ref_shims
ref_shims.install()
import importlib
for m in %(entry)r:
    importlib.import_module(m)
# exercise the factory once: imports made inside functions (VecNormalize.__init__ -> running_mean_std, ...) show up too
import contextlib, io, tempfile, types, torch
from acktr.envs import make_vec_envs
args = types.SimpleNamespace(enable_rotation=False, container_size=(10, 10, 10), data_type="rs",
                             box_size_set=[(i, j, k) for i in range(2, 6) for j in range(2, 6) for k in range(2, 6)])
with contextlib.redirect_stdout(io.StringIO()):
    for n in (2, 1):
        envs = make_vec_envs("Bpp-v0", 1, n, 1.0, tempfile.mkdtemp(), torch.device("cpu"), False, args=args)
        envs.reset()
        envs.step(torch.zeros((n, 1), dtype=torch.long))
        envs.close()
src = os.path.realpath(%(src)r) + os.sep
files = sorted({os.path.relpath(os.path.realpath(m.__file__), src) for m in list(sys.modules.values())
                if getattr(m, "__file__", None) and os.path.realpath(m.__file__).startswith(src)})
print("TRACE=" + json.dumps(files))
"""


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def traced_files():
    out = subprocess.check_output([sys.executable, "-c", TRACE % dict(root=ROOT, src=SRC, entry=ENTRY)],
                                  stderr=subprocess.STDOUT, text=True)
    line = [ln for ln in out.splitlines() if ln.startswith("TRACE=")][-1]
    return json.loads(line[len("TRACE="):])


def make(verbose=True):
    if not os.path.isfile(os.path.join(SRC, "envs", "bpp0", "bin3D.py")):
        raise SystemExit("%s is not here: oracle/_ref/ can only be made in the build container" % SRC)
    files = traced_files() + DATA
    tmp = DST + ".tmp.%d" % os.getpid()
    shutil.rmtree(tmp, ignore_errors=True)
    manifest = {}
    for rel in files:
        s, d = os.path.join(SRC, rel), os.path.join(tmp, rel)
        os.makedirs(os.path.dirname(d), exist_ok=True)
        shutil.copyfile(s, d)
        os.chmod(d, 0o644)
        manifest[rel] = {"sha256": sha(d), "bytes": os.path.getsize(d)}
    with open(os.path.join(tmp, "MANIFEST.json"), "w") as f:
        json.dump({"source": SRC, "recipe": "oracle/make_ref.py", "entry_modules": ENTRY, "files": manifest}, f, indent=1, sort_keys=True)
    with open(os.path.join(tmp, "README.txt"), "w") as f:
        f.write("Byte-for-byte copies of files of alexfrom0815/Online-3D-BPP-DRL (the upstream reference), made by\n"
                "oracle/make_ref.py from %s.  TEST INFRASTRUCTURE: timed by bench.py's cpu_baseline leg and\n"
                "diffed against the HIP path by the GPU test-suite on a box where /root/reference does not exist.\n"
                "Git-ignored on purpose; never imported by the product.  sha256 of every file: MANIFEST.json.\n" % SRC)
    shutil.rmtree(DST, ignore_errors=True)
    os.replace(tmp, DST)
    if verbose:
        print("oracle/_ref: %d files, %d bytes" % (len(manifest), sum(v["bytes"] for v in manifest.values())))
    return manifest


def check():
    """oracle/_ref/ == /root/reference for every file in the manifest, and the manifest covers what a fresh trace finds."""
    man = json.load(open(os.path.join(DST, "MANIFEST.json")))["files"]
    bad = [rel for rel, ent in man.items() if sha(os.path.join(DST, rel)) != ent["sha256"]]
    if os.path.isdir(SRC):
        bad += [rel for rel, ent in man.items() if sha(os.path.join(SRC, rel)) != ent["sha256"]]
        bad += [rel for rel in traced_files() + DATA if rel not in man]
    return sorted(set(bad))


def fresh():
    """True when oracle/_ref/ exists and matches its manifest (and the source tree, where that is present)."""
    try:
        return not check()
    except Exception:
        return False


if __name__ == "__main__":
    if "--check" in sys.argv:
        bad = check()
        print("oracle/_ref: %s" % ("ok" if not bad else "DIFFERS: %s" % bad))
        sys.exit(1 if bad else 0)
    make()
