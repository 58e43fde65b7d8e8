"""CPU-side checks of the product boundary: the HIP shared library loads without a GPU, exports every
symbol include/bpp_abi.h declares, validates arguments before touching the device, and the ctypes
struct layouts match the header.  No compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest

import bpp_amd
from bpp_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.lib()


def header_functions():
    src = open(os.path.join(ROOT, "include", "bpp_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bpp_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = header_functions()
    assert sorted(names) == sorted(_lib.SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n
    from oracle import oracle as orc
    orc.build()
    for n in names:
        assert hasattr(orc.lib(), n), n


def test_abi_version_and_limits(lib):
    assert lib.bpp_abi_version() == 16
    assert _lib.limits() == (1024, 255)


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_lib.Batch) == 8 * 4 + 2 * 8 + 4 * 8 + 8 + 8 and _lib.Batch.seq_cache.offset == 88
    assert _lib.Batch.env_id_base.offset == 32 and _lib.Batch.seq_pool.offset == 48
    assert ctypes.sizeof(_lib.StepOut) == 64 + 24 + 16
    from oracle import oracle as orc
    assert ctypes.sizeof(orc.Batch) == ctypes.sizeof(_lib.Batch) and orc.STATE_DTYPE.itemsize == 48


def test_argument_validation_happens_before_any_device_work(lib):
    b = _lib.Batch(16, 10, 10, 10, 0, 0, 4, 8, 0, 16, None, None, None, None)
    o = _lib.StepOut()
    assert lib.bpp_reset(ctypes.byref(b), 0, ctypes.byref(o), None) == -1
    assert b"NULL" in lib.bpp_last_error()
    b2 = _lib.Batch(16, 64, 64, 10, 0, 0, 4, 8, 0, 16, 16, 16, 16, None)
    assert lib.bpp_step(ctypes.byref(b2), 16, ctypes.byref(o), None) == -2   # W*L > 1024
    assert b"too large" in lib.bpp_last_error()
    assert lib.bpp_mask_from_obs(None, None, 1, 10, 10, 10, 0, 0, None) == -1
    assert lib.bpp_mask_from_obs(16, 16, 1, 10, 10, 10, 0, 7, None) == -1       # unknown rule
    assert lib.bpp_mask_from_hmap(16, 16, 16, 0, 10, 10, 10, 0, 0, None) == -1  # E <= 0
    assert lib.bpp_sample_feasible(None, None, 1, 1, 0, 0, 0, None) == -1
    assert lib.bpp_reset(ctypes.byref(b2), 5, ctypes.byref(o), None) == -1      # bad mode
    o2 = _lib.StepOut(16)
    b3 = _lib.Batch(16, 10, 10, 10, 0, 0, 4, 8, 0, 16, 16, 16, 16, None, _lib.POOL_STATIC, 0, 128)
    assert lib.bpp_reset(ctypes.byref(b3), 0, ctypes.byref(o2), None) == -1 and b"seq_cache" in lib.bpp_last_error()   # a row cache needs a ring
    b4 = _lib.Batch(16, 10, 10, 10, 0, 0, 64, 8, 0, 16, 16, 16, 16, None, _lib.POOL_RING, 0, 128)
    assert lib.bpp_reset(ctypes.byref(b4), 0, ctypes.byref(o2), None) == -1 and b"depth >= 5" in lib.bpp_last_error()


def test_no_silent_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="HIP device"):
        bpp_amd.BppVecEnv(4, (10, 10, 10), pool=bpp_amd.sequences.cut2_pool((10, 10, 10), 2))
    with pytest.raises(RuntimeError, match="HIP device"):
        bpp_amd.batched_mask_from_obs(np.zeros((1, 400), np.float32), (10, 10, 10))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "online-3d-bpp-drl_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libbpp_oracle" not in txt, f
                # ... nor the reference copies under oracle/_ref/ or the import shims that load them
                assert "ref_shims" not in txt and "oracle/_ref" not in txt and "BPP_REFERENCE_ROOT" not in txt, f


def test_bench_touches_the_reference_only_inside_the_cpu_baseline_leg():
    """bench.py may execute oracle/_ref/ only as the reported CPU baseline and oracle/ only there and as the CHECKER of its
    in-run parity gate (parity_gate, outside every timed region); /root/reference never."""
    txt = open(os.path.join(ROOT, "bench.py")).read()
    assert "/root/reference" not in txt.replace("/root/reference is never read", "")
    body = txt[txt.index("def main():"):]
    assert "ref_shims" not in body and "ref_baseline" not in body and "from oracle" not in body


def test_mark_and_wait_mark_on_the_oracle_library(oracle):
    """bpp_mark / bpp_wait_mark (ABI v15): the completion word a step leaves behind in host memory; on the CPU libraries
    everything enqueued is complete at once."""
    import ctypes
    lib = oracle.lib()
    lib.bpp_mark.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    lib.bpp_wait_mark.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    word = np.zeros(2, dtype=np.uint32)
    assert lib.bpp_wait_mark(word.ctypes.data, 7, None) != 0         # nothing marked yet
    assert lib.bpp_mark(word.ctypes.data, 7, None) == 0 and word[0] == 7 and word[1] == 0
    assert lib.bpp_wait_mark(word.ctypes.data, 7, None) == 0
    assert lib.bpp_mark(None, 1, None) != 0 and lib.bpp_wait_mark(None, 1, None) != 0
