"""TEST INFRASTRUCTURE ONLY -- builds and binds tests/emu/libbpp_emu.so: the PRODUCT kernel source
(online-3d-bpp-drl_amd/csrc/bpp_kernels.hip, unmodified) compiled by g++ against the cooperative-fiber SIMT
emulator (tests/emu/emu_runtime.cpp, tests/emu/hip/hip_runtime.h).  Same C ABI as libbpp_hip.so with host
pointers, so the numpy front-end written for the oracle library drives it unchanged: this module is a second
instance of oracle/oracle.py bound to the emulated library.  Used by the CPU test suite to check kernel
LOGIC against the oracle without a GPU; it is never imported by the product and measures nothing."""
import importlib.util
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, "online-3d-bpp-drl_amd", "csrc", "bpp_kernels.hip")
DEPS = [SRC, os.path.join(ROOT, "online-3d-bpp-drl_amd", "csrc", "bpp_tile_kernel.inl"), os.path.join(ROOT, "online-3d-bpp-drl_amd", "csrc", "bpp_tile_body.inl"),
        os.path.join(ROOT, "online-3d-bpp-drl_amd", "csrc", "bpp_stream_gen.inl"), os.path.join(ROOT, "online-3d-bpp-drl_amd", "csrc", "bpp_heads.inl"), os.path.join(ROOT, "online-3d-bpp-drl_amd", "csrc", "bpp_rt_kernels.inl"),
        os.path.join(ROOT, "online-3d-bpp-drl_amd", "csrc", "bpp_stats.inl"), os.path.join(HERE, "emu_runtime.cpp"), os.path.join(HERE, "hip", "hip_runtime.h"),
        os.path.join(ROOT, "include", "bpp_abi.h"), os.path.join(ROOT, "include", "bpp_gen.inl")]
LIB = os.path.join(HERE, "libbpp_emu.so" if not os.environ.get("BPP_EMU_DEFINES") else
                   "libbpp_emu.so." + "".join(c if c.isalnum() else "_" for c in os.environ["BPP_EMU_DEFINES"]))


def build(force=False):
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in DEPS):
        return LIB
    tmp = LIB + ".tmp.%d" % os.getpid()
    # BPP_EMU_DEFINES="-DX -DY": emulate an A/B build of the product source (tools/build_variant.sh's -D switches); the library is
    # then rebuilt on every change of that variable because it is a file of its own
    extra = os.environ.get("BPP_EMU_DEFINES", "").split()
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-w", "-fPIC", "-shared", "-I", HERE] + extra +
                          ["-x", "c++", os.path.join(HERE, "emu_runtime.cpp"), "-o", tmp, "-lpthread"])
    os.replace(tmp, LIB)
    return LIB


def load():
    """A private copy of the oracle's numpy front-end (OracleEnv, mask_from_obs, sample_feasible, ...) whose
    library handle is the emulated product."""
    build()
    spec = importlib.util.spec_from_file_location("bpp_emu_frontend", os.path.join(ROOT, "oracle", "oracle.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.LIB = LIB
    m.build = build
    m._lib = None
    m.lib()
    m.EmuEnv = m.OracleEnv
    return m
