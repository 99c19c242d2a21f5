"""Builds the native code in-tree (no JIT cache: the built .so travels to the GPU box).

  libqvc_hip.so      -- the product: gfx950 kernels + C ABI + host packer (hipcc --offload-arch=gfx950)
  libqvc_io.so       -- the product's host-side batch file I/O (g++; include/qvc_io.h)
  oracle/_build/libqvc_emu.so -- TEST-ONLY host emulation of the launch sequence (g++/hipcc host code)

hipcc cross-compiles gfx950 without a GPU.  Translation units compile in parallel.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "csrc", "_obj")
LIB = os.path.join(PKG, "libqvc_hip.so")
IO_LIB = os.path.join(PKG, "libqvc_io.so")
EMU_DIR = os.path.join(ROOT, "oracle", "_build")
EMU_LIB = os.path.join(EMU_DIR, "libqvc_emu.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

SOURCES = ["qvc_conv_f16.hip", "qvc_conv_bf16.hip", "qvc_wn2.hip", "qvc_chain.hip", "qvc_small.hip", "qvc_spk.hip", "qvc_mel.hip", "qvc_api.hip", "qvc_pack.cpp"]
HEADERS = ["qvc_plan.h", "qvc_kernels.h", "qvc_conv_impl.h", "qvc_wn2_impl.h", "qvc_chain_impl.h", "qvc_post_tail_impl.h", "qvc_tail_impl.h", "qvc_path.h", "qvc_stream.h", "qvc_pack_util.h", "qvc_launch_util.h"]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _run(cmd):
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), res.stdout))
    # per-kernel resource usage (registers, scratch, occupancy) is kept next to the object file:
    # tests/test_host_side.py checks that no hot kernel spills to scratch memory
    if "-c" in cmd and "-o" in cmd and cmd[cmd.index("-o") + 1].endswith(".o"):
        with open(cmd[cmd.index("-o") + 1][:-2] + ".remarks.txt", "w") as f:
            f.write(res.stdout)
    return res.stdout


def build_hip(force: bool = False, verbose: bool = False, variant: str = "") -> str:
    """variant "" = the product library; "sat" = libqvc_hip_sat.so, the same sources with -DQVC_SATCOUNT (counts
    saturating f16 conversions: the f16 dynamic-range GPU test loads it, nothing else does)."""
    extra = {"": [], "sat": ["-DQVC_SATCOUNT"]}[variant]
    obj_dir = OBJ if not variant else os.path.join(OBJ, variant)
    lib = LIB if not variant else os.path.join(PKG, f"libqvc_hip_{variant}.so")
    os.makedirs(obj_dir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(ROOT, "include", "qvc.h")]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or not _newer(o, [s] + hdrs):
            jobs.append([HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage"] + extra +
                        ["-c", s, "-o", o])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), 6)) as ex:
            for out in ex.map(_run, jobs):
                if verbose and out.strip():
                    print(out)
    if jobs or not os.path.exists(lib):
        _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


def build_io(force: bool = False) -> str:
    """libqvc_io.so: host-side batch file I/O (unit .npy in, float32 wav out) -- plain C++, no GPU code."""
    src = os.path.join(CSRC, "qvc_io.cpp")
    hdr = os.path.join(ROOT, "include", "qvc_io.h")
    if force or not _newer(IO_LIB, [src, hdr]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", IO_LIB, src])
    return IO_LIB


def build_emu(force: bool = False) -> str:
    """The oracle-side host emulation (tests only)."""
    os.makedirs(EMU_DIR, exist_ok=True)
    srcs = [os.path.join(ROOT, "oracle", "qvc_emu.cpp"), os.path.join(CSRC, "qvc_pack.cpp")]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(ROOT, "include", "qvc.h")]
    if force or not _newer(EMU_LIB, srcs + hdrs):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", EMU_LIB] + srcs)
    return EMU_LIB


if __name__ == "__main__":
    force = "--force" in sys.argv
    print(build_hip(force, verbose=True))
    print(build_hip(force, variant="sat"))
    print(build_io(force))
    print(build_emu(force))
