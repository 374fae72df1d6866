"""Builds the in-tree native libraries: libvr_hip.so (HIP kernels + C ABI, gfx950) and libvr_host.so
(C++ host surface mirroring the reference's Volume / TransferFunction / Camera / MiniApp classes)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# (xnack-: the MI355X runs with XNACK off; code for "either" makes the compiler break every run of vector-memory instructions in
# which a destination overlaps an earlier address register with an s_nop -- seven per request of march_p2_kernel.
# VR_HIP_ARCH=gfx950 in the environment builds the generic form.)
HIP_ARCH = os.environ.get("VR_HIP_ARCH", "gfx950:xnack-")
HIP_FLAGS = ["-O3", "--offload-arch=" + HIP_ARCH, "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared",
             "-std=c++17", "-Wall", "-Wno-unused-function"]
HOST_FLAGS = ["-O2", "-std=c++20", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall"]


def _newer(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _sources(d: str, exts: tuple[str, ...]) -> list[str]:
    out = []
    for dp, _, fs in os.walk(d):
        out += [os.path.join(dp, f) for f in fs if f.endswith(exts)]
    return sorted(out)


def build_hip(force: bool = False) -> str:
    """Two translation units, compiled side by side: vr_api.hip (C ABI + kernels with separately rounded multiply-adds)
    and vr_fused.hip (the march kernels once more with fused multiply-adds), linked into one libvr_hip.so."""
    target = os.path.join(HERE, "libvr_hip.so")
    hdrs = [os.path.join(CSRC, f) for f in ("vr_kernels.h", "vr_wtb.h", "vr_dp.h", "vr_pw.h", "vr_p2.h", "vr_mixed.h", "vr_lt.h", "vr_device.h", "vr_launch.h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "vr.h"))
    flags = [f for f in HIP_FLAGS if f != "-shared"] + os.environ.get("VR_EXTRA_HIPCC_FLAGS", "").split()
    if os.environ.get("VR_EXPERIMENTAL_FLAVOURS", "0") not in ("", "0"):
        flags.append("-DVR_EXPERIMENTAL_FLAVOURS=1")  # the kernel forms that lost every A/B (vr_launch.h)
    # the flags the objects were compiled with: a change (VR_EXTRA_HIPCC_FLAGS, VR_EXPERIMENTAL_FLAVOURS) rebuilds them
    stamp = os.path.join(CSRC, ".build_flags")
    want = " ".join(flags)
    if not os.path.exists(stamp) or open(stamp).read() != want:
        force = True
    objs, procs = [], []
    for name in ("vr_api", "vr_fused"):
        src, obj = os.path.join(CSRC, name + ".hip"), os.path.join(CSRC, name + ".o")
        objs.append(obj)
        if force or _newer(obj, hdrs + [src]):
            procs.append((name, subprocess.Popen([HIPCC, *flags, "-c", "-o", obj, src])))
    for name, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, f"hipcc {name}.hip")
    if procs or force or _newer(target, objs):
        subprocess.run([HIPCC, "--offload-arch=" + HIP_ARCH, "-fPIC", "-shared", "-o", target, *objs], check=True)
    with open(stamp, "w") as fh:
        fh.write(want)
    return target


def build_host(force: bool = False) -> str | None:
    hostdir = os.path.join(CSRC, "host")
    cpps = _sources(hostdir, (".cpp",))
    if not cpps:
        return None
    target = os.path.join(HERE, "libvr_host.so")
    deps = _sources(hostdir, (".cpp", ".h")) + [os.path.join(os.path.dirname(HERE), "include", "vr.h")]
    if force or _newer(target, deps):
        subprocess.run(["g++", *HOST_FLAGS, "-I", os.path.join(os.path.dirname(HERE), "include"), "-o", target, *cpps,
                        "-L", HERE, "-lvr_hip", "-Wl,-rpath,$ORIGIN"], check=True)
    return target


def build_mgpu(force: bool = False) -> str:
    """libvr_mgpu.so: the multi-GPU frame loop (C++ host code on the HIP runtime + RCCL, above the C ABI of libvr_hip.so)."""
    target = os.path.join(HERE, "libvr_mgpu.so")
    src = os.path.join(CSRC, "mgpu", "vr_mgpu.cpp")
    inc = os.path.join(os.path.dirname(HERE), "include")
    deps = [src, os.path.join(inc, "vr_mgpu.h"), os.path.join(inc, "vr.h")]
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    if force or _newer(target, deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", inc,
                        "-I", os.path.join(rocm, "include"), "-o", target, src, "-L", HERE, "-lvr_hip",
                        "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-lrccl",
                        "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(rocm, "lib")], check=True)
    return target


def build_all(force: bool = False):
    build_hip(force)
    build_host(force)
    build_mgpu(force)
