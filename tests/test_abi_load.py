"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/vr.h declares,
the Python struct matches the C struct, and the product fails loudly (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re
import subprocess

import pytest

from volumerendering_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    header = open(os.path.join(ROOT, "include", "vr.h")).read()
    declared = set(re.findall(r"\b(vr_[a-z_0-9]+)\s*\(", header))
    declared -= {"vr_ctx", "vr_status", "vr_variant", "vr_uniforms"}
    assert declared == set(capi.ABI_SYMBOLS), declared ^ set(capi.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.vr_abi_version() == 1


def test_multi_gpu_library_exports_every_declared_symbol():
    """libvr_mgpu.so (the C++ multi-GPU frame loop: RCCL gather) loads without a GPU and exports what vr_mgpu.h declares;
    ncclGather is resolved against RCCL."""
    from volumerendering_amd import mgpu
    lib = mgpu.load()
    header = open(os.path.join(ROOT, "include", "vr_mgpu.h")).read()
    declared = set(re.findall(r"\b(vr_mgpu_[a-z_0-9]+)\s*\(", header))
    assert declared == set(mgpu.ABI_SYMBOLS), declared ^ set(mgpu.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    und = subprocess.run(["nm", "-D", "--undefined-only", mgpu.LIB_PATH], capture_output=True, text=True).stdout
    for sym in ("ncclGather", "ncclCommInitRank", "ncclCommInitAll", "ncclGroupStart", "vr_render_tiles_async", "vr_render_tiles_batch_async", "vr_unpack_tiles_strided_async"):
        assert sym in und, sym
    assert "rccl" in subprocess.run(["ldd", mgpu.LIB_PATH], capture_output=True, text=True).stdout


def test_uniform_struct_layout_matches_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "vr.h"\nint main(){printf("%zu %zu %zu %zu %zu",'
                   'sizeof(vr_uniforms), offsetof(vr_uniforms, camera_pos), offsetof(vr_uniforms, step_size),'
                   'offsetof(vr_uniforms, toggles), offsetof(vr_uniforms, light_pos));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    U = capi.Uniforms
    assert [int(x) for x in out] == [C.sizeof(U), U.camera_pos.offset, U.step_size.offset, U.toggles.offset,
                                     U.light_pos.offset]


def test_no_silent_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.VrError) as ei:
        capi.Context(64, 64)
    assert ei.value.code == capi.VR_ERR_HIP
    assert "no CPU fallback" in str(ei.value)


def test_product_never_references_the_oracle():
    """The shipped package must not import, load or link anything under oracle/."""
    pkg = os.path.join(ROOT, "volumerendering_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "vr_oracle" not in txt and "oracle_binding" not in txt and "host_ref" not in txt, f
    for so in ("libvr_hip.so", "libvr_host.so", "libvr_mgpu.so"):
        p = os.path.join(pkg, so)
        if os.path.exists(p):
            out = subprocess.run(["ldd", p], capture_output=True, text=True).stdout
            assert "oracle" not in out
