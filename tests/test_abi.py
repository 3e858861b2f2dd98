"""The C-ABI shared library loads, exports every symbol include/qmgpu.h declares, agrees with ctypes on struct layout,
and refuses to compute without a GPU (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

import support as S
from qm_door_amd import abi


def test_header_symbols_exported(hip_lib):
    header = open(os.path.join(S.ROOT, "include", "qmgpu.h")).read()
    declared = sorted(set(re.findall(r"\b(qmgpu_[a-z_0-9]+)\s*\(", header)))
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(hip_lib, name), f"{name} declared in qmgpu.h but not exported"
    assert sorted(abi.SYMBOLS) == declared


def test_struct_layout_matches_c():
    src = r'''
    #include <stdio.h>
    #include "qmgpu.h"
    int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(qmgpu_model), sizeof(qmgpu_settings), sizeof(qmgpu_problem), sizeof(qmgpu_gait),
                            sizeof(qmgpu_mpc_args), sizeof(qmgpu_wbc_args), sizeof(qmgpu_frontend_args)); return 0; }
    '''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(S.ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        sizes = [int(v) for v in subprocess.check_output([os.path.join(d, "s")]).split()]
    assert sizes == [C.sizeof(abi.Model), C.sizeof(abi.Settings), C.sizeof(abi.Problem), C.sizeof(abi.Gait), C.sizeof(abi.MpcArgs), C.sizeof(abi.WbcArgs), C.sizeof(abi.FrontendArgs)]


def test_no_cpu_fallback(hip_lib, interface):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    st = hip_lib.qmgpu_create(C.byref(interface.problem), 0, 4, 8, C.byref(h))
    assert st == abi.ERR_NO_DEVICE and not h.value
    assert b"no" in hip_lib.qmgpu_strerror(st).lower()


def test_product_never_links_oracle():
    """The shipped library must not reference the oracle (test infrastructure)."""
    out = subprocess.check_output(["readelf", "-d", abi.LIB_PATH]).decode()
    assert "oracle" not in out
    for root, _, files in os.walk(os.path.join(S.ROOT, "qm_door_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                assert "libqm_oracle" not in txt and "qmo_" not in txt, f
