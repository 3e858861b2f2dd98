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


def test_fp32_translation_unit_contains_no_fp64_arithmetic():
    """qmgpu_mpc32.hip is the MPC kernel sources compiled with real = float: its arithmetic kernels must not contain a single fp64 instruction (a literal
    without the _r suffix or a forgotten `double` would promote a whole expression).  fp64 is allowed where it is meant: the two conversion kernels at
    the library boundary and mpc_init_kernel, which forms node times / steps / schedule phases from the caller's fp64 times."""
    import re
    import shutil
    import subprocess
    import tempfile
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not shutil.which(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "mpc32.s")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-DQM_REAL=float", "-Dqmk=qmk32", "--cuda-device-only", "-S",
                               os.path.join(root, "qm_door_amd", "csrc", "qmgpu_mpc32.hip"), "-o", asm], stderr=subprocess.DEVNULL)
        text = open(asm).read()
    kernels = {}
    for m in re.finditer(r"^(_ZN5qmk32\w+):.*?\n(.*?)^\.Lfunc_end\d+:", text, re.M | re.S):   # kernels and called device functions alike
        kernels[m.group(1)] = m.group(2)
    assert len(kernels) >= 9
    arithmetic = ("ad_node_kernel", "lq_node_kernel", "riccati_kernel", "linesearch_kernel", "ddp_rollout_kernel", "ddp_select_kernel", "input_weight_kernel",
                  "nodePerformance")     # the per-node cost / violation body is compiled as a function (linesearch_kernel.h)
    seen = set()
    for name, body in kernels.items():
        f64 = re.findall(r"^\s*(v_\w*f64\w*)", body, re.M)
        for k in arithmetic:
            if k in name:
                seen.add(k)
                assert not f64, (name, sorted(set(f64)))
    assert seen == set(arithmetic), seen
    assert "v_mfma_f32_16x16x4_f32" in text and "v_mfma_f64" not in text


def test_generated_device_structs_are_current():
    """kernels/problem_r.h (the fp32 build's mirrors of qmgpu_model / qmgpu_settings + the field-by-field conversion) is generated from
    include/qmgpu.h; a field added to the public header and not regenerated would silently be missing from the fp32 path."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(S.ROOT, "tools", "gen_problem_r.py"), "--check"])
    assert r.returncode == 0, "run python tools/gen_problem_r.py"


def test_inline_assembly_has_no_data_hazard_the_compiler_cannot_see():
    """tools/check_asm_hazards.py on the device assembly of both translation units (hipcc cross-compiles here): the DPP multiply-adds written as
    inline assembly (gpu_rt.h: qmFmacRowBcast) must not read a source register a VALU instruction wrote in the two preceding wait states."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(S.ROOT, "tools", "check_asm_hazards.py"), "--build"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "DPP instructions, 0 hazards" in r.stdout
