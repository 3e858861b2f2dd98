"""Builds qm_door_amd/libqmgpu.so in-tree: hipcc (gfx950) for the kernels + C ABI, g++ for the host loaders.

The HIP runtime the library binds to is the one the host process already uses: under the Python harness that is the
libamdhip64 bundled with PyTorch-ROCm (two HIP/HSA runtimes in one process cannot both see the GPU); a C++ host such as
the qm_controllers plugin links the same objects against the system ROCm instead.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OUT = os.path.join(PKG, "libqmgpu.so")
OBJ = os.path.join(PKG, "build")
ARCH = "gfx950"


def _torch_lib_dir():
    try:
        import torch
        d = os.path.join(os.path.dirname(torch.__file__), "lib")
        if os.path.exists(os.path.join(d, "libamdhip64.so")):
            return d
    except Exception:
        pass
    return None


def _sources():
    deps = [os.path.join(PKG, "..", "include", "qmgpu.h")]
    for root, _, files in os.walk(CSRC):
        deps += [os.path.join(root, f) for f in files]
    return deps


def build_library(force=False, verbose=False, extra_flags=(), out=None, obj_dir=None):
    OUT, OBJ = out or globals()["OUT"], obj_dir or globals()["OBJ"]
    deps = _sources() + [os.path.abspath(__file__)]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    api_o, mpc32_o, host_o, ls_o, lq_o = (os.path.join(OBJ, n) for n in ("qmgpu_api.o", "qmgpu_mpc32.o", "host_config.o", "qmgpu_ls.o", "qmgpu_lq.o"))
    # -enable-ipra=0: LLVM's interprocedural register allocation (on by default for AMDGPU) miscompiles a call on wbc_kernel's helper wavefront path in
    # some build variants of these sources (reproducer tools/wbc_variants.py --run opq x_noipra); with it off every variant
    # computes the same cycle.  a variant switches it back on with extra_flags = (-mllvm, -enable-ipra=1).
    ipra = [] if any(str(f).startswith("-enable-ipra") for f in extra_flags) else ["-mllvm", "-enable-ipra=0"]
    hip_flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", *ipra, *extra_flags]
    # the kernel sources are written in terms of `real` (kernels/real.h) and compiled twice: fp64 = every kernel + the C ABI,
    # fp32 = the MPC kernels a second time in namespace qmk32
    # linesearch_kernel lives in a translation unit of its own, compiled with the interprocedural register allocation ON (qmgpu_ls.hip says why); a variant that sets the
    # switch itself, or the profiling build (one device symbol for all clocks), keeps the single translation unit
    split_ls = bool(ipra) and "-DQM_RICCATI_TIMING" not in extra_flags
    base_flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", *extra_flags]
    # lq_node_kernel lives in a translation unit of its own (qmgpu_lq.hip) compiled at -O2: measured 2.7 % faster than at -O3 (0.456 -> 0.443 ms, profiles/r04i_variant_timing.txt),
    # bit-identical results; the profiling build keeps the single translation unit
    split_lq = "-DQM_RICCATI_TIMING" not in extra_flags
    cmds = [
        [hipcc, *hip_flags, *(["-DQM_LS_EXTERN"] if split_ls else []), *(["-DQM_LQ_EXTERN"] if split_lq else []), "-c", os.path.join(CSRC, "qmgpu_api.hip"), "-o", api_o],
        [hipcc, *hip_flags, "-DQM_REAL=float", "-Dqmk=qmk32", "-c", os.path.join(CSRC, "qmgpu_mpc32.hip"), "-o", mpc32_o],
        ["g++", "-O2", "-std=c++17", "-fPIC", "-c", os.path.join(CSRC, "host", "host_config.cpp"), "-o", host_o],
    ]
    if split_ls:
        cmds.append([hipcc, *base_flags, "-mllvm", "-enable-ipra=1", "-c", os.path.join(CSRC, "qmgpu_ls.hip"), "-o", ls_o])
    if split_lq:
        cmds.append([hipcc, *[("-O2" if f == "-O3" else f) for f in hip_flags], "-c", os.path.join(CSRC, "qmgpu_lq.hip"), "-o", lq_o])
    # Inside this repository the process already holds PyTorch's bundled HIP runtime, so link against that one first; a catkin
    # workspace without PyTorch sets QMGPU_HIP_LIBDIR=/opt/rocm/lib.
    override = os.environ.get("QMGPU_HIP_LIBDIR")
    tl = None if override else _torch_lib_dir()
    libdirs = [override] if override else (([tl] if tl else []) + ["/opt/rocm/lib"])
    link = ["g++", "-shared", "-o", OUT, api_o, mpc32_o, host_o] + ([ls_o] if split_ls else []) + ([lq_o] if split_lq else [])
    for d in libdirs:
        link += [f"-L{d}", f"-Wl,-rpath,{d}"]
    link += ["-lamdhip64", "-lstdc++", "-lm"]
    cmds.append(link)
    procs = []
    for c in cmds[:-1]:   # the compilations are independent
        if verbose:
            print(" ".join(c), file=sys.stderr)
        procs.append(subprocess.Popen(c))
    for c, pr in zip(cmds, procs):
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, c)
    if verbose:
        print(" ".join(cmds[-1]), file=sys.stderr)
    subprocess.check_call(cmds[-1])
    return OUT


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
