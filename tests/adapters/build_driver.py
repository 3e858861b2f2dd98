"""Builds tests/adapters/build/adapter_driver: the reference-side adapters (qm_door_amd/adapters/*.h) compiled against the OCS2 / ROS
type stand-ins of tests/adapters/mock and linked with libqmgpu.so.  Test infrastructure; the binary travels to the GPU box."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "build", "adapter_driver")
INCLUDES = ["-I/opt/rocm/include", f"-I{os.path.join(HERE, 'mock')}", f"-I{os.path.join(ROOT, 'include')}", f"-I{os.path.join(ROOT, 'qm_door_amd', 'adapters')}"]
FLAGS = ["-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__"]


def syntax_check(header):
    """g++ -fsyntax-only of one adapter header against the stand-ins; returns the compiler's stderr ('' = clean)."""
    src = f'#include "{header}"\n'
    r = subprocess.run(["g++", *FLAGS, *INCLUDES, "-fsyntax-only", "-x", "c++", "-"], input=src, capture_output=True, text=True)
    return r.stderr if r.returncode else ""


def build_driver(force=False):
    from qm_door_amd import build as qb
    lib = qb.build_library()
    deps = [os.path.join(HERE, "adapter_driver.cpp"), lib] + [os.path.join(ROOT, "qm_door_amd", "adapters", f) for f in ("GpuMpc.h", "GpuWbc.h", "QMGpuController.h")]
    for root, _, files in os.walk(os.path.join(HERE, "mock")):
        deps += [os.path.join(root, f) for f in files]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    libdirs = [os.path.dirname(lib)] + ([qb._torch_lib_dir()] if qb._torch_lib_dir() else []) + ["/opt/rocm/lib"]
    cmd = ["g++", *FLAGS, *INCLUDES, os.path.join(HERE, "adapter_driver.cpp"), "-o", OUT]
    for d in libdirs:
        cmd += [f"-L{d}", f"-Wl,-rpath,{d}"]
    cmd += ["-lqmgpu", "-lamdhip64"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build_driver(force=True))
