"""Static check of the data hazards the compiler cannot see: instructions written as inline assembly.

The hazard recogniser of the AMDGPU backend pads / reorders around the instructions IT selects; the text of an `asm volatile` is opaque to it.
gpu_rt.h: qmFmacRowBcast emits `v_fmac_f64_dpp acc, bc, m row_newbcast:R`; on gfx9xx a DPP source operand (src0 = bc) must not have been written by a
VALU instruction in the two preceding wait states (and EXEC not by a VALU in the five preceding ones), and the hardware does not interlock.  The
operand arrives through a "v" constraint: when the value lives in an accumulation register (a 400+-VGPR kernel keeps hundreds there) or was
spilled, the register allocator materialises it with v_accvgpr_read / a reload IMMEDIATELY in front of the asm statement.

  python tools/check_asm_hazards.py file.s [...]      # device assembly (hipcc -S --offload-device-only)
  python tools/check_asm_hazards.py --build           # compiles both translation units of the product (+ every variant of tests/build_variants.py) and checks them

Exit status 1 if a hazard is found.  tests/test_abi.py runs it on the product sources (CPU test: hipcc cross-compiles here)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), i))
    return out


def is_valu(op):
    return op.startswith("v_") and not op.startswith("v_readlane") and not op.startswith("v_readfirstlane")


def wait_states(op, args):
    if op == "s_nop":
        return int(args.strip() or "0", 0) + 1
    return 1


def check(path, verbose=False):
    """returns (number of DPP instructions checked, list of findings)"""
    func = "?"
    window = []   # (op, dst regs, writes_exec, wait states, line no, text) -- most recent last
    found, n = [], 0
    with open(path) as f:
        for ln, line in enumerate(f, 1):
            s = line.split(";")[0].strip()
            if not s or s.startswith(".") or s.startswith("//"):
                continue
            if s.endswith(":"):
                if not s.startswith(".L"):
                    func = s[:-1]
                # a label is a possible branch target: instructions before it are not necessarily the predecessors; be conservative and keep the window
                continue
            parts = s.split(None, 1)
            op, args = parts[0], (parts[1] if len(parts) > 1 else "")
            ops = [a.strip() for a in args.split(",")]
            if "_dpp" in op and op.startswith("v_"):
                n += 1
                src0 = regs(ops[1]) if len(ops) > 1 else set()
                ws = 0
                for (pop, pdst, pexec, pws, pln, ptxt) in reversed(window):
                    if ws >= 5:
                        break
                    if ws < 2 and is_valu(pop) and (pdst & src0):
                        found.append((path, func, ln, s, pln, ptxt, "VALU write of the DPP source %d wait state(s) before" % ws))
                    if pexec and is_valu(pop):
                        found.append((path, func, ln, s, pln, ptxt, "VALU write of EXEC %d wait state(s) before a DPP instruction" % ws))
                    ws += pws
            dst = regs(ops[0]) if ops and (op.startswith("v_") or op.startswith("ds_") or op.startswith("global_") or op.startswith("scratch_") or op.startswith("buffer_") or op.startswith("flat_")) else set()
            wexec = op.startswith("v_cmpx")
            window.append((op, dst, wexec, wait_states(op, args), ln, s))
            if len(window) > 12:
                window.pop(0)
    return n, found


def build_asm(out_dir, extra=()):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    csrc = os.path.join(ROOT, "qm_door_amd", "csrc")
    ipra = [] if any(str(f).startswith("-enable-ipra") for f in extra) else ["-mllvm", "-enable-ipra=0"]   # as qm_door_amd/build.py
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", *ipra, "--offload-device-only", "-S", *extra]
    jobs = [(base + [os.path.join(csrc, "qmgpu_api.hip"), "-o", os.path.join(out_dir, "api.s")]),
            (base + ["-DQM_REAL=float", "-Dqmk=qmk32", os.path.join(csrc, "qmgpu_mpc32.hip"), "-o", os.path.join(out_dir, "mpc32.s")])]
    procs = [subprocess.Popen(j, stderr=subprocess.DEVNULL) for j in jobs]
    for p, j in zip(procs, jobs):
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, j)
    return [j[-1] for j in jobs]


def main(argv):
    files = [a for a in argv if not a.startswith("-")]
    if "--build" in argv:
        tmp = tempfile.mkdtemp(prefix="qm_asm_")
        files += build_asm(tmp, tuple(a for a in argv if a.startswith("-D") or a.startswith("-O") or a.startswith("-m") or a.startswith("-f")))
    bad = 0
    for p in files:
        n, found = check(p)
        print("%s: %d DPP instructions, %d hazards" % (p, n, len(found)))
        for (_, func, ln, s, pln, ptxt, why) in found[:40]:
            print("  %s:%d  %s\n      <- line %d  %s   [%s]" % (func[:60], ln, s, pln, ptxt, why))
        bad += len(found)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
