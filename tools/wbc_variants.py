"""Build variants of the SAME kernel sources that must all compute the same cycle.

wbc_kernel is a 400+-VGPR kernel with called functions; in round 2 its result depended three times on how it had been compiled.  This
tool builds the library several times (different macro / optimisation / inlining choices, none of which changes the arithmetic) and
compares one MPC + WBC cycle between them on the GPU, for every contact mode.

  python tools/wbc_variants.py --build [names...]   # here (hipcc cross-compiles): qm_door_amd/build/variants/<name>/libqmgpu_<name>.so
  python tools/wbc_variants.py --run [names...]     # on the GPU box: compares every built variant with the product library
  python tools/wbc_variants.py --asm  [names...]    # here: device assembly of each variant through tools/check_asm_hazards.py

tests/test_gpu_two_builds.py uses VARIANTS / lib_path / compare() of this module for the variants marked `required`.
"""
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
VDIR = os.path.join(ROOT, "qm_door_amd", "build", "variants")

# name -> (extra hipcc flags, required by the -m gpu test, what it changes)
VARIANTS = {
    "ticks": (("-DQM_RICCATI_TIMING",), True, "phase clocks (s_memtime + s_waitcnt at every phase boundary): different register allocation and scheduling"),
    "o2": (("-O2",), True, "-O2 instead of -O3: different inlining / unrolling decisions"),
    "noinl": (("-mllvm", "-inline-threshold=40"), True, "inliner threshold 40 (default 225 at -O3): helpers that are inlined in the product become calls"),
    # (required until round 5, when it was bit-identical.  Round 6: after a semantically neutral edit of qp_dev.h -- the pinned-row count checked after the slots are assigned
    #  instead of before -- the bare build returns torques 1e-9 .. 7e-5 off on every RF_RH (mode 5) instance of the sweep and on no other, while the product agrees with the CPU
    #  restatement to 5e-14 on those instances and with the five other variants bit for bit.  Bisected over the round's commits with both builds of each: the bare builtin's memory
    #  semantics at the hand-off are whatever this LLVM gives it; the product states them -- fence release / barrier / fence acquire -- and does not depend on that.  Kept buildable.)
    "bare": (("-DQM_WAVE_SYNC_BARE",), False, "round 2's QM_WAVE_SYNC (bare wave barrier, no fences): NOT a build that has to agree since round 6"),
    "ipra": (("-mllvm", "-enable-ipra=1"), True, "interprocedural register allocation on (LLVM's default for AMDGPU; rounds 1-2 shipped this)"),
    "opq_noipra": (("-DQM_WBC_OPAQUE_MASK=511",), True, "whole LDS carve of wbc_kernel behind one opaque address-space-3 base: wrong torques with IPRA on (round 2), correct with it off"),
    "opq": (("-DQM_WBC_OPAQUE_MASK=511", "-mllvm", "-enable-ipra=1"), False, "round 2's failing experiment (REPRODUCER: returns wrong torques): opaque LDS base + IPRA on"),
}
# bisection of the failing mask (wbc_kernel.h: nine array groups): coarse groups {inputs+coordinates, task arrays, vectors} = bits 0,1 | 3,4 | 6,7,8 fail
# together; F = that set, F minus one fine group each
_F = 0b111011011
VARIANTS["f_all"] = (("-DQM_WBC_OPAQUE_MASK=%d" % _F, "-mllvm", "-enable-ipra=1"), False, "groups {inputs, coordinates, tasks, vectors} opaque, IPRA on: correct (the failure needs four of the five coarse groups)")


_OPQ = ("-DQM_WBC_OPAQUE_MASK=511", "-mllvm", "-enable-ipra=1")   # the failing combination; each experiment below changes ONE thing
VARIANTS["x_inline"] = ((*_OPQ, "-DQM_WBC_EXP=2"), False, "failing combination, bodyPass inlined (no call on the helper wavefront's extra path): correct")
VARIANTS["x_wave2"] = ((*_OPQ, "-DQM_WBC_EXP=3"), False, "failing combination, desired pass on helper wavefront 2: fails too (the failure follows the call)")
VARIANTS["x_drain"] = ((*_OPQ, "-DQM_WBC_EXP=4"), False, "failing combination, s_waitcnt vmcnt(0) lgkmcnt(0) before the fork-join loop: still fails (not memory ordering)")
VARIANTS["x_sleep"] = ((*_OPQ, "-DQM_WBC_EXP=5"), False, "failing combination, helpers sleep before the fork-join loop: still fails (not timing)")
# timing experiments (tools/variant_timing.py): the instruction scheduler's strategy for the whole translation unit
VARIANTS["qptrace"] = (("-DQM_QP_TRACE=0",), False, "device printf of instance 0's level-solver iterations (experiments: tools/wbc_variants.py --build qptrace, then a batch of one through that library)")
VARIANTS["s_maxilp"] = (("-mllvm", "-amdgpu-sched-strategy=max-ilp"), False, "LLVM's max-ILP scheduling strategy (timing experiment)")
VARIANTS["s_iterilp"] = (("-mllvm", "-amdgpu-sched-strategy=iterative-ilp"), False, "LLVM's iterative ILP scheduling strategy (timing experiment)")
VARIANTS["s_maxmem"] = (("-mllvm", "-amdgpu-sched-strategy=max-memory-clause"), False, "LLVM's max-memory-clause scheduling strategy (timing experiment)")
VARIANTS["s_track"] = (("-mllvm", "-amdgpu-use-amdgpu-trackers"), False, "AMDGPU register-pressure trackers in the scheduler (timing experiment)")
VARIANTS["s_bias0"] = (("-mllvm", "-amdgpu-schedule-metric-bias=0"), False, "scheduler metric bias 0: latency over occupancy (timing experiment)")
VARIANTS["s_bias100"] = (("-mllvm", "-amdgpu-schedule-metric-bias=100"), False, "scheduler metric bias 100: occupancy only (timing experiment)")
VARIANTS["s_relax"] = (("-mllvm", "-amdgpu-schedule-relaxed-occupancy"), False, "relaxed occupancy targets (timing experiment)")
VARIANTS["s_nopost"] = (("-mllvm", "-enable-post-misched=0"), False, "no post-RA machine scheduler (timing experiment)")
VARIANTS["s_nounroll"] = (("-fno-unroll-loops",), False, "no loop unrolling beyond the pragmas (timing experiment)")
VARIANTS["dump"] = (("-DQM_WBC_DUMP",), False, "product + LDS dump checkpoints of instance 0")
VARIANTS["dump_opq"] = (("-DQM_WBC_DUMP", "-DQM_WBC_OPAQUE_MASK=511", "-mllvm", "-enable-ipra=1"), False, "failing combination + LDS dump checkpoints (the instrumentation of the helper loop hides the failure)")

# LDS carve of wbc_kernel.h (doubles), for naming what differs between two dumps
CARVE = {'IN': 0, 'Q': 160, 'BODY': 256, 'DOF': 896, 'WR': 1040, 'M': 1160, 'NLE': 1736, 'JF': 1760, 'JA': 2048, 'MISC': 2192, 'A': 2336, 'B': 3128, 'D0': 3152, 'F0': 5168, 'Z': 5280,
         'ZN': 6612, 'AZ': 7944, 'DZ': 8758, 'K': 10830, 'G': 12162, 'VH': 13494, 'VEC(x z g rd rhs dz)': 14374, 'fhat lam wt tz': 14590, 'red': 14814, 'ctl': 15838, 'BODY2': 15846, 'DOF2': 16486,
         'END': 16630}
CHECKPOINTS = ["0 model", "1 task0 assembled", "2 reduced data L0", "3 after ipm L0", "4 x after L0", "5 Z after null space 0", "6 after null space 1", "7 end"]


def dump_of(lib):
    """runs the WBC alone (bench scenario, B = 64, from the oracle-free inputs of cycle_all_modes) and returns the LDS images of instance 0"""
    import ctypes as C
    import torch
    import bench
    import gpu_harness as G
    from qm_door_amd import api
    itf = api.QMInterface(lib=lib)
    B, N = 64, 20
    sc = bench.build_scenario(itf, B, seed=1)
    sol = G.make_solver(itf, B, N)
    mb = G.MpcBatch(sc["x0"], sc["tt"], sc["ts"], np.full(B, sc["nev"], dtype=np.int32), np.tile(sc["ev"], (B, 1)), np.tile(sc["md"], (B, 1)), N)
    wb = G.WbcBatch(sc["rbd"], np.full(B, 0.002), np.full(B, 20.0), np.zeros((B, 30)))
    sol.cycle(mb.args, G.dev(np.zeros(B), torch.float64), wb.args)
    w = wb.results()
    n = 8 * 17000
    buf = np.zeros(n)
    lib.qmgpu_debug_wbc_dump.argtypes = [C.c_void_p, C.c_int]
    assert lib.qmgpu_debug_wbc_dump(buf.ctypes.data_as(C.c_void_p), n) == 0
    sol.close()
    img = buf.reshape(8, 17000)
    for wv in range(3):   # what the helper wavefronts saw: [op, exec lo, exec hi, wave, first active lane, job M, N, K] per loop iteration
        rec = img[wv, 16640:17000].reshape(45, 8)
        print("  helper wavefront %d:" % (wv + 1))
        for it in range(8):
            r = rec[it]
            print("     it %d: op %d exec %08x%08x wave %d first lane %d job M N K = %d %d %d" % (it, r[0], int(r[2]), int(r[1]), r[3], r[4], r[5], r[6], r[7]))
    return img[:, :CARVE["END"]], w


def compare_dumps(a_name, b_name):
    from qm_door_amd import abi
    da, wa = dump_of(abi.load_library(lib_path(a_name)))
    db, wb_ = dump_of(abi.load_library(lib_path(b_name)))
    print("status", a_name, int((wa["status"] != 0).sum()), b_name, int((wb_["status"] != 0).sum()))
    names = list(CARVE.items())
    for cp in range(8):
        diffs = []
        for (nm, lo), (_, hi) in zip(names[:-1], names[1:]):
            x, y = da[cp, lo:hi], db[cp, lo:hi]
            both_nan = np.isnan(x) & np.isnan(y)
            bad = ~both_nan & ~(x == y)
            if bad.any():
                idx = np.flatnonzero(bad)
                diffs.append("%s: %d of %d differ, first at +%d (%.6g vs %.6g), max |d| %.3g" % (nm, bad.sum(), hi - lo, idx[0], x[idx[0]], y[idx[0]], np.nanmax(np.abs(np.nan_to_num(x[bad]) - np.nan_to_num(y[bad])))))
        print("checkpoint", CHECKPOINTS[cp], "--", "identical" if not diffs else "")
        for d in diffs:
            print("     ", d)
    np.savez(os.path.join(ROOT, "gpurun_out", "wbc_dumps.npz"), a=da, b=db)


def lib_path(name):
    return os.path.join(VDIR, name, "libqmgpu_%s.so" % name)


def source_hash(name):
    """sha256 over the kernel / ABI sources and the variant's flags: what a variant library was built from (modification times do not survive every copy)"""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(ROOT, "include", "qmgpu.h"), os.path.join(ROOT, "qm_door_amd", "build.py")]
    for d, _, fs in os.walk(os.path.join(ROOT, "qm_door_amd", "csrc")):
        files += [os.path.join(d, f) for f in fs]
    for f in sorted(files):
        h.update(os.path.relpath(f, ROOT).encode()); h.update(open(f, "rb").read())
    h.update(repr(VARIANTS[name][0]).encode())
    return h.hexdigest()


def is_current(name):
    p = lib_path(name)
    try:
        return os.path.exists(p) and open(p + ".srchash").read().strip() == source_hash(name)
    except OSError:
        return False


def build(names, force=True):
    from qm_door_amd import build as qb

    def one(n):
        out = lib_path(n)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        qb.build_library(force=force, extra_flags=VARIANTS[n][0], out=out, obj_dir=os.path.dirname(out))
        open(out + ".srchash", "w").write(source_hash(n) + "\n")
        return out
    with ThreadPoolExecutor(max_workers=3) as ex:
        for p in ex.map(one, names):
            print("built", p, flush=True)


def cycle_all_modes(lib, B=256, N=40, cycles=2):
    """One MPC + policy + WBC cycle (twice: the second WBC runs from the first one's inputLast) on the bench scenario, then the WBC alone on
    every contact mode (the ten modes of SURVEY section 8c(7), B instances each, random configurations) -- returns plain numpy results."""
    import torch
    import bench
    import gpu_harness as G
    import support as S
    from qm_door_amd import api
    itf = api.QMInterface(lib=lib)
    sc = bench.build_scenario(itf, B, seed=1)
    sol = G.make_solver(itf, B, N)
    mb = G.MpcBatch(sc["x0"], sc["tt"], sc["ts"], np.full(B, sc["nev"], dtype=np.int32), np.tile(sc["ev"], (B, 1)), np.tile(sc["md"], (B, 1)), N)
    wb = G.WbcBatch(sc["rbd"], np.full(B, 0.002), np.full(B, 20.0), np.zeros((B, 30)))
    t_eval = G.dev(np.zeros(B), torch.float64)
    for _ in range(cycles):
        sol.cycle(mb.args, t_eval, wb.args)
    r, w = mb.results(), wb.results()
    out = {"X": r["X"], "U": r["U"], "mode": r["mode"], "wbc": w["out"], "wbc_status": w["status"]}
    # the WBC alone on every contact mode, both variants, start-up branch and normal branch
    rng = np.random.default_rng(5)
    x_nom = itf.initial_state
    modes = np.array([15, 9, 6, 0, 10, 5, 13, 7, 14, 11], dtype=np.int32)
    md = np.repeat(modes, B // 8 + 1)[: 10 * (B // 8)]
    nb = md.shape[0]
    xd = np.tile(x_nom, (nb, 1)) + rng.uniform(-1, 1, (nb, 30)) * np.r_[np.full(6, 0.1), np.full(3, 0.05), np.full(3, 0.05), np.full(18, 0.1)]
    ud = np.zeros((nb, 30))
    for i in range(nb):
        st = [(md[i] >> (3 - c)) & 1 for c in range(4)]
        ns = max(1, sum(st))
        for c in range(4):
            if st[c]:
                ud[i, 3 * c: 3 * c + 3] = [rng.uniform(-5, 5), rng.uniform(-5, 5), 27.87 * 9.81 / ns * rng.uniform(0.8, 1.2)]
        ud[i, 12:] = rng.uniform(-0.3, 0.3, 18)
    rbd = np.zeros((nb, 55))
    xm = xd + rng.uniform(-1, 1, (nb, 30)) * 0.02
    rbd[:, 0:3] = xm[:, 9:12]; rbd[:, 3:6] = xm[:, 6:9]; rbd[:, 6:24] = xm[:, 12:30]
    rbd[:, 24:48] = rng.uniform(-0.2, 0.2, (nb, 24))
    rbd[:, 48:51] = xm[:, 6:9] + np.array([0.6, 0.0, 0.3]); rbd[:, 54] = 1.0
    sol2 = G.make_solver(itf, nb, 4)
    for variant, time in ((0, 20.0), (0, 1.0), (1, 20.0)):
        wb2 = G.WbcBatch(rbd, np.full(nb, 0.002), np.full(nb, time), ud * 0.9, state_desired=xd, input_desired=ud, mode=md, variant=variant)
        sol2.wbc(wb2.args)
        w2 = wb2.results()
        out["wbc_modes_v%d_t%g" % (variant, time)] = w2["out"]
        out["wbc_modes_status_v%d_t%g" % (variant, time)] = w2["status"]
    sol2.close()
    # the ten slowest ticks of round 6's steady-state leg (robots whose torque limits cannot hold: held-variable form given up, interior point in front of the first level): the
    # outputs AND the pass counts of every solve -- five of seven builds once lost the interior point's hand-over there (same torques, 24 iterations more)
    d = np.load(os.path.join(ROOT, "tests", "golden", "wbc_slow_ticks.npz"))
    ns = len(d["mode"])
    sol3 = G.make_solver(itf, ns, 4)
    wb3 = G.WbcBatch(d["rbd"], d["period"].astype(np.float64), d["time"].astype(np.float64), d["il"].copy(), d["xd"], d["ud"], d["mode"].astype(np.int32), 0, carry=True)
    sol3.wbc(wb3.args)
    w3 = wb3.results()
    out["wbc_slow_ticks"] = w3["out"]; out["wbc_slow_ticks_status"] = w3["status"]
    out["wbc_slow_ticks_passes"] = np.ascontiguousarray(w3["working_set"][:, 13:15]).view(np.uint8).astype(np.int32)
    sol3.close()
    sol.close()
    return out


def compare(ref, got):
    """worst relative-inf deviation per output block, and whether the integer blocks are identical"""
    rep = {}
    for k in ref:
        a, b = ref[k], got[k]
        if a.dtype.kind in "iu":
            rep[k] = {"equal": bool(np.array_equal(a, b)), "nonzero_ref": int((a != 0).sum()), "nonzero_got": int((b != 0).sum())}
        else:
            ok = np.isfinite(b).all()
            rep[k] = {"finite": bool(ok), "max_rel": float(np.abs(np.nan_to_num(a - b, nan=1e300)).max() / max(1.0, np.abs(a).max()))}
    return rep


def run(names):
    from qm_door_amd import abi
    ref = cycle_all_modes(abi.load_library())
    report = {}
    for n in names:
        if not os.path.exists(lib_path(n)):
            print(n, "not built", flush=True)
            continue
        try:
            got = cycle_all_modes(abi.load_library(lib_path(n)))
            report[n] = compare(ref, got)
        except Exception as e:  # a variant that fails to run is a finding, not a crash of the tool
            report[n] = {"error": repr(e)}
        worst = max((v.get("max_rel", 0.0) for v in report[n].values() if isinstance(v, dict)), default=None)
        bad_int = [k for k, v in report[n].items() if isinstance(v, dict) and v.get("equal") is False]
        print("%-10s worst float deviation %s; integer blocks that differ: %s" % (n, worst, bad_int), flush=True)
    out = os.path.join(ROOT, "gpurun_out", "wbc_variants.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", out)


def asm(names):
    import check_asm_hazards as H
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    for n in names:
        d = os.path.join(VDIR, n, "asm")
        os.makedirs(d, exist_ok=True)
        for f in H.build_asm(d, VARIANTS[n][0]):
            cnt, found = H.check(f)
            print(n, os.path.basename(f), cnt, "DPP instructions,", len(found), "hazards")


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if not a.startswith("-")] or list(VARIANTS)
    if "--build" in sys.argv:
        build(names)
    if "--asm" in sys.argv:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        asm(names)
    if "--run" in sys.argv:
        run(names)
    if "--dumps" in sys.argv:
        compare_dumps(names[0], names[1])
