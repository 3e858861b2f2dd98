"""Gait front end on the device (qmgpu_gait_schedule_batch): per-instance mode schedules tiled from gait.info templates by a
kernel, bit-identical to the host tiler qmgpu_tile_gait (what upstream GaitSchedule does with the template GaitTopicPublisher.cpp:31-44 publishes)."""
import ctypes as C

import numpy as np
import pytest

import support as S
from qm_door_amd import abi, api

GAITS = ["stance", "trot", "standing_trot", "flying_trot", "pace", "standing_pace", "dynamic_walk", "static_walk", "amble", "lindyhop", "skipping", "pawup"]


def _cases(lib, batch, seed):
    gs = api.GaitSchedule(lib=lib)
    names = []
    for g in GAITS:
        try:
            gs.template(g); names.append(g)
        except abi.QmGpuError:
            pass
    assert len(names) >= 6
    templates = [gs.template(g) for g in names]
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, len(names), batch).astype(np.int32)
    t_phase0 = np.round(rng.uniform(0.0, 3.0, batch), 3)
    t_begin = t_phase0 + rng.uniform(-1.0, 6.0, batch)
    t_begin[::7] = t_phase0[::7]                                    # horizon starting exactly on the first cycle
    t_end = t_begin + rng.uniform(0.5, 3.0, batch)
    t_end[::11] = t_begin[::11] + 30.0                              # too long for MAX_EVENTS with the fast gaits: capacity status
    prev = rng.choice(np.array([15, 15, 9, 6, 10, 5, 13, 0], dtype=np.int32), batch)    # the mode running when the gait command arrives
    return gs, names, templates, idx, t_phase0, t_begin, t_end, prev


def _check(gs, names, idx, t_phase0, t_begin, t_end, n, ev, md, st, prev, transition):
    over = 0
    gs.lib.qmgpu_switch_gait.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    for i in range(len(idx)):
        g = gs.template(names[idx[i]])
        nn = abi.i32(0); e = (abi.d * abi.MAX_EVENTS)(); m = (abi.i32 * (abi.MAX_EVENTS + 1))()
        rc = gs.lib.qmgpu_switch_gait(C.byref(g), int(prev[i]), transition, float(t_phase0[i]), float(t_begin[i]), float(t_end[i]), C.byref(nn), e, m)
        if rc == abi.ERR_CAPACITY:
            over += 1
            assert st[i] == abi.ERR_CAPACITY and n[i] == 0 and (md[i] == 15).all()
            continue
        assert rc == 0 and st[i] == 0
        assert n[i] == nn.value
        assert np.array_equal(ev[i], np.array(e[:])), (i, names[idx[i]])       # bit-identical event times (padding 1e300 included)
        assert np.array_equal(md[i], np.array(m[:], dtype=np.int32))
    return over


def test_emu_gait_schedule_matches_host_tiler():
    lib = abi.load_library(S.build_emu())
    itf = api.QMInterface(lib=lib)
    B = 96
    gs, names, templates, idx, t_phase0, t_begin, t_end, prev = _cases(lib, B, 1)
    sol = api.GpuSolver(itf, max_batch=4, max_nodes=4)
    n, ev, md, st = np.zeros(B, dtype=np.int32), np.zeros((B, abi.MAX_EVENTS)), np.zeros((B, abi.MAX_EVENTS + 1), dtype=np.int32), np.zeros(B, dtype=np.int32)
    trans = itf.problem.settings.phase_transition_stance_time
    assert trans > 0
    sol.gait_schedule(templates, idx, t_phase0, t_begin, t_end, n, ev, md, st, prev_mode=prev)
    assert _check(gs, names, idx, t_phase0, t_begin, t_end, n, ev, md, st, prev, trans) >= 1
    # without prev_mode: STANCE before the first cycle, i.e. the plain tiler
    sol.gait_schedule(templates, idx, t_phase0, t_begin, t_end, n, ev, md, st)
    _check(gs, names, idx, t_phase0, t_begin, t_end, n, ev, md, st, np.full(B, 15), trans)


def test_one_mode_non_stance_template_with_a_short_period_keeps_its_mode_until_t_end():
    """a template whose modes are all equal but not STANCE, tiled over more than MAX_EVENTS + 2 periods: the bounded loop stops early without
    overflowing (every cycle merges away); the default final STANCE must still begin at the end of the last whole cycle at or after t_end -- not where the
    loop stopped, which would put a spurious switch to STANCE inside the horizon.  Host and device tilers, bit-identical."""
    lib = abi.load_library(S.build_emu())
    itf = api.QMInterface(lib=lib)
    g = api.GaitSchedule(lib=lib).template("trot")
    g.num_modes = 2; g.modes[0] = 9; g.modes[1] = 9
    g.switching_times[0] = 0.0; g.switching_times[1] = 0.01; g.switching_times[2] = 0.02          # period 0.02: 500 periods in 10 s
    n = abi.i32(0); ev = (abi.d * abi.MAX_EVENTS)(); md = (abi.i32 * (abi.MAX_EVENTS + 1))()
    abi.check(lib, lib.qmgpu_tile_gait(C.byref(g), 1.0, 1.0, 11.0, C.byref(n), ev, md))
    assert n.value == 2 and list(md[:3]) == [15, 9, 15] and ev[0] == 1.0
    assert 11.0 <= ev[1] <= 11.0 + 0.02 + 1e-12                     # the one switch back to STANCE lies at / after t_end
    sol = api.GpuSolver(itf, max_batch=4, max_nodes=4)
    nn, e2, m2, st = np.zeros(1, dtype=np.int32), np.zeros((1, abi.MAX_EVENTS)), np.zeros((1, abi.MAX_EVENTS + 1), dtype=np.int32), np.zeros(1, dtype=np.int32)
    sol.gait_schedule([g], np.zeros(1, dtype=np.int32), np.array([1.0]), np.array([1.0]), np.array([11.0]), nn, e2, m2, st)
    assert st[0] == 0 and nn[0] == 2 and np.array_equal(e2[0], np.array(ev[:])) and np.array_equal(m2[0], np.array(md[:], dtype=np.int32))


def test_gait_switch_inserts_the_transition_stance():
    """Hand-checked: a trot command arriving during RF_LH (6): STANCE for phaseTransitionStanceTime, then LF_RH / RF_LH cycles; arriving during
    LF_RH (9, the template's first mode) or during STANCE: no transition phase (upstream GaitSchedule::insertModeSequenceTemplate)."""
    lib = abi.load_library(S.build_emu())
    gs = api.GaitSchedule(lib=lib)
    g = gs.template("trot")
    lib.qmgpu_switch_gait.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    def run(prev, tr=0.4):
        nn = abi.i32(0); e = (abi.d * abi.MAX_EVENTS)(); m = (abi.i32 * (abi.MAX_EVENTS + 1))()
        assert lib.qmgpu_switch_gait(C.byref(g), prev, tr, 1.0, 0.5, 2.0, C.byref(nn), e, m) == 0
        return nn.value, np.array(e[:nn.value]), list(m[:nn.value + 1])
    n, ev, md = run(6)
    assert md[:4] == [6, 15, 9, 6] and np.allclose(ev[:3], [1.0, 1.4, 1.75])
    n, ev, md = run(9)
    assert md[:3] == [9, 6, 9] and np.allclose(ev[:2], [1.35, 1.7])            # merged with the running LF_RH phase
    n, ev, md = run(15)
    assert md[:3] == [15, 9, 6] and np.allclose(ev[:2], [1.0, 1.35])


@pytest.mark.gpu
def test_gpu_gait_schedule_matches_host_tiler_and_feeds_the_mpc(interface, oracle):
    import torch
    import gpu_harness as G
    B = 512
    gs, names, templates, idx, t_phase0, t_begin, t_end, prev = _cases(interface.lib, B, 2)
    sol = G.make_solver(interface, B, 20)
    dn = torch.zeros(B, dtype=torch.int32, device="cuda"); dev_ = torch.zeros((B, abi.MAX_EVENTS), dtype=torch.float64, device="cuda")
    dmd = torch.zeros((B, abi.MAX_EVENTS + 1), dtype=torch.int32, device="cuda"); dst = torch.zeros(B, dtype=torch.int32, device="cuda")
    sol.gait_schedule(templates, G.dev(idx, torch.int32), G.dev(t_phase0, torch.float64), G.dev(t_begin, torch.float64), G.dev(t_end, torch.float64), dn, dev_, dmd, dst,
                      prev_mode=G.dev(prev, torch.int32))
    torch.cuda.synchronize()
    n, ev, md, st = dn.cpu().numpy(), dev_.cpu().numpy(), dmd.cpu().numpy(), dst.cpu().numpy()
    assert _check(gs, names, idx, t_phase0, t_begin, t_end, n, ev, md, st, prev, interface.problem.settings.phase_transition_stance_time) >= 1
    # the device-made schedules go straight into the MPC call (no host round trip): compared with the oracle on the same schedule
    N = 20
    x0 = S.perturbed_states(interface.initial_state, B, seed=6)
    tgt = S.nominal_target(oracle, interface.initial_state)
    tt = np.zeros((B, 1)); ts = np.tile(tgt, (B, 1, 1)).copy()
    mb = G.MpcBatch(x0, tt, ts, n, ev, md, N, t0=t_begin)
    mb.args.sched_num_events = dn.data_ptr(); mb.args.sched_event_times = dev_.data_ptr(); mb.args.sched_modes = dmd.data_ptr()
    sol.mpc(mb.args)
    r = mb.results()
    assert np.isfinite(r["X"]).all()
    ref = S.Oracle(interface.problem, fast=True).cycle_batch(N, x0, tt, ts, n, ev, md, t0=t_begin)      # every instance, each on its own device-made schedule
    S.assert_parity(S.parity_report("gait_frontend_512xN20_all_templates", r, ref, tau=False))


def test_malformed_gait_templates_are_refused_by_host_and_device_tilers():
    """num_modes outside 1..QMGPU_MAX_EVENTS is refused before switching_times[num_modes] is read; non-finite times are refused; a template whose modes
    are all STANCE ends (it used to tile for ever: no event is ever pushed) -- on the host tiler and on the device kernel (emulated here)."""
    lib = abi.load_library(S.build_emu())
    itf = api.QMInterface(lib=lib)
    gs = api.GaitSchedule(lib=lib)
    lib.qmgpu_switch_gait.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    good = gs.template("trot")
    too_many, none = gs.template("trot"), gs.template("trot")
    too_many.num_modes = abi.MAX_EVENTS + 1; none.num_modes = 0
    stance = abi.Gait(); stance.num_modes = 2
    stance.modes[0] = stance.modes[1] = 15
    stance.switching_times[0], stance.switching_times[1], stance.switching_times[2] = 0.0, 0.5, 1.0

    def host(g, t_end, prev=15):
        nn = abi.i32(0); e = (abi.d * abi.MAX_EVENTS)(); m = (abi.i32 * (abi.MAX_EVENTS + 1))()
        return lib.qmgpu_switch_gait(C.byref(g), prev, 0.1, 0.0, 0.0, t_end, C.byref(nn), e, m), nn.value

    assert host(too_many, 1.0)[0] == abi.ERR_INVALID_ARGUMENT and host(none, 1.0)[0] == abi.ERR_INVALID_ARGUMENT
    assert host(good, float("inf"))[0] == abi.ERR_INVALID_ARGUMENT and host(good, float("nan"))[0] == abi.ERR_INVALID_ARGUMENT
    assert host(stance, 1e9) == (0, 0)                         # a billion seconds of stance: no events, and it returns
    assert host(good, 1e9)[0] == abi.ERR_CAPACITY              # a billion seconds of trot: does not fit, and it returns
    sol = api.GpuSolver(itf, max_batch=6, max_nodes=4)
    templates = (abi.Gait * 4)(good, too_many, none, stance)
    idx = np.array([0, 1, 2, 3, 3, 0], dtype=np.int32)
    t_end = np.array([1.0, 1.0, 1.0, 1e9, 2.0, np.inf])
    n, ev, md, st = np.zeros(6, dtype=np.int32), np.zeros((6, abi.MAX_EVENTS)), np.zeros((6, abi.MAX_EVENTS + 1), dtype=np.int32), np.zeros(6, dtype=np.int32)
    sol.gait_schedule(templates, idx, np.zeros(6), np.zeros(6), t_end, n, ev, md, st)
    assert st.tolist() == [0, abi.ERR_INVALID_ARGUMENT, abi.ERR_INVALID_ARGUMENT, 0, 0, abi.ERR_INVALID_ARGUMENT]
    assert n[3] == 0 and n[4] == 0 and (md[3] == 15).all() and n[0] > 0
