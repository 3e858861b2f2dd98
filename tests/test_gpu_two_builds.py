"""-m gpu: build variants of the SAME kernel sources must compute the same cycle.

wbc_kernel is a 400+-VGPR kernel with called functions; in round 2 a variant of it was computed correctly by one build and wrongly by another of
the same sources, which the parity tests of a single binary cannot see.  tools/wbc_variants.py builds the library with phase clocks
(-DQM_RICCATI_TIMING), at -O2, with a low inliner threshold, with LLVM's interprocedural register allocation on and with the LDS carve behind an opaque base (round 2's bare
wave barrier was one of them until round 5: tools/wbc_variants.py says why it no longer has to agree); every variant runs one MPC + WBC cycle at batch
256 (twice, the second WBC from the first one's inputLast) and the WBC alone on every contact mode (both controllers, start-up branch), and must
agree with the product library: modes / status words bit-exact, X, U and the WBC output within 1e-9.

The variant libraries are built by __graft_entry__.build() (git-ignored, they travel with the snapshot like libqmgpu.so); one that is missing or
older than the kernel sources is a FAILURE here, not a skip -- a green run of this file means the variants really ran."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import wbc_variants as V  # noqa: E402

REQUIRED = [n for n, (_, req, _) in V.VARIANTS.items() if req]


@pytest.fixture(scope="module")
def product_cycle(hip_lib):
    return V.cycle_all_modes(hip_lib)


def test_at_least_three_variants_are_required():
    assert len(REQUIRED) >= 3 and {"ticks", "o2"} <= set(REQUIRED)


@pytest.mark.parametrize("name", REQUIRED)
def test_variant_agrees_with_the_product_build(name, product_cycle):
    from qm_door_amd import abi
    assert V.is_current(name), f"{V.lib_path(name)} is missing or stale: python tools/wbc_variants.py --build {name} (or __graft_entry__.build())"
    got = V.cycle_all_modes(abi.load_library(V.lib_path(name)))
    assert (product_cycle["wbc_status"] == 0).all()
    rep = V.compare(product_cycle, got)
    for key, r in rep.items():
        if "equal" in r:
            assert r["equal"], (name, key, r)          # contact modes, WBC status words: bit-exact
        else:
            assert r["finite"] and r["max_rel"] <= 1e-9, (name, key, r)


def test_every_contact_mode_is_exercised(product_cycle):
    # the mode sweep of cycle_all_modes: ten contact modes x 32 instances, three controller configurations, all levels converge
    for k in ("wbc_modes_status_v0_t20", "wbc_modes_status_v0_t1", "wbc_modes_status_v1_t20"):
        assert product_cycle[k].shape == (320,) and (product_cycle[k] == 0).all(), k
    assert np.isfinite(product_cycle["wbc_modes_v0_t20"]).all()
