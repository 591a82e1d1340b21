"""What the compiler reports for the kernels of csrc/uhdr_kernels.hip (device-only compile for gfx950, no GPU needed).

Round 2's dominant kernel, k_apply_s4<HLG / PQ>, spilled six registers inside its per-cell loop (ScratchSize 28 bytes per lane, and
an s_waitcnt vmcnt(0) per cell behind the reload).  The streaming kernels are held to no scratch at all from here on.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import kernel_resources  # noqa: E402


@pytest.fixture(scope="module")
def table():
    if not os.path.exists(kernel_resources.HIPCC):
        pytest.skip("hipcc not present")
    return kernel_resources.resources()


def test_no_kernel_uses_scratch(table):
    assert len(table) > 100   # every template instance of the file
    # (SGPR "spills" are v_writelane / v_readlane into a VGPR, no memory: reported, not refused)
    bad = {k: v for k, v in table.items() if v.get("scratch", 0) != 0 or v.get("vgpr_spill", 0) != 0}
    assert not bad, bad


def test_streaming_kernels_present_and_at_their_occupancy(table):
    apply_fast = [k for k in table if "k_apply_s4<" in k]
    gen = [k for k in table if "k_generate<" in k]
    assert len(apply_fast) == 8 and len(gen) >= 40, (len(apply_fast), len(gen))
    for k in apply_fast:   # two blocks of 512 threads per CU = four waves per SIMD: at most 128 registers
        assert table[k]["vgpr"] <= 128 and table[k]["occupancy"] >= 4, (k, table[k])
    # the headline pair: 64 x 4K HLG, filtered generate with the exact path deferred (4 spans per block) and the HLG table walk
    g = [k for k in gen if "<1, true, false, true, 4, true, 256>" in k]
    assert len(g) == 1 and table[g[0]]["occupancy"] >= 5, (g, [table[x] for x in g])
