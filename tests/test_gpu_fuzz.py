"""GPU (-m gpu): the randomised parity cases of tests/fuzz_cases.py with FIXED seeds and a bounded number of cases each — the soak
tools of round 2 (tools/fuzz_*.py) brought under pytest: STD and focus maps bit-exact against the oracle, TEN_WM within one LSB of
M16, on random shapes around the tile and chunk sizes, both view layouts, the main kernel variants.  One process, one context at a time."""
import pytest

import fuzz_cases

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import lfi_oracle_c as oc
    oc.build()
    return oc


def test_fuzz_fixed_focus_blend(gpu):
    bad = fuzz_cases.fuzz_blend(gpu, _oracle(), n_cases=40, seed=1)
    assert not bad, bad[:5]


def test_fuzz_focus_maps(gpu):
    bad = fuzz_cases.fuzz_focus(gpu, _oracle(), n_cases=40, seed=1)
    assert not bad, bad[:5]


def test_fuzz_all_focus_renders(gpu):
    bad = fuzz_cases.fuzz_allfocus(gpu, _oracle(), n_cases=40, seed=3)
    assert not bad, bad[:5]
