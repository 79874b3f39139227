"""Runs after the parity tests (file name): every check that was settled by the float64 tie-break of
`util.assert_parity` instead of the plain 1e-5 tolerance must be on the committed allow-list
(`tests/golden/parity_tie_breaks.json`), so that the set of ill-conditioned cases cannot grow unnoticed.  The
benchmark configurations never reach the tie-break: their checks are `hard`."""
import json
import os

import pytest

import util

ALLOW = os.path.join(util.GOLDEN, "parity_tie_breaks.json")


def allowed():
    with open(ALLOW) as fh:
        return json.load(fh)


def test_allow_list_is_short_and_names_no_benchmark_configuration():
    names = allowed()["allowed"]
    assert len(names) <= 40
    for n in names:
        first = n.split()[0]
        assert not util.is_hard_case(first), n


@pytest.mark.gpu
def test_tie_breaks_taken_are_on_the_allow_list(request):
    names = set(allowed()["allowed"])
    taken = set(util.TIE_BREAKS)
    # a widened random-system sweep (RTT_FUZZ_SEEDS / RTT_FUZZ_FIRST) visits systems the list does not know
    widened = "RTT_FUZZ_SEEDS" in os.environ or "RTT_FUZZ_FIRST" in os.environ
    extra = sorted(t for t in taken - names if not (widened and t.startswith("seed ")))
    assert not extra, f"parity checks newly settled by the float64 tie-break: {extra}; details {[util.TIE_BREAKS[e] for e in extra]}"
