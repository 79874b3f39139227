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
    """{case: {check name: deviations when the list was taken}}"""
    with open(ALLOW) as fh:
        return json.load(fh)["cases"]


def test_allow_list_is_short_and_names_no_benchmark_configuration():
    cases = allowed()
    assert len(cases) <= 20
    for case, checks in cases.items():
        assert not util.is_hard_case(case.split()[0]), case
        for n in checks:
            assert not util.is_hard_case(n.split()[0]) and not n.startswith(("naive step", "system5", "singlet n=")), n


@pytest.mark.gpu
def test_tie_breaks_taken_are_on_the_allow_list(request):
    names = {n for checks in allowed().values() for n in checks}
    taken = set(util.TIE_BREAKS)
    # a widened random-system sweep (RTT_FUZZ_SEEDS / RTT_FUZZ_FIRST) visits systems the list does not know
    widened = "RTT_FUZZ_SEEDS" in os.environ or "RTT_FUZZ_FIRST" in os.environ
    extra = sorted(t for t in taken - names if not (widened and t.startswith("seed ")))
    assert not extra, f"parity checks newly settled by the float64 tie-break: {extra}; details {[util.TIE_BREAKS[e] for e in extra]}"
