"""The headline kernel's generated code is what was measured.

Its speed moves by +-2.5 % with details of the generated code that source-level reasoning does not predict (round 3: a dead
`REFILL ? a : b` in place of `b`, the position of a scheduling boundary -- profiles/r03_dgroup_ab.txt), and such a change is silent:
every test stays green and only the driver's bench shows it, a round later. So the build's code for the plain quadrotor N=50 kernel
is pinned: if this test fails, the change may be perfectly fine -- A/B it (tools/headline_ab.py, the previous build against the new
one on ONE box) and then record the new hash with `python tools/headline_code_hash.py --record`."""
from __future__ import annotations

import json
import os

import pytest
from conftest import ROOT

from tools.headline_code_hash import RECORD, current_hash


def test_the_headline_kernel_is_the_code_that_was_measured():
    if not os.path.exists(RECORD):
        pytest.skip("no recorded hash")
    want = json.load(open(RECORD))
    got = current_hash()
    if got is None:
        pytest.skip("no build assembly (run __graft_entry__.build())")
    if got["compiler"] != want["compiler"]:
        pytest.skip(f"another compiler ({got['compiler']} against {want['compiler']}): the recorded hash does not apply")
    assert got["sha256"] == want["sha256"], (
        f"the headline kernel's code changed ({got['instructions']} instructions, recorded {want['instructions']}): A/B the builds with "
        f"tools/headline_ab.py on one box, then `python tools/headline_code_hash.py --record` (recorded state: {want['measured']})")
