"""bench.py's host-side pieces that need no GPU."""
import json
import os

from conftest import ROOT

import bench


def test_csrc_hash_follows_code_not_comments():
    """profiles/k_trace_counters.json is stamped with a hash of the device sources; bench.py withholds the roofline fractions
    when the tree's hash differs.  A comment or blank-line edit must not do that, a code edit must."""
    base = bench.csrc_hash()
    plain = lambda path: open(path, errors="replace").read()  # noqa: E731
    commented = bench.csrc_hash(lambda path: "// a new comment\n/* and\n a block */\n" + plain(path).replace("\n", "\n\n   "))
    assert commented == base
    edited = bench.csrc_hash(lambda path: plain(path) + ("\nstatic int jade_extra;\n" if path.endswith("jade_trace.h") else ""))
    assert edited != base


def test_counter_file_belongs_to_the_tree():
    """The committed per-ray counters were cut from this build of csrc/ (tools/summarize_prof.py): the driver's bench line then
    carries roofline fractions instead of nulls."""
    ctr = json.load(open(os.path.join(ROOT, "profiles", "k_trace_counters.json")))
    assert ctr["csrc_sha"] == bench.csrc_hash(), "csrc/ changed since profiles/k_trace_counters.json was cut: re-run tools/profile_r03.sh + tools/summarize_prof.py"
    for key in ("C3", "C5"):
        e = ctr[key]
        assert e["valu_lane_ops_per_ray"] > 0 and e["l2_requests_per_ray"] > 0 and e["hbm_bytes_per_ray"] > 0 and e["fetch_size_factor"] == 1.0
