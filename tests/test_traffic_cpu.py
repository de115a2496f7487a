"""Evidence hygiene: bench.py's `roofline.traffic` is a STATIC file -- PMC counters cannot be read inside the
timed run -- so nothing used to fail when a kernel changed and profiles/traffic.json did not.  Every entry that bench.py reads
carries the sha1 of the kernel's source set as it was when the counters were taken (scripts/pmc_reduce.py); this test recomputes it."""
import json
import os

from mirror_nerf_amd import source_hash as SH

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the entries bench.py / benchlegs.py look up
READ_BY_BENCH = ["mnrf::h3::field_split_kernel<false,false,false,false>", "mnrf::h3::field_split_kernel<false,false,false,true>",
                 "mnrf::mf::tcnn_encode_kernel"]


def test_traffic_entries_describe_the_current_sources():
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    stale = []
    for k in READ_BY_BENCH:
        e = t.get(k)
        assert e is not None, f"profiles/traffic.json has no entry for {k}"
        assert e.get("source_sha1"), f"{k}: no source_sha1 recorded with its PMC pass (scripts/profile_round.sh regenerates it)"
        if e["source_sha1"] != SH.source_sha1(k.split(" [")[0]):
            stale.append((k, e.get("commit")))
    assert not stale, f"kernel sources changed after their PMC pass: {stale} -- run scripts/profile_round.sh / scripts/pmc_tcnn.sh and merge"
    for k, c in t.get("pmc", {}).items():
        if c.get("source_sha1") and SH.files_of(k.split(" [")[0]) is not None:
            assert c["source_sha1"] == SH.source_sha1(k.split(" [")[0]), f"pmc[{k}] is stale"


def test_every_kernel_family_lists_existing_sources():
    for _prefix, files in SH.FAMILIES:
        for f in files:
            assert os.path.exists(os.path.join(SH.CSRC, f)), f
