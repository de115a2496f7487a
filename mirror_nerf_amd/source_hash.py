"""sha1 of the kernel sources a measured number depends on (round 5, evidence hygiene): profiles/traffic.json records, next to the
commit, the hash of the source set of every kernel whose PMC counters it holds; tests/test_traffic_cpu.py fails when a kernel's
sources changed after its counters were taken (the static `roofline.traffic` of bench.py would silently describe other code)."""
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
_FIELD = ["mnrf_field_split.inc", "mnrf_field_split_common.inc", "mnrf_field_stream.inc", "mnrf_composite.inc", "mnrf_layout.h",
          "mnrf_field_args.h"]
FAMILIES = (            # kernel-name prefix -> source files (relative to mirror_nerf_amd/csrc)
    ("mnrf::h3::field_split_kernel", _FIELD + ["mnrf_field_split3.hip"]),
    ("mnrf::h2::field_split_kernel", _FIELD + ["mnrf_field_split.hip"]),
    ("mnrf::h2x::field_split", _FIELD + ["mnrf_field_split.hip", "mnrf_field_split_bwd.inc", "mnrf_dwp.h"]),
    ("mnrf::dwp_", ["mnrf_dwp.hip", "mnrf_dwp.h", "mnrf_layout.h"]),
    ("mnrf::mf::tcnn_", ["mnrf_tcnn.hip"]),
    ("mnrf::tcnn_", ["mnrf_tcnn.hip"]),
    ("mnrf::s2::field_kernel", ["mnrf_field.hip", "mnrf_field_impl.inc", "mnrf_layout.h", "mnrf_field_args.h"]),
)


def files_of(kernel):
    for prefix, files in FAMILIES:
        if kernel.startswith(prefix):
            return files
    return None


def source_sha1(kernel):
    """sha1 over the (name, contents) of the kernel's source set, or None for a kernel outside FAMILIES."""
    files = files_of(kernel)
    if files is None:
        return None
    h = hashlib.sha1()
    for f in files:
        h.update(f.encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()
