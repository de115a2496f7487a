"""Deterministic MirrorNeRF weights for fixtures and tests: re-export of mirror_nerf_amd.synthetic
(the builder lives in the package so that bench.py and smoke() do not import from tests/)."""
from mirror_nerf_amd.synthetic import (ALL_MIRROR, OPAQUE, ROUGH, STRADDLE, apply_tweaks, checksum,  # noqa: F401
                                       make_tcnn_table,
                                       make_state_dict)
