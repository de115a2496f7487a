#!/usr/bin/env python3
"""Static checks on the gfx950 ISA of the field kernels (run at build time on the CPU box).

The A-operand reads of the field kernel are hand-placed `ds_read_b128` with counted
`s_waitcnt lgkmcnt(n)`.  That is only valid while nothing else that returns out of order shares
the counter: scalar-memory loads do.  This script disassembles nothing -- it asks hipcc for the
.s of mnrf_field.hip and asserts, per field kernel:
  * no s_load / s_buffer_load between the first and the last MFMA,
  * no scratch traffic between the first and last MFMA of the non-gradient variants,
  * the hand-placed reads and waits are present.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mirror_nerf_amd", "csrc", "mnrf_field.hip")
SRC_SPLIT = os.path.join(ROOT, "mirror_nerf_amd", "csrc", "mnrf_field_split.hip")


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "field.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-pragma-unroll-threshold=1000000", "-S",
                        "--cuda-device-only", SRC, "-o", out], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    ok = True
    for m in re.finditer(r"^(_ZN4mnrf2s[12]1[26]field_(?:bwd_)?kernel\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        mf = [i for i, l in enumerate(body) if "v_mfma_f32_16x16x4" in l]
        inner = body[mf[0]:mf[-1] + 1]
        smem = [l for l in inner if re.search(r"\bs_(buffer_)?load_", l)]
        scratch = [l for l in inner if "scratch_" in l]
        reads = sum("ds_read_b128" in l for l in inner)
        counted = sum(bool(re.search(r"s_waitcnt lgkmcnt\([1-4]\)", l)) for l in inner)
        grad = "Lb1EEEv" in name or "bwd" in name
        print(f"{name}: {len(mf)} MFMA, {reads} ds_read_b128, {counted} counted waits, "
              f"{len(smem)} scalar loads inside, {len(scratch)} scratch ops inside")
        if smem or (scratch and not grad) or counted < 100:
            ok = False
    # split-f16 kernels: same rule for their counted lgkmcnt waits (the LDS-DMA pieces go through the compiler's builtin,
    # which owns M0 and the VMEM hazards); and nothing may spill
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "split.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-pragma-unroll-threshold=1000000", "-S",
                        "--cuda-device-only", SRC_SPLIT, "-o", out], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    n = 0
    for m in re.finditer(r"^(_ZN4mnrf\d+h2x?18field_split_kernel\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        mf = [i for i, l in enumerate(body) if "v_mfma_f32_16x16x32_f16" in l]
        inner = body[mf[0]:mf[-1] + 1]
        smem = [l for l in inner if re.search(r"\bs_(buffer_)?load_", l)]
        scratch = [l for l in body if "scratch_" in l]
        counted = sum(bool(re.search(r"s_waitcnt lgkmcnt\([24]\)", l)) for l in inner)
        dma = sum("global_load_lds_dwordx4" in l for l in body)
        print(f"{name}: {len(mf)} MFMA, {dma} LDS-DMA pieces, {counted} counted waits, {len(smem)} scalar loads inside, "
              f"{len(scratch)} scratch ops")
        n += 1
        if smem or scratch or counted < 100:
            ok = False
    if n != 8:
        print(f"expected 8 split kernels, found {n}")
        ok = False
    if not ok:
        print("ISA CHECK FAILED")
        sys.exit(1)
    print("ISA check ok")


if __name__ == "__main__":
    main()
