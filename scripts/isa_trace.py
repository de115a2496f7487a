"""Instruction-class trace of the busiest barrier-to-barrier stretch of a kernel in a hipcc object:
M = MFMA, r = ds_read, D = LDS-DMA, W(..) = s_waitcnt, | = s_barrier, S = scratch, . = anything else.
python scripts/isa_trace.py <object> <kernel-name-substring>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_isa as C  # noqa: E402

k = C.disassemble(sys.argv[1])
name = [n for n in k if sys.argv[2] in n][0]
lines = [ln.split("//")[0].strip() for ln in k[name]]
bars = [i for i, ln in enumerate(lines) if ln.startswith("s_barrier")]
best = None
for a, b in zip(bars, bars[1:] + [len(lines)]):
    n = sum("v_mfma" in ln for ln in lines[a:b])
    if best is None or n > best[0]:
        best = (n, a, b)
n, a, b = best
out = []
for ln in lines[a:b]:
    op = ln.split()[0]
    if op.startswith("v_mfma"):
        c = "M"
    elif op.startswith("ds_read"):
        c = "r"
    elif op.startswith("s_waitcnt"):
        c = "W(" + ln.split(None, 1)[1] + ")"
    elif op.startswith("global_load_lds"):
        c = "D"
    elif op.startswith("s_barrier"):
        c = "|"
    elif "scratch" in op:
        c = "S"
    else:
        c = "."
    out.append(c)
print(name, n, "MFMAs")
print("".join(out))
