"""Instruction mix between the first and the last MFMA of every kernel of a device assembly file (hipcc --cuda-device-only -S):
in a matrix-bound kernel every VALU instruction there is matrix-pipe time.   python tools/isa_mix.py file.s"""
import re
import sys
from collections import Counter

name, body = None, []
kernels = []
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        if name:
            kernels.append((name, body))
        name, body = m.group(1), []
    elif name and line.startswith("\t") and not line.startswith("\t.") and not line.startswith("\t;"):
        body.append(line.split()[0])
        if body[-1] == "s_endpgm":
            kernels.append((name, body))
            name, body = None, []
for name, body in kernels:
    idx = [i for i, op in enumerate(body) if op.startswith("v_mfma")]
    if len(idx) < 8:
        continue
    seg = body[idx[0]:idx[-1] + 1]
    c = Counter()
    for op in seg:
        if op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith("v_"): c["valu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem"] += 1
        elif op.startswith("s_waitcnt"): c["wait"] += 1
        elif op.startswith("s_cbranch"): c["branch"] += 1
        elif op.startswith("s_barrier"): c["barrier"] += 1
        else: c["salu"] += 1
    print(f"{c['mfma']:5d} mfma {c['valu']:5d} valu ({c['valu'] / c['mfma']:.2f}/mfma) {c['lds']:4d} lds {c['vmem']:4d} vmem {c['branch']:4d} br {c['barrier']:3d} bar  {name[:100]}")
