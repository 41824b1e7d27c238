#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a `hipcc -S --cuda-device-only` listing.
    python tools/isa_mix.py /tmp/dec.s _Z17dec_conv16_kernelI4FP16Li2ELi3ELi3EEv8ConvArgs [min instructions per block]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
name, lo = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 16
start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])


def cls(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_cvt") or op.startswith("v_pk_"):
        return op
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("ds_", "global_", "buffer_", "flat_")):
        return op
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    return "salu" if op.startswith("s_") else op


blocks, cur, tot = [], ["entry", collections.Counter()], collections.Counter()
for l in lines[start + 1:end + 1]:
    m = re.match(r"^(\.LBB\S+):", l)
    if m:
        blocks.append(cur)
        cur = [m.group(1), collections.Counter()]
        continue
    t = l.strip().split()
    if not t or t[0][0] in ".;/":
        continue
    cur[1][cls(t[0])] += 1
    tot[cls(t[0])] += 1
blocks.append(cur)
for b in blocks:
    n = sum(b[1].values())
    if n >= lo:
        print(b[0], n, dict(sorted(b[1].items(), key=lambda kv: -kv[1])))
print("TOTAL", sum(tot.values()), dict(sorted(tot.items(), key=lambda kv: -kv[1])))
for l in lines[end:end + 80]:
    if re.search(r"NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize", l):
        print(l.strip())
