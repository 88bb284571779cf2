#!/usr/bin/env python3
"""Static instruction mix of one kernel's gfx950 assembly per innermost loop (scripts/isa_of.sh writes the .s):
    scripts/isa_loops.py /tmp/isa_of.s
Blocks are attributed to the loop header named in the label comments the compiler emits."""
import collections
import re
import sys

path = sys.argv[1]
cur = "(no loop)"
depth = {"(no loop)": 0}
counts = collections.defaultdict(lambda: collections.Counter())
first_line = {}
for ln, line in enumerate(open(path), 1):
    t = line.strip()
    m = re.match(r"^(\.LBB\d+_\d+):\s*;\s*(.*)$", t)
    if m:
        lab, com = m.group(1), m.group(2)
        if "Loop Header" in com and "Depth=" in com and "in Loop" not in com:
            cur = lab
            depth[cur] = int(re.search(r"Depth=(\d+)", com).group(1))
        else:
            m2 = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", com)
            if m2:
                cur = ".L" + m2.group(1)
                depth[cur] = int(m2.group(2))
            else:
                cur = "(no loop)"
        first_line.setdefault(cur, ln)
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:  # label without comment: keeps the loop of the previous block unless a header comment follows
        continue
    if "Loop Header" in t and t.startswith(";"):
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    op = t.split()[0]
    if op.startswith("v_"):
        kind = "valu"
        if "dpp" in t or op.startswith(("v_cmp", "v_cndmask", "v_readlane", "v_readfirstlane", "v_addc", "v_max3", "v_pk_", "v_and_or", "v_mbcnt")) or "_u64" in op or "_b64" in op:
            counts[cur]["valu_slow"] += 1
    elif op.startswith("s_cbranch") or op == "s_branch":
        kind = "branch"
    elif op.startswith(("s_waitcnt", "s_nop")):
        kind = "wait/nop"
    elif op.startswith(("s_load", "s_buffer")):
        kind = "smem"
    elif op.startswith("s_"):
        kind = "salu"
    elif op.startswith("ds_"):
        kind = "lds"
    elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        kind = "vmem"
    else:
        kind = "other"
    counts[cur][kind] += 1
    first_line.setdefault(cur, ln)
print("%-14s %5s %5s %5s(%4s) %5s %6s %4s %5s %5s" % ("loop", "line", "depth", "valu", "slow", "salu", "branch", "lds", "vmem", "wait"))
for k in sorted(counts, key=lambda k: first_line.get(k, 0)):
    c = counts[k]
    print("%-14s %5d %5d %5d(%4d) %5d %6d %4d %5d %5d" % (k, first_line.get(k, 0), depth.get(k, 0), c["valu"], c["valu_slow"], c["salu"], c["branch"], c["lds"], c["vmem"], c["wait/nop"]))
