#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in an ISA dump (hipcc -S --cuda-device-only): where the SGPR-spill traffic
(v_readlane / v_writelane) sits relative to the hot loops (blocks with many table gathers).
usage: isa_blocks.py file.s kernel_name [min_instructions]"""
import re, sys
path, kern = sys.argv[1], sys.argv[2]
mini = int(sys.argv[3]) if len(sys.argv) > 3 else 60
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur, name = [], [], "entry"
for l in lines[start + 1:end + 1]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append((name, cur)); cur, name = [], m.group(1)
        continue
    t = l.strip()
    if t and not t.startswith((";", ".", "//")):
        cur.append(t.split()[0])
blocks.append((name, cur))
tot = {}
print("%-14s %6s %6s %6s %6s %6s %6s %6s %6s" % ("block", "insts", "valu", "ds_rd", "ds_wr", "rdlane", "wrlane", "s_nop", "vmem"))
for name, ins in blocks:
    n = len(ins)
    c = lambda p: sum(1 for x in ins if x.startswith(p))
    row = (n, c("v_"), c("ds_read") + c("ds_load"), c("ds_write") + c("ds_store"), c("v_readlane"), c("v_writelane"), c("s_nop"), c("global_") + c("buffer_"))
    for k, v in zip(("insts", "valu", "ds_rd", "ds_wr", "rdlane", "wrlane", "s_nop", "vmem"), row):
        tot[k] = tot.get(k, 0) + v
    if n >= mini:
        print("%-14s %6d %6d %6d %6d %6d %6d %6d %6d" % ((name,) + row))
print("total", tot, "blocks", len(blocks))
