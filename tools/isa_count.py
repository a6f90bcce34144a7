#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc assembly dump (-save-temps): VALU / SALU / LDS / VMEM
counts, s_nop count, registers.  usage: isa_count.py file.s <substring of the mangled name> [--dump out.s]"""
import collections, re, sys
src = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r"^(_Z\S*):\s", src, re.M):
    name = m.group(1)
    if pat not in name:
        continue
    end = src.find(".Lfunc_end", m.end())
    body = src[m.end():end]
    ins = [l.split()[0] for l in (x.strip() for x in body.split("\n")) if l and not l.startswith((".", ";")) and not l.endswith(":")]
    kind = lambda i: "valu" if i.startswith("v_") else "salu" if i.startswith("s_") else "lds" if i.startswith("ds_") else "vmem"
    c = collections.Counter(kind(i) for i in ins)
    regs = re.search(re.escape(name) + r"\.num_vgpr, (\d+)", src)
    print(name[:90], dict(c), "s_nop", sum(i == "s_nop" for i in ins), "vgpr", regs.group(1) if regs else "?")
    if "--dump" in sys.argv:
        open(sys.argv[sys.argv.index("--dump") + 1], "w").write(body)
