"""Instruction-class stream of a kernel's innermost loop, from a hipcc -S file (ISA audit helper).

usage: python tools/loop_stream.py file.s <substring of the mangled kernel name>
M = MFMA, R = ds_read, W = ds_write, D = LDS-DMA, G = other VMEM load, v = VALU, w = s_waitcnt,
B = s_barrier, n = s_nop, s = other SALU.
"""
import collections
import re
import sys
import textwrap

path, pat = sys.argv[1], sys.argv[2]
s = open(path).read()
for m in re.finditer(r"^(_Z\S*):", s, re.M):
    name = m.group(1)
    if pat not in name:
        continue
    body = s[m.end():s.index(".end_amdhsa_kernel", m.end())]
    lines = body.split("\n")
    vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
    sp = len(re.findall(r"scratch_(load|store)", body))
    print(name, "vgpr", vg.group(1) if vg else "?", "scratch ops", sp)
    heads = [i for i, l in enumerate(lines) if "Loop Header" in l]
    for st in heads:
        lab = lines[st].split(":")[0]
        ends = [i for i, l in enumerate(lines) if i > st and "s_cbranch" in l and l.strip().endswith(lab)]
        if not ends:
            continue
        loop = lines[st:ends[-1] + 1]

        def cls(l):
            t = l.strip().split()[0] if l.strip() else ""
            if not t or t.startswith(";"):
                return ""
            for p, c in (("v_mfma", "M"), ("ds_read", "R"), ("ds_write", "W"), ("buffer_load_dwordx4", "D"), ("buffer_load", "G"),
                         ("global_load", "G"), ("v_", "v"), ("s_waitcnt", "w"), ("s_barrier", "B"), ("s_nop", "n"), ("s_", "s"), (".L", "L")):
                if t.startswith(p):
                    return c
            return "?"

        stream = "".join(cls(l) for l in loop)
        print(f"  loop {lab}: {len(stream)} instructions", dict(collections.Counter(stream)))
        print(textwrap.indent("\n".join(textwrap.wrap(stream, 140)), "    "))
        print("    waits:", dict(collections.Counter(l.strip() for l in loop if l.strip().startswith("s_waitcnt"))))
