"""Instruction mix of the loops of one kernel in a hipcc -S device listing (which issue port the K loop leans on).
usage: python tools/isa_loop.py file.s <substring of the mangled kernel name> [min loop length]"""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 200
text = open(path).read()
funcs = re.split(r"\n(?=_Z[^\n:]*:)", text)


def klass(op: str) -> str:
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


for f in funcs:
    name = f.split(":")[0]
    if not name.startswith("_Z") or pat not in name:
        continue
    lines = [l.strip() for l in f.split("\n")]
    labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"(\.LBB\d+_\d+):", l))}
    print(name)
    for i, l in enumerate(lines):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if not m or m.group(1) not in labels or labels[m.group(1)] >= i:
            continue
        body = [x for x in lines[labels[m.group(1)]:i] if x and not x.startswith((";", ".", "//")) and not x.endswith(":")]
        if len(body) < min_len:
            continue
        c = collections.Counter(klass(re.match(r"([a-z_0-9]+)", x).group(1)) for x in body if re.match(r"[a-z]", x))
        ops = collections.Counter(re.match(r"([a-z_0-9]+)", x).group(1) for x in body if re.match(r"[a-z]", x))
        print(f"  loop {m.group(1)}: {len(body)} instructions  {dict(c)}")
        print("    valu:", {k: v for k, v in ops.most_common() if k.startswith("v_") and not k.startswith("v_mfma")})
        print("    lds/vmem:", {k: v for k, v in ops.items() if k.startswith(("ds_", "buffer_"))})
        print("    waits:", dict(collections.Counter(x for x in body if x.startswith("s_waitcnt"))))
