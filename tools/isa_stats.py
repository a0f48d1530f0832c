"""Per-kernel instruction statistics from a hipcc -save-temps .s file (ISA audit helper)."""
import collections
import re
import sys

path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
s = open(path).read()
funcs = re.split(r"\n(?=_Z[^\n:]*:)", s)
KEYS = ["v_mfma", "ds_read_b128", "ds_read_b64", "ds_write_b128", "buffer_load_dwordx4", "global_load", "s_barrier",
        "scratch_load", "scratch_store", "v_mov_b32", "v_accvgpr", "s_waitcnt", "s_setprio", "s_nop", "v_perm", "v_pk_"]
for f in funcs:
    name = f.split(":")[0]
    if not name.startswith("_Z") or pat not in name:
        continue
    lines = [l.strip() for l in f.split("\n")]
    c = collections.Counter()
    for l in lines:
        m = re.match(r"([a-z_0-9]+)", l)
        if m:
            for k in KEYS:
                if m.group(1).startswith(k):
                    c[k] += 1
    print(name, len(lines))
    print("   ", {k: c[k] for k in KEYS if c[k]})
    print("    waits:", dict(collections.Counter(l for l in lines if l.startswith("s_waitcnt"))))
