"""Bank-conflict checker for the LDS operand-tile layout used by gemm_mfma.hip.

ds_read_b128 on gfx950 is serviced in four 16-lane groups; within a group, lanes whose 16-byte
slots (byte address / 16 mod 16, i.e. a 256-byte bank row) collide cost one extra LDS cycle each
(MI355X_MICROARCH.md, LDS table).  This script enumerates the fragment reads of the GEMM kernels
and reports the worst-case ways per group, so a layout can be validated without a GPU.
"""
import itertools

GROUPS = [
    [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
    [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
    [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63],
]


def tile_byte(row, chunk, swz):
    """Byte offset of logical (row, 16-byte chunk) in a unit made of 8-row x 128-byte subtiles."""
    return (row >> 3) * 1024 + (row & 7) * 128 + ((chunk ^ swz(row)) * 16)


def ways(addr_of_lane):
    worst = 0
    for grp in GROUPS:
        slots = {}
        for lane in grp:
            a = addr_of_lane(lane)
            slots.setdefault((a // 16) % 16, set()).add(a)
        worst = max(worst, max(len(v) for v in slots.values()))
    return worst


def row_plain(lane, tile):
    return tile * 16 + (lane & 15)


def row_interleaved(lane, tile):
    # two 16-row MFMA tiles cover 32 rows so that a lane's 4+4 accumulator rows are 8 consecutive n
    r = lane & 15
    return (tile >> 1) * 32 + 8 * (r >> 2) + (r & 3) + 4 * (tile & 1)


def check(swz, name):
    res = {}
    for rname, rowf in (("plain", row_plain), ("interleaved", row_interleaved)):
        for tile in range(4):
            for half in (0, 1):
                w = ways(lambda l: tile_byte(rowf(l, tile), (l >> 4) + 4 * half, swz))
                res[(rname, tile, half)] = w
    worst = max(res.values())
    print(f"{name}: worst {worst}-way", {k: v for k, v in res.items() if v > 1})
    return worst


if __name__ == "__main__":
    cands = {
        "none": lambda r: 0,
        "(r>>1)&7": lambda r: (r >> 1) & 7,
        "r&7": lambda r: r & 7,
        "(r>>1)&3": lambda r: (r >> 1) & 3,
        "((r>>1)&3)|((r>>3)&1)<<2": lambda r: ((r >> 1) & 3) | (((r >> 3) & 1) << 2),
        "((r>>1)&3)^((r>>3)&3)": lambda r: ((r >> 1) & 3) ^ ((r >> 3) & 3),
        "((r>>1)&7)^((r>>4)&1)": lambda r: ((r >> 1) & 7) ^ ((r >> 4) & 1),
    }
    for k, f in cands.items():
        check(f, k)
