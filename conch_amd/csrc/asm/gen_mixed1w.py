#!/usr/bin/env python3
"""Generator of the ONE-WAVE-PER-SIMD int4 x fp16 mixed-precision GEMM kernel for gfx950 (hand-allocated registers, assembly).

Replaces, for the reference benchmark's configuration (uint4 weights with a bias, group 128, no per-group zero points, fp16),
the mixed-precision instantiation of the reference's Triton kernel (conch/kernels/quantization/gemm.py:176-216 dequantisation,
:219-457 K loop, launcher :482-545) where the column-strip HIP kernel (gemm_mixed_strip.hip) runs today.  Same idea as that
kernel -- every wave owns a COLUMN STRIP of the tile, fetches the packed words of its own columns, dequantises them in registers
(bit-identical to w_ref: (w & mask) | 0x6400 is the fp16 number 1024 + q, an exact v_pk_add_f16 / v_pk_fma_f16 removes
1024 + bias, ONE rounding in v_pk_mul_f16 by the group scale) and feeds v_mfma_f32_16x16x32_f16 -- but with ONE wave per SIMD:

  * tile 256 rows x (4 waves x 16 W) columns, W = 3 (192 columns: C4 = 232 tiles, one per CU) or 4 (256 columns: the
    reference's README shape = 256 tiles); a wave's 16 x W accumulator tiles (192 / 256 registers) live in a[0:...];
  * a hand-placed instruction stream: per 64-element K step and wave 32 W MFMAs, the 17-instruction dequantisation chains of the
    NEXT step's 2 W chunks dealt one or two instructions per MFMA gap, 32 X fragment reads (each feeds W MFMAs: half the LDS
    read bytes of the eight-wave kernel), 2 W + W packed-word / scale loads straight into registers, 8 LDS-DMA pieces of X;
  * X (256 rows x 128 bytes per step) through a FOUR-stage LDS-DMA ring (128 KiB; three steps of flight), the library's swizzled
    row image; ONE workgroup barrier per step, placed so that the reads it waits for were issued three fragment slots earlier;
  * every vector-memory count is exact (vmcnt retires in order): `vmcnt(6 W + 16)` at the barrier, `vmcnt(3 W + 9)` in front of
    the first dequantisation slice of a step.

Contract (the launcher in gemm_asm.hip checks; everything else keeps the HIP kernels): fp16 activations / scales / output,
4-bit weights (any bias, no zero-point tensor; a scalar zero point is folded into the bias), group_size == 128, K % 128 == 0,
N % 16 == 0, 16-byte aligned activation rows, every array below 2 GiB.

usage: gen_mixed1w.py OUT.s
"""
from __future__ import annotations

import sys

MT = 16                      # 16-row m tiles per wave (256 rows)
XSLOTS = 2 * MT              # X fragments per step: (k half, m tile)
STAGE = 256 * 128            # 32 KiB: one K step of X
NSTAGES = 4
RING = NSTAGES * STAGE       # 128 KiB
XR = 8                       # X fragment ring (registers: 4 each)
NSTAMPS = 4

# ---- SGPRs ----
S_KARG, S_WG = 0, 2
S_X, S_WQ, S_WS, S_C = 8, 10, 12, 14                       # pointers as loaded (s[8:15])
S_M, S_N, S_K, S_LDX, S_LDQ, S_LDS, S_LDC, S_OFF = 16, 17, 18, 19, 20, 21, 22, 23
S_TM, S_TN, S_NWG, S_MAGIC_PG, S_GRID, S_MAGIC_LAST = 24, 25, 26, 27, 28, 29
S_PROBE = 30                                                # s[30:31]
S_T = 32                                                    # s32..s45 temporaries
S_WAVE = 46
S_BM0, S_BN0 = 47, 48
S_M0X = 49                  # LDS-DMA destination of this wave's piece 0 in the stage being filled
S_XK = 50                   # K byte offset of the X step whose pieces are being issued
S_WOFF0, S_WOFF1 = 51, 52   # byte offsets of the word rows (k half 0 / 1) of the step whose words are being loaded
S_SOFF = 53                 # byte offset of the scale row of that step's group
S_STEP = 54                 # step whose words are being loaded (t + 2)
S_CNT = 55
S_MAGIC = 56                # 0x64006400
S_K16 = 57                  # 0.0625 packed
S_PLO, S_PHI = 58, 59       # v_perm selectors
S_NRX = 60                  # X bytes (descriptor num_records)
S_DX, S_DQ, S_DS, S_DC = 64, 68, 72, 76                     # descriptors
S_WSTEP, S_WHALF, S_SSTEP = 80, 81, 82                      # 32 ldq, 16 ldq, 2 lds (bytes per step / half step / group)
S_STAMP = 84                                                # s[84:99]
S_END = S_STAMP + 4 * NSTAMPS

# ---- VGPRs ----
V_TID = 0
V_XDMA = 1                  # v1..v8: source byte offsets of the wave's 8 X pieces
V_WOFF = 9                  # packed-word byte offset of this lane (k group g, column r of n tile 0)
V_SCOFF = 10                # scale byte offset of this lane (column r of n tile 0)
V_XLO, V_XHI = 11, 12       # X fragment read bases of the CURRENT stage (chunk g / g + 4)
V_XLON, V_XHIN = 13, 14     # ... of the NEXT stage
V_MASK, V_MASKHI = 15, 16   # 0x000f000f, 0x00f000f0
V_SUB, V_NSUB16 = 17, 18    # packed 1024 + off ; -(64 + off)
V_T = 19                    # v19..v23 temporaries
V_XR = 24                   # v24..v55: X fragment ring (8 x 4)
V_DQ = 56                   # v56..v65: two dequantisation chains x (w1, a, b, c, d)
V_W = 66                    # weights, raw words, scales: laid out in Gen.__init__ (depends on W)


class Gen:
    def __init__(self, name: str, w: int, probe: bool = False):
        self.name, self.W, self.probe = name, w, probe
        self.lines: list[str] = []
        self.lgkm: list[str] = []
        W = w
        self.v_wfrag = V_W                          # [parity][nt][half] x 4 registers
        self.v_raw = self.v_wfrag + 2 * W * 2 * 4   # [parity][nt][half]
        self.v_sc = self.v_raw + 2 * W * 2          # [parity][nt] raw fp16 scale (low 16 bits)
        self.v_scp = self.v_sc + 2 * W              # [nt] packed scale of the chunk set being dequantised
        self.v_e = (self.v_scp + W + 1) & ~1         # epilogue temporaries (even: 64-bit tuples)
        self.v_end = self.v_e + 24
        assert self.v_end <= 256, self.v_end
        self.nacc = MT * W * 4                      # a[0 : nacc)
        assert self.nacc <= 256

    # ---- emission -----------------------------------------------------------------------------------------------------------
    def L(self, s: str) -> str:
        return f".L{self.name}_{s}"

    def e(self, s: str, comment: str = "") -> None:
        self.lines.append(f"\t{s}" + (f"\t; {comment}" if comment else ""))

    def label(self, s: str) -> None:
        self.lines.append(f"{s}:")

    def c(self, s: str) -> None:
        self.lines.append(f"\t; {s}")

    def stamp(self, k: int) -> None:
        if self.probe:
            self.e(f"s_memtime s[{S_STAMP + 4 * k}:{S_STAMP + 4 * k + 1}]")
            self.e(f"s_memrealtime s[{S_STAMP + 4 * k + 2}:{S_STAMP + 4 * k + 3}]")

    # ---- LDS read tracking -----------------------------------------------------------------------------------------------------
    def ds_read128(self, dst: int, addr: int, off: int, tag: str) -> None:
        assert 0 <= off < 65536, off
        self.e(f"ds_read_b128 v[{dst}:{dst + 3}], v{addr} offset:{off}", tag)
        self.lgkm.append(tag)

    def wait_tag(self, tag: str) -> None:
        if tag not in self.lgkm:
            return
        idx = self.lgkm.index(tag)
        after = len(self.lgkm) - 1 - idx
        self.e(f"s_waitcnt lgkmcnt({min(after, 15)})")
        self.lgkm = self.lgkm[idx + 1:] if after <= 15 else self.lgkm[-15:]

    # ---- register names ----------------------------------------------------------------------------------------------------------
    def acc(self, i: int, nt: int) -> str:
        t = i * self.W + nt
        return f"a[{4 * t}:{4 * t + 3}]"

    def wfrag(self, par: int, nt: int, h: int) -> int:
        return self.v_wfrag + ((par * self.W + nt) * 2 + h) * 4

    def raw(self, par: int, nt: int, h: int) -> int:
        return self.v_raw + (par * self.W + nt) * 2 + h

    def sc(self, par: int, nt: int) -> int:
        return self.v_sc + par * self.W + nt

    # ---- loads ---------------------------------------------------------------------------------------------------------------------
    def x_piece(self, p: int) -> None:
        """LDS-DMA piece p (rows 8 (4 p + wave) ..) of the X step at S_XK into the stage at S_M0X."""
        self.e(f"s_add_i32 m0, s{S_M0X}, {4096 * p}")
        self.e("s_nop 0")
        self.e(f"buffer_load_dwordx4 v{V_XDMA + p}, s[{S_DX}:{S_DX + 3}], s{S_XK} offen lds", f"X piece {p}")

    def word_loads(self, par: int) -> None:
        """Packed words and scales of the step S_STEP -> raw[par], sc[par] (3 W vector-memory operations)."""
        for nt in range(self.W):
            self.e(f"buffer_load_dword v{self.raw(par, nt, 0)}, v{V_WOFF}, s[{S_DQ}:{S_DQ + 3}], s{S_WOFF0} offen offset:{64 * nt}")
            self.e(f"buffer_load_dword v{self.raw(par, nt, 1)}, v{V_WOFF}, s[{S_DQ}:{S_DQ + 3}], s{S_WOFF1} offen offset:{64 * nt}")
        for nt in range(self.W):
            self.e(f"buffer_load_ushort v{self.sc(par, nt)}, v{V_SCOFF}, s[{S_DS}:{S_DS + 3}], s{S_SOFF} offen offset:{32 * nt}")

    def advance_words(self) -> None:
        """S_STEP += 1 and the three byte offsets that follow it (scale row = group = step / 2)."""
        e = self.e
        e(f"s_add_u32 s{S_STEP}, s{S_STEP}, 1")
        e(f"s_add_u32 s{S_WOFF0}, s{S_WOFF0}, s{S_WSTEP}")
        e(f"s_add_u32 s{S_WOFF1}, s{S_WOFF1}, s{S_WSTEP}")
        e(f"s_lshr_b32 s{S_T}, s{S_STEP}, 1")
        e(f"s_mul_i32 s{S_SOFF}, s{S_T}, s{S_SSTEP}")

    # ---- dequantisation --------------------------------------------------------------------------------------------------------------
    def dequant_ops(self, par: int, nt: int, h: int, chain: int) -> list[str]:
        """The 17 instructions that turn raw word (par, nt, h) into the four operand registers wfrag(par, nt, h): int4, fp16."""
        t = V_DQ + 5 * chain
        w1, a, b, c, d = t, t + 1, t + 2, t + 3, t + 4
        w0 = self.raw(par, nt, h)
        out = self.wfrag(par, nt, h)
        scp = self.v_scp + nt
        return [
            f"v_lshrrev_b32 v{w1}, 8, v{w0}",
            f"v_and_or_b32 v{a}, v{w0}, v{V_MASK}, s{S_MAGIC}",
            f"v_and_or_b32 v{b}, v{w0}, v{V_MASKHI}, s{S_MAGIC}",
            f"v_and_or_b32 v{c}, v{w1}, v{V_MASK}, s{S_MAGIC}",
            f"v_and_or_b32 v{d}, v{w1}, v{V_MASKHI}, s{S_MAGIC}",
            f"v_pk_add_f16 v{a}, v{a}, v{V_SUB} neg_lo:[0,1] neg_hi:[0,1]",
            f"v_pk_fma_f16 v{b}, v{b}, s{S_K16}, v{V_NSUB16}",
            f"v_pk_add_f16 v{c}, v{c}, v{V_SUB} neg_lo:[0,1] neg_hi:[0,1]",
            f"v_pk_fma_f16 v{d}, v{d}, s{S_K16}, v{V_NSUB16}",
            f"v_pk_mul_f16 v{a}, v{a}, v{scp}",
            f"v_pk_mul_f16 v{b}, v{b}, v{scp}",
            f"v_pk_mul_f16 v{c}, v{c}, v{scp}",
            f"v_pk_mul_f16 v{d}, v{d}, v{scp}",
            f"v_perm_b32 v{out}, v{b}, v{a}, s{S_PLO}",
            f"v_perm_b32 v{out + 2}, v{b}, v{a}, s{S_PHI}",
            f"v_perm_b32 v{out + 1}, v{d}, v{c}, s{S_PLO}",
            f"v_perm_b32 v{out + 3}, v{d}, v{c}, s{S_PHI}",
        ]

    def pack_scales(self, par: int) -> list[str]:
        return [f"v_lshl_or_b32 v{self.v_scp + nt}, v{self.sc(par, nt)}, 16, v{self.sc(par, nt)}" for nt in range(self.W)]

    def dequant_stream(self, par: int) -> list[str]:
        """All VALU work of dequantising step (par): scale packing, then the 2 W chunks, two chains interleaved."""
        chunks = [(nt, h) for h in range(2) for nt in range(self.W)]
        ops: list[str] = []
        ops += [f"v_and_b32 v{self.sc(par, nt)}, 0xffff, v{self.sc(par, nt)}" for nt in range(self.W)]
        ops += self.pack_scales(par)
        for k in range(0, len(chunks), 2):
            pair = chunks[k:k + 2]
            streams = [self.dequant_ops(par, nt, h, ci) for ci, (nt, h) in enumerate(pair)]
            for j in range(17):
                for s_ in streams:
                    ops.append(s_[j])
        return ops

    # ---- X fragments -------------------------------------------------------------------------------------------------------------------
    def read_x(self, f: int, nxt: bool, tag: str) -> None:
        """X fragment f = (k half f // 16, m tile f % 16) of the current / next stage into ring slot f % XR."""
        h, i = divmod(f, MT)
        base = (V_XHIN if h else V_XLON) if nxt else (V_XHI if h else V_XLO)
        self.ds_read128(V_XR + 4 * (f % XR), base, 2048 * i, tag)

    @staticmethod
    def x_issue_slot(f: int) -> int:
        """X slot in which fragment f (0..31 this step, 32..35 = 0..3 of the next step) is requested."""
        if f >= XSLOTS:
            return 28 + (f - XSLOTS)
        if f >= 28:
            return 22 + (f - 28)
        return f - 4

    # ---- one K step ------------------------------------------------------------------------------------------------------------------------
    def step(self, par: int, tag: str, nxt: str) -> None:
        """Step t with weights wfrag[par] (dequantised during step t - 1); dequantises wfrag[1 - par] from raw[1 - par] for step
        t + 1, loads raw[par] for step t + 2, requests X of steps t + 3 / t + 4."""
        W = self.W
        e = self.e
        dq = self.dequant_stream(1 - par)
        nslots = XSLOTS * W
        # deal the dequantisation: none in the first 2 X slots (its operands are waited for there), then evenly to the end
        first = 2 * W
        room = nslots - first
        per = [0] * nslots
        for k in range(len(dq)):
            per[first + (k * room) // len(dq)] += 1
        dqi = 0
        for x in range(XSLOTS):
            h, i = divmod(x, MT)
            if x == 28:
                self.c("---- barrier: X of step t + 1 has landed everywhere; every wave has read this stage ----")
                e(f"s_waitcnt vmcnt({6 * W + 16})")
                e("s_waitcnt lgkmcnt(0)")
                self.lgkm = []
                e("s_barrier")
            self.wait_tag(f"{tag}.x{x}")
            xr = V_XR + 4 * (x % XR)
            for nt in range(W):
                slot = x * W + nt
                wf = self.wfrag(par, nt, h)
                acc = self.acc(i, nt)
                e(f"v_mfma_f32_16x16x32_f16 {acc}, v[{wf}:{wf + 3}], v[{xr}:{xr + 3}], {acc}", f"x slot {x} (k half {h}, m tile {i}), n tile {nt}")
                for _ in range(per[slot]):
                    e(dq[dqi])
                    dqi += 1
                if nt == 0:
                    # fragment requests of this X slot
                    for f in range(XSLOTS + 4):
                        if self.x_issue_slot(f) == x:
                            if f >= XSLOTS:
                                self.read_x(f - XSLOTS, True, f"{nxt}.x{f - XSLOTS}")
                            else:
                                self.read_x(f, False, f"{tag}.x{f}")
                if nt == 1 or W == 1:
                    if x == 0:
                        self.word_loads(par)               # words + scales of step t + 2
                        self.advance_words()
                    if x % 4 == 0 and x < 28:
                        self.x_piece(1 + x // 4)           # pieces 1..7 of the X step at S_XK (t + 3)
                    if x == 1:
                        e(f"s_waitcnt vmcnt({3 * W + 9})", "words + scales of step t + 1 (requested one step ago) are in registers")
                    if x == 26:
                        # from the barrier on the pieces belong to step t + 4, in the stage this step has finished with
                        e(f"s_add_u32 s{S_XK}, s{S_XK}, 128")
                        e(f"s_add_u32 s{S_M0X}, s{S_M0X}, {STAGE}")
                        e(f"s_cmp_ge_u32 s{S_M0X}, {RING}")
                        e(f"s_cselect_b32 s{S_T}, {RING}, 0")
                        e(f"s_sub_u32 s{S_M0X}, s{S_M0X}, s{S_T}")
                    if x == 28:
                        self.x_piece(0)
                    if x == 30:
                        # the fragment bases move on one stage (the next step's requests in X slots 28-31 have been issued)
                        pass
            if x == 31:
                # rotate the read bases: current <- next, next <- next + STAGE (mod ring)
                e(f"v_mov_b32 v{V_XLO}, v{V_XLON}")
                e(f"v_mov_b32 v{V_XHI}, v{V_XHIN}")
                e(f"v_add_u32 v{V_XLON}, {STAGE}, v{V_XLON}")
                e(f"v_add_u32 v{V_XHIN}, {STAGE}, v{V_XHIN}")
                e(f"v_cmp_le_u32 vcc, {RING}, v{V_XLON}")
                e(f"v_subrev_u32 v{V_T}, {RING}, v{V_XLON}")
                e(f"v_cndmask_b32 v{V_XLON}, v{V_XLON}, v{V_T}, vcc")
                e(f"v_subrev_u32 v{V_T}, {RING}, v{V_XHIN}")
                e(f"v_cndmask_b32 v{V_XHIN}, v{V_XHIN}, v{V_T}, vcc")
        assert dqi == len(dq), (dqi, len(dq))

    # ---- whole kernel ---------------------------------------------------------------------------------------------------------------------
    def tile_coords(self) -> None:
        e = self.e
        T = S_T
        tile_n = 64 * self.W
        e(f"s_and_b32 s{T}, s{S_WG}, 7", "xcd")
        e(f"s_lshr_b32 s{T + 1}, s{S_WG}, 3")
        e(f"s_lshr_b32 s{T + 2}, s{S_NWG}, 3")
        e(f"s_and_b32 s{T + 3}, s{S_NWG}, 7")
        e(f"s_mul_i32 s{T + 4}, s{T}, s{T + 2}")
        e(f"s_min_u32 s{T + 5}, s{T}, s{T + 3}")
        e(f"s_add_u32 s{T + 4}, s{T + 4}, s{T + 5}")
        e(f"s_add_u32 s{T + 4}, s{T + 4}, s{T + 1}", "lin")
        e(f"s_mul_hi_u32 s{T + 5}, s{T + 4}, s{S_MAGIC_PG}", "group = lin / (8 tiles_n)")
        e(f"s_lshl_b32 s{T + 6}, s{S_TN}, 3")
        e(f"s_mul_i32 s{T + 7}, s{T + 5}, s{T + 6}")
        e(f"s_sub_u32 s{T + 7}, s{T + 4}, s{T + 7}", "in_group")
        e(f"s_lshl_b32 s{T + 8}, s{T + 5}, 3", "first_m")
        e(f"s_sub_u32 s{T + 9}, s{S_TM}, s{T + 8}")
        e(f"s_min_u32 s{T + 9}, s{T + 9}, 8", "group height")
        e(f"s_cmp_eq_u32 s{T + 9}, 8")
        e(f"s_cbranch_scc0 {self.L('partial_group')}")
        e(f"s_lshr_b32 s{T + 10}, s{T + 7}, 3")
        e(f"s_and_b32 s{T + 11}, s{T + 7}, 7")
        e(f"s_branch {self.L('tile_done')}")
        self.label(self.L("partial_group"))
        e(f"s_mul_hi_u32 s{T + 10}, s{T + 7}, s{S_MAGIC_LAST}")
        e(f"s_cmp_eq_u32 s{T + 9}, 1")
        e(f"s_cselect_b32 s{T + 10}, s{T + 7}, s{T + 10}")
        e(f"s_mul_i32 s{T + 11}, s{T + 10}, s{T + 9}")
        e(f"s_sub_u32 s{T + 11}, s{T + 7}, s{T + 11}")
        self.label(self.L("tile_done"))
        e(f"s_add_u32 s{T + 11}, s{T + 11}, s{T + 8}", "tm")
        e(f"s_lshl_b32 s{S_BM0}, s{T + 11}, 8")
        e(f"s_mul_i32 s{S_BN0}, s{T + 10}, {tile_n}")

    def prologue(self) -> None:
        e = self.e
        T = S_T
        W = self.W
        e(f"s_load_dwordx8 s[8:15], s[{S_KARG}:{S_KARG + 1}], 0x0")
        e(f"s_load_dwordx8 s[16:23], s[{S_KARG}:{S_KARG + 1}], 0x20")
        e(f"s_load_dwordx8 s[24:31], s[{S_KARG}:{S_KARG + 1}], 0x40")
        self.stamp(0)
        vl = V_T
        e(f"v_and_b32 v{vl}, 63, v{V_TID}", "lane")
        e(f"v_lshrrev_b32 v{V_T + 1}, 6, v{V_TID}")
        e("s_nop 0")
        e(f"v_readfirstlane_b32 s{S_WAVE}, v{V_T + 1}")
        e("s_waitcnt lgkmcnt(0)")
        self.tile_coords()
        # ---- descriptors ----
        e(f"s_sub_u32 s{T}, s{S_M}, 1")
        e(f"s_mul_i32 s{T}, s{T}, s{S_LDX}")
        e(f"s_add_u32 s{T}, s{T}, s{S_K}")
        e(f"s_lshl_b32 s{S_NRX}, s{T}, 1", "X bytes")
        e(f"s_lshr_b32 s{T + 1}, s{S_K}, 3")
        e(f"s_sub_u32 s{T + 1}, s{T + 1}, 1")
        e(f"s_mul_i32 s{T + 1}, s{T + 1}, s{S_LDQ}")
        e(f"s_add_u32 s{T + 1}, s{T + 1}, s{S_N}")
        e(f"s_lshl_b32 s{T + 1}, s{T + 1}, 2", "packed-word bytes")
        e(f"s_lshr_b32 s{T + 2}, s{S_K}, 7")
        e(f"s_sub_u32 s{T + 2}, s{T + 2}, 1")
        e(f"s_mul_i32 s{T + 2}, s{T + 2}, s{S_LDS}")
        e(f"s_add_u32 s{T + 2}, s{T + 2}, s{S_N}")
        e(f"s_lshl_b32 s{T + 2}, s{T + 2}, 1", "scale bytes")
        e(f"s_sub_u32 s{T + 3}, s{S_M}, 1")
        e(f"s_mul_i32 s{T + 3}, s{T + 3}, s{S_LDC}")
        e(f"s_add_u32 s{T + 3}, s{T + 3}, s{S_N}")
        e(f"s_lshl_b32 s{T + 3}, s{T + 3}, 1", "C bytes")
        for d, p, nr in ((S_DX, S_X, f"s{S_NRX}"), (S_DQ, S_WQ, f"s{T + 1}"), (S_DS, S_WS, f"s{T + 2}"), (S_DC, S_C, f"s{T + 3}")):
            e(f"s_mov_b32 s{d}, s{p}")
            e(f"s_and_b32 s{d + 1}, s{p + 1}, 0xffff")
            e(f"s_mov_b32 s{d + 2}, {nr}")
            e(f"s_mov_b32 s{d + 3}, 0x00020000")
        # ---- constants ----
        e(f"s_mov_b32 s{S_MAGIC}, 0x64006400")
        e(f"s_mov_b32 s{S_K16}, 0x2c002c00", "0.0625 | 0.0625")
        e(f"s_mov_b32 s{S_PLO}, 0x05040100")
        e(f"s_mov_b32 s{S_PHI}, 0x07060302")
        e(f"v_mov_b32 v{V_MASK}, 0x000f000f")
        e(f"v_mov_b32 v{V_MASKHI}, 0x00f000f0")
        # sub = half(1024 + off) twice; nsub16 = half(960) - sub = -(64 + off)
        e(f"s_add_u32 s{T}, s{S_OFF}, 1024")
        e(f"v_cvt_f32_i32 v{V_T + 1}, s{T}")
        e(f"v_cvt_f16_f32 v{V_T + 1}, v{V_T + 1}")
        e(f"v_and_b32 v{V_T + 1}, 0xffff, v{V_T + 1}")
        e(f"v_lshl_or_b32 v{V_SUB}, v{V_T + 1}, 16, v{V_T + 1}")
        e(f"v_mov_b32 v{V_T + 2}, 0x63806380")
        e(f"v_pk_add_f16 v{V_NSUB16}, v{V_T + 2}, v{V_SUB} neg_lo:[0,1] neg_hi:[0,1]")
        e(f"s_lshl_b32 s{S_WSTEP}, s{S_LDQ}, 5", "8 word rows x 4 bytes")
        e(f"s_lshl_b32 s{S_WHALF}, s{S_LDQ}, 4")
        e(f"s_lshl_b32 s{S_SSTEP}, s{S_LDS}, 1")
        # ---- X pieces: lane -> row 8 (4 p + wave) + (lane >> 3), swizzled source chunk ----
        v3, vc, vt = V_T + 1, V_T + 2, V_T + 3
        e(f"v_lshrrev_b32 v{v3}, 3, v{vl}")
        e(f"v_and_b32 v{vc}, 7, v{vl}")
        e(f"v_lshrrev_b32 v{vt}, 1, v{v3}")
        e(f"s_and_b32 s{T}, s{S_WAVE}, 1")
        e(f"s_lshl_b32 s{T}, s{T}, 2")
        e(f"v_add_u32 v{vt}, s{T}, v{vt}", "swizzle = 4 (wave & 1) + (lane >> 4)")
        e(f"v_xor_b32 v{vc}, v{vc}, v{vt}")
        e(f"v_lshlrev_b32 v{vc}, 4, v{vc}")
        e(f"s_lshl_b32 s{T}, s{S_WAVE}, 3")
        e(f"s_add_u32 s{T}, s{T}, s{S_BM0}")
        e(f"v_add_u32 v{v3}, s{T}, v{v3}", "row of piece 0")
        e(f"s_sub_u32 s{T + 1}, s{S_M}, 1")
        e(f"s_lshl_b32 s{T + 2}, s{S_LDX}, 1")
        for p in range(8):
            e(f"v_add_u32 v{vt}, {32 * p}, v{v3}")
            e(f"v_min_u32 v{vt}, s{T + 1}, v{vt}")
            e(f"v_mul_lo_u32 v{vt}, v{vt}, s{T + 2}")
            e(f"v_add_u32 v{V_XDMA + p}, v{vt}, v{vc}")
        # ---- weights: lane (r, g) -> word row g, column bn0 + 16 W wave + r (clamped) ----
        r, g = V_T + 1, V_T + 2
        e(f"v_and_b32 v{r}, 15, v{vl}")
        e(f"v_lshrrev_b32 v{g}, 4, v{vl}")
        e(f"s_mul_i32 s{T}, s{S_WAVE}, {16 * W}")
        e(f"s_add_u32 s{T}, s{T}, s{S_BN0}", "first column of this wave")
        e(f"v_add_u32 v{vt}, s{T}, v{r}", "(no clamp: columns past N read the next row or, past the tensor, the descriptor's zeros -- n tiles past N are never stored)")
        e(f"v_lshlrev_b32 v{V_SCOFF}, 1, v{vt}")
        e(f"v_mul_lo_u32 v{V_WOFF}, v{g}, s{S_LDQ}")
        e(f"v_add_u32 v{V_WOFF}, v{V_WOFF}, v{vt}")
        e(f"v_lshlrev_b32 v{V_WOFF}, 2, v{V_WOFF}")
        # ---- X fragment read bases ----
        lo = V_T + 3
        e(f"v_lshrrev_b32 v{lo}, 3, v{r}")
        e(f"v_lshlrev_b32 v{lo}, 10, v{lo}")
        e(f"v_and_b32 v{V_T + 4}, 7, v{r}")
        e(f"v_lshlrev_b32 v{V_T + 4}, 7, v{V_T + 4}")
        e(f"v_add_u32 v{lo}, v{lo}, v{V_T + 4}")
        e(f"v_lshrrev_b32 v{V_T + 4}, 1, v{r}")
        e(f"v_and_b32 v{V_T + 4}, 7, v{V_T + 4}")
        e(f"v_xor_b32 v{V_T + 4}, v{V_T + 4}, v{g}")
        e(f"v_lshlrev_b32 v{V_T + 4}, 4, v{V_T + 4}")
        e(f"v_add_u32 v{V_XLO}, v{lo}, v{V_T + 4}")
        e(f"v_xor_b32 v{V_XHI}, 64, v{V_XLO}")
        e(f"v_add_u32 v{V_XLON}, {STAGE}, v{V_XLO}")
        e(f"v_add_u32 v{V_XHIN}, {STAGE}, v{V_XHI}")
        # ---- first loads: X of steps 0-2 and piece 0 of step 3; words + scales of steps 0, 1 ----
        e(f"s_lshl_b32 s{S_M0X}, s{S_WAVE}, 10")
        e(f"s_mov_b32 s{S_XK}, 0")
        e("s_nop 3")
        for st in range(3):
            for p in range(8):
                self.x_piece(p)
            e(f"s_add_u32 s{S_XK}, s{S_XK}, 128")
            e(f"s_add_u32 s{S_M0X}, s{S_M0X}, {STAGE}")
            e("s_nop 3")
        self.x_piece(0)
        e(f"s_mov_b32 s{S_STEP}, 0")
        e(f"s_mov_b32 s{S_WOFF0}, 0")
        e(f"s_mov_b32 s{S_WOFF1}, s{S_WHALF}")
        e(f"s_mov_b32 s{S_SOFF}, 0")
        e("s_nop 3")
        self.word_loads(0)
        self.advance_words()
        e("s_nop 3")
        self.word_loads(1)
        self.advance_words()
        # accumulators
        for k in range(self.nacc):
            e(f"v_accvgpr_write_b32 a{k}, 0")
        e("s_waitcnt vmcnt(0)", "everything requested so far has landed: the loop's counted waits only ever wait for younger operations")
        for op in self.dequant_stream(0):
            e(op)
        e(f"s_lshr_b32 s{S_CNT}, s{S_K}, 7", "pairs of 64-element K steps")
        e("s_barrier")
        # the fragments a step finds requested on entry: 0-3 of its own stage
        for f in range(4):
            self.read_x(f, False, f"t0.x{f}")

    def loop(self) -> None:
        e = self.e
        self.label(self.L("loop"))
        entry = list(self.lgkm)
        self.step(0, "t0", "t1")
        self.step(1, "t1", "t0")
        assert self.lgkm == entry, (self.lgkm, entry)
        e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
        e(f"s_cmp_lg_u32 s{S_CNT}, 0")
        e(f"s_cbranch_scc1 {self.L('loop')}")

    def epilogue(self) -> None:
        """Cast the fp32 sums to fp16 (one rounding, as the reference's `accumulator.to(out)`) and store: 8 bytes per lane, rows past M
        and n tiles past N masked."""
        e = self.e
        T = S_T
        W = self.W
        e("s_waitcnt vmcnt(0)")
        e("s_waitcnt lgkmcnt(0)")
        e("s_nop 15")
        e("s_nop 7")
        vl, r, g, vb, x = self.v_e, self.v_e + 1, self.v_e + 2, self.v_e + 3, self.v_e + 4
        big = self.v_e + 12
        e(f"v_and_b32 v{vl}, 63, v{V_TID}")
        e(f"v_and_b32 v{r}, 15, v{vl}")
        e(f"v_lshrrev_b32 v{g}, 4, v{vl}")
        e(f"s_lshl_b32 s{T + 2}, s{S_LDC}, 1", "C row pitch in bytes")
        e(f"v_add_u32 v{vb}, s{S_BM0}, v{r}", "row of m tile 0")
        e(f"v_mul_lo_u32 v{vb}, v{vb}, s{T + 2}")
        e(f"s_mul_i32 s{T}, s{S_WAVE}, {16 * W}")
        e(f"s_add_u32 s{T}, s{T}, s{S_BN0}", "first column of this wave")
        e(f"v_lshlrev_b32 v{x}, 2, v{g}")
        e(f"v_add_u32 v{x}, s{T}, v{x}")
        e(f"v_lshl_add_u32 v{vb}, v{x}, 1, v{vb}", "+ column bytes")
        e(f"v_mov_b32 v{big}, 0x80000000")
        e(f"s_lshl_b32 s{T + 3}, s{T + 2}, 4", "16 rows of C in bytes")
        e(f"s_sub_u32 s{T + 4}, s{S_M}, s{S_BM0}", "rows of this tile inside M")
        # n tiles of this wave that start below N
        e(f"s_sub_u32 s{T + 5}, s{S_N}, s{T}")
        e(f"s_cmp_gt_i32 s{T + 5}, 0")
        e(f"s_cbranch_scc0 {self.L('end')}")
        e(f"s_lshr_b32 s{T + 5}, s{T + 5}, 4")
        e(f"s_mov_b32 s{T + 6}, 0", "scalar offset: 16 i rows")
        for i in range(MT):
            e(f"v_add_u32 v{x + 6}, {16 * i}, v{r}")
            e(f"v_cmp_gt_u32 vcc, s{T + 4}, v{x + 6}", "row inside M")
            e(f"v_cndmask_b32 v{x + 7}, v{big}, v{vb}, vcc")
            for nt in range(W):
                e(f"s_cmp_gt_u32 s{T + 5}, {nt}")
                e(f"s_cbranch_scc0 {self.L(f'skip_{i}_{nt}')}")
                t = i * W + nt
                for k in range(4):
                    e(f"v_accvgpr_read_b32 v{x + k}, a{4 * t + k}")
                for k in range(4):
                    e(f"v_cvt_f16_f32 v{x + k}, v{x + k}")
                e(f"v_pack_b32_f16 v{x + 4}, v{x}, v{x + 1}")
                e(f"v_pack_b32_f16 v{x + 5}, v{x + 2}, v{x + 3}")
                e(f"buffer_store_dwordx2 v[{x + 4}:{x + 5}], v{x + 7}, s[{S_DC}:{S_DC + 3}], s{T + 6} offen offset:{32 * nt}")
                e("s_nop 1")
                self.label(self.L(f"skip_{i}_{nt}"))
            e(f"s_add_u32 s{T + 6}, s{T + 6}, s{T + 3}")
        self.label(self.L("end"))
        self.stamp(2)
        if self.probe:
            e("s_waitcnt vmcnt(0)")
            self.stamp(3)
            e("s_waitcnt lgkmcnt(0)")
            e(f"s_cmp_lg_u32 s{S_WAVE}, 0")
            e(f"s_cbranch_scc1 {self.L('noprobe')}")
            e(f"s_cmp_eq_u64 s[{S_PROBE}:{S_PROBE + 1}], 0")
            e(f"s_cbranch_scc1 {self.L('noprobe')}")
            e(f"s_mul_i32 s{T}, s{S_WG}, {16 * NSTAMPS}")
            e(f"s_add_u32 s{T + 2}, s{S_PROBE}, s{T}")
            e(f"s_addc_u32 s{T + 3}, s{S_PROBE + 1}, 0")
            for k in range(NSTAMPS):
                e(f"s_store_dwordx4 s[{S_STAMP + 4 * k}:{S_STAMP + 4 * k + 3}], s[{T + 2}:{T + 3}], {16 * k}")
            e("s_waitcnt lgkmcnt(0)")
            e("s_dcache_wb")
            self.label(self.L("noprobe"))
        e("s_endpgm")

    def build(self) -> str:
        self.prologue()
        self.stamp(1)
        self.loop()
        self.epilogue()
        return self.render()

    def render(self) -> str:
        name = self.name
        head = f"""\t.text
\t.globl\t{name}
\t.p2align\t8
\t.type\t{name},@function
{name}:
"""
        tail = f"""
.Lfunc_end_{name}:
\t.size\t{name}, .Lfunc_end_{name}-{name}

\t.rodata
\t.p2align\t6, 0x0
\t.amdhsa_kernel {name}
\t\t.amdhsa_group_segment_fixed_size {RING}
\t\t.amdhsa_private_segment_fixed_size 0
\t\t.amdhsa_kernarg_size 96
\t\t.amdhsa_user_sgpr_count 2
\t\t.amdhsa_user_sgpr_dispatch_ptr 0
\t\t.amdhsa_user_sgpr_queue_ptr 0
\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1
\t\t.amdhsa_user_sgpr_dispatch_id 0
\t\t.amdhsa_user_sgpr_kernarg_preload_length 0
\t\t.amdhsa_user_sgpr_kernarg_preload_offset 0
\t\t.amdhsa_user_sgpr_private_segment_size 0
\t\t.amdhsa_uses_dynamic_stack 0
\t\t.amdhsa_enable_private_segment 0
\t\t.amdhsa_system_sgpr_workgroup_id_x 1
\t\t.amdhsa_system_sgpr_workgroup_id_y 0
\t\t.amdhsa_system_sgpr_workgroup_id_z 0
\t\t.amdhsa_system_sgpr_workgroup_info 0
\t\t.amdhsa_system_vgpr_workitem_id 0
\t\t.amdhsa_next_free_vgpr 512
\t\t.amdhsa_next_free_sgpr {S_END + 1}
\t\t.amdhsa_accum_offset 256
\t\t.amdhsa_reserve_vcc 1
\t\t.amdhsa_float_round_mode_32 0
\t\t.amdhsa_float_round_mode_16_64 0
\t\t.amdhsa_float_denorm_mode_32 3
\t\t.amdhsa_float_denorm_mode_16_64 3
\t\t.amdhsa_dx10_clamp 1
\t\t.amdhsa_ieee_mode 1
\t\t.amdhsa_fp16_overflow 0
\t\t.amdhsa_tg_split 0
\t\t.amdhsa_exception_fp_ieee_invalid_op 0
\t\t.amdhsa_exception_fp_denorm_src 0
\t\t.amdhsa_exception_fp_ieee_div_zero 0
\t\t.amdhsa_exception_fp_ieee_overflow 0
\t\t.amdhsa_exception_fp_ieee_underflow 0
\t\t.amdhsa_exception_fp_ieee_inexact 0
\t\t.amdhsa_exception_int_div_zero 0
\t.end_amdhsa_kernel
\t.text
"""
        return head + "\n".join(self.lines) + tail


def metadata(names: list[str]) -> str:
    kernels = ""
    for name in names:
        kernels += f"""  - .agpr_count:     256
    .args:
      - .offset:         0
        .size:           96
        .value_kind:     by_value
    .group_segment_fixed_size: {RING}
    .kernarg_segment_align: 8
    .kernarg_segment_size: 96
    .language:       OpenCL C
    .language_version:
      - 2
      - 0
    .max_flat_workgroup_size: 256
    .name:           {name}
    .private_segment_fixed_size: 0
    .sgpr_count:     {S_END + 7}
    .sgpr_spill_count: 0
    .symbol:         {name}.kd
    .uniform_work_group_size: 1
    .uses_dynamic_stack: false
    .vgpr_count:     512
    .vgpr_spill_count: 0
    .wavefront_size: 64
"""
    return f"""\t.amdgpu_metadata
---
amdhsa.kernels:
{kernels}amdhsa.target:   amdgcn-amd-amdhsa--gfx950
amdhsa.version:
  - 1
  - 2
...

\t.end_amdgpu_metadata
"""


def main() -> None:
    out = sys.argv[1]
    text = '\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"\n\t.amdhsa_code_object_version 6\n'
    names = []
    for name, w, probe in (("conch_mixed1w_f16_i4_w3", 3, False), ("conch_mixed1w_f16_i4_w4", 4, False),
                           ("conch_mixed1w_f16_i4_w3_probe", 3, True), ("conch_mixed1w_f16_i4_w4_probe", 4, True)):
        text += Gen(name, w, probe).build()
        names.append(name)
    text += metadata(names)
    with open(out, "w") as f:
        f.write(text)


if __name__ == "__main__":
    main()
