#!/usr/bin/env python3
"""Generator of the ONE-WAVE-PER-SIMD fp8 scaled-GEMM kernel for gfx950 (hand-allocated registers; writes an assembly file).

Why it exists (scaled GEMM): the 256 x 256 tile of gemm_mfma.hip (eight waves, two per SIMD, 128 x 64 per wave) is
what 256 registers per wave allow.  BASELINE config C3 (4096 x 4096 x 11008) is 688 such tiles on 256 CUs = 2.69 rounds run as
three, with three epilogues per CU, 24 KiB of fragment reads per 32 MFMAs and 64 KiB of L2 -> LDS traffic per 64 MFMAs of a SIMD.
With ONE wave per SIMD a wave may use all 512 registers of its SIMD (256 architectural + 256 accumulation registers), which
hipcc cannot allocate without shuffling accumulators through v_accvgpr moves (profiles/r03/mfma_shape.txt) -- hence this file:

  * tile 256 rows x 352 columns per 256-thread workgroup: four waves as 2 (m) x 2 (n), 128 x 176 per wave = 8 x 11 accumulator
    tiles of 16 x 16 = 352 registers (tiles 0-63 in a[0:255], tiles 64-87 in v[160:255]).  C3 = 16 x 32 = 512 tiles = exactly
    TWO rounds on 256 CUs (97.7 % of the MFMAs are useful: the last tile column is 96 of 352 columns wide);
  * per 128-byte K step and wave: 88 MFMAs (v_mfma_f32_16x16x128_f8f6f4) on 8 + 11 fragments = 38 KiB of ds_read_b128 -- 42 %
    fewer fragment bytes per MFMA than the 128 x 64 wave tile -- and 76 KiB staged per step and workgroup for 352 MFMAs -- 14 %
    fewer L2 -> LDS bytes per MFMA;
  * LDS: two stages of (256 + 352) rows x 128 bytes = 152 KiB, the library's row image (8-row x 128-byte pieces written by
    `buffer_load_dwordx4 ... lds`, 16-byte chunks XOR-swizzled on the SOURCE side, conflict-free ds_read_b128);
  * the wave's instruction stream is one MFMA slot after another; fragment reads, LDS-DMA issues and scalar bookkeeping sit in
    the gaps (an MFMA occupies the matrix pipe for 32 cycles and the SIMD's issue port for 8).  A fragments (64 registers) are
    reloaded for the NEXT step under the last n tile of this one; B fragments stream through three 8-register slots one n tile
    ahead.  Two workgroup barriers per step: Z (after n tile 5's fragment is read: the A part and B tiles 0-5 of this stage are
    dead -> 14 DMA pieces of step t + 2 follow) and X (after n tile 10's: the rest is dead -> 5 pieces; and everything of step
    t + 1 has landed: `s_waitcnt vmcnt(14)` before it).  vmcnt retires in order, so the counts are exact.

The kernel takes M, N, K, leading dimensions and scale vectors at run time; what is fixed: fp8 e4m3 (OCP) operands, K-contiguous
A and B^T with 16-byte aligned rows, K a multiple of 256 bytes, bf16 or fp16 row-major C, N a multiple of 16; an optional bias vector
of the output dtype is added by the *_bias twins of every kernel (the launcher in gemm_asm.hip checks the contract; other problems
keep the HIP kernels).

usage: gen_gemm1w.py OUT.s
"""
from __future__ import annotations

import os
import sys

MT = 8                       # 16-row m tiles per wave
TILE_M = 256
A_BYTES = TILE_M * 128                       # 32 KiB of a stage
SCALE_BUF = 3072                             # two scale buffers: the current tile's, and the next tile's parked under the epilogue
BIAS_BUF = 1024                              # (bias variants) ... and two buffers of the tile's 352 bias values, 16 bits each


def configure(nt: int) -> None:
    """Everything that follows from NT = 16-column n tiles per wave (tile width 32 NT).  NT = 11 is the form described in the module
    docstring (256 x 352: C3 in exactly two rounds); NT = 9 / 7 (288 / 224 columns) serve shapes whose N the wider tile quantises
    badly (the C5 shard of 3584 columns = 16 x 224: two full rounds at M = 8192).  Numbers in comments below are NT = 11's."""
    global NT, TILE_N, B_WAVECOL, STAGE, LDS_SA, LDS_SB, LDS_TOTAL, LDS_BIAS, SLOTS, Z_SLOT, X_SLOT, PIECES_Z, PIECES_X, JZ
    assert 5 <= nt <= 11 and nt % 2 == 1, "odd: the last n tile walks the m tiles upwards, which the A-fragment reload relies on"
    NT = nt
    TILE_N = 2 * 16 * NT                     # 352
    B_WAVECOL = 16 * NT * 128                # 22528: one wave column's rows of a stage
    STAGE = A_BYTES + 2 * B_WAVECOL          # 77824
    LDS_SA = 2 * STAGE                       # 155648: float sa[256] of scale buffer 0
    LDS_SB = LDS_SA + 1024                   # float sb[512] (352 used)
    LDS_TOTAL = LDS_SA + 2 * SCALE_BUF       # 161792 of 163840
    LDS_BIAS = LDS_TOTAL                     # bias variants: two buffers of 512 16-bit values behind the scale buffers (163840 in all)
    SLOTS = MT * NT                          # 88 MFMAs per step and wave
    JZ = NT // 2                             # 5: barrier Z sits behind the fragment read of B tile JZ
    Z_SLOT = 8 * (JZ - 1) + 4                # 36: barriers in FRONT of this slot's MFMA
    X_SLOT = 8 * (NT - 2) + 4                # 76
    PIECES_Z = 8 + JZ + 1                    # 14: LDS-DMA pieces issued behind each barrier (8 A + B tiles 0-5 | B tiles 6-10)
    PIECES_X = NT - JZ - 1                   # 5


configure(11)

# ---- register map ------------------------------------------------------------------------------------------------------------
# SGPRs
S_KARG = 0          # s[0:1]
S_WG = 2
S_A, S_B, S_C, S_SA = 8, 10, 12, 14          # pointers (pairs) as loaded
S_SB = 16
S_M, S_N, S_K, S_LDA, S_LDB, S_LDC = 18, 19, 20, 21, 22, 23
S_TM, S_TN, S_NWG, S_MAGIC_PG, S_GRID, S_MAGIC_LAST, S_SAVEC, S_SBVEC = 24, 25, 26, 27, 28, 29, 30, 31
S_ACCSCALE, S_OUTDT = 32, 33
S_T = 34            # s34..s47 temporaries
S_WAVE, S_WR, S_WC = 48, 49, 50
S_BM0, S_BN0 = 51, 52
S_M0A, S_M0B = 53, 54                        # LDS-DMA destination bases of this wave's A / B pieces (stage 0)
S_KOFF = 55                                  # K byte offset of the step whose pieces are being issued
S_CNT = 56                                   # step pairs left
S_NRA, S_NRB = 57, 58                        # num_records of the operand descriptors
S_DA, S_DB, S_DC = 60, 64, 68                # buffer descriptors s[60:63], s[64:67], s[68:71]
S_NVALID = 72                                # n tiles of this wave that start below N
S_PROBE = 74                                 # s[74:75]: debug buffer of the diagnostic build (kernarg offset 104; 0 = none)
S_BIAS = 74                                  # ... in the bias variants (never diagnostic builds): the bias vector, same kernarg slot
S_STAMP = 76                                 # s[76:76+4*NSTAMPS): (s_memtime, s_memrealtime) pairs of the diagnostic build
NSTAMPS = 5
S_END = S_STAMP + 4 * NSTAMPS                # 96 (s0..s101 exist)
S_TILE = 3                                   # tile of this pass of the persistent walk (= virtual workgroup id)
S_TNEXT = 4                                  # tile + grid
S_BM0N, S_BN0N = 5, 6                        # first row / column of the NEXT tile
S_SCOFF = 7                                  # LDS offset of the current tile's scale buffer (0 or SCALE_BUF)
S_HASNEXT = 59                               # 1 = another tile follows

# VGPRs
V_TID = 0
V_DMA = 1           # v1..v19: per-lane source byte offsets of the wave's 8 A + 11 B pieces
V_ALO = (20, 21)    # fragment read bases, stage 0 / 1, chunk g
V_AHI = (22, 23)    # ... chunk g + 4
V_BLO = (24, 25)
V_BHI = (26, 27)
V_T = 28            # v28..v31 temporaries of the prologue
V_FA = 32           # v32..v95: A fragments of the 8 m tiles (8 registers each)
V_FB = 96           # v96..v119: three B fragment slots
V_E = 120           # v120..v151: epilogue temporaries
V_NSC = 152         # v152..154: the NEXT tile's sa / sb / sb + 256 values, loaded before the epilogue, parked in LDS after it
V_NBIAS = 157       # v157..158 (bias variants): the NEXT tile's bias / bias + 256 values (16 bits each), parked the same way
V_L3 = 155          # lane >> 3                       } per-lane constants of the LDS-DMA source offsets
V_CH16 = 156        # swizzled source chunk * 16      }
V_ACC = 160         # v160..v255: accumulator tiles 64..87
A_ACC = 0           # a0..a255: accumulator tiles 0..63


def fb_slot(j: int) -> int:
    return 2 if j == NT - 1 else j % 2


def acc_reg(i: int, j: int) -> str:
    t = 8 * j + i
    return f"a[{4 * t}:{4 * t + 3}]" if t < 64 else f"v[{V_ACC + 4 * (t - 64)}:{V_ACC + 4 * (t - 64) + 3}]"


class Gen:
    def __init__(self, name: str, out_fp16: bool = False, probe: bool = False, rows: bool = True, pk: bool = True, int8: bool = False, chains: int = 4, store_policy: str = "",
                 fused_tail: bool = False, bias: bool = False):
        self.name = name
        # bias variant: out = cast(sb * (sa * acc)) + bias, the sum formed in fp32 from the two 16-bit values and rounded to the
        # output dtype (conch/reference/quantization/scaled_gemm.py:24-25: `out + bias` on tensors of the output dtype) -- the
        # same operations, element for element, as the HIP kernels' epilogue (gemm_mfma.hip)
        self.bias = bias
        assert not (bias and (probe or fused_tail))
        self.lds_total = LDS_TOTAL + (2 * BIAS_BUF if bias else 0)
        self.probe = probe
        self.lines: list[str] = []
        self.lgkm: list[str] = []        # outstanding LDS reads, oldest first (tags)
        self.out_fp16 = out_fp16
        self.rows = rows
        self.pk = pk
        self.chains = chains
        self.store_policy = store_policy   # cache-policy suffix of the C stores ("" | " sc1" | " nt")
        self.int8 = int8          # int8 operands: 2 x v_mfma_i32_16x16x64_i8 per tile and step, int32 accumulators
        # TIMING EXPERIMENT (the A/B twin only; WRONG results): the tile's conversion and stores ride in its LAST K step (8-byte
        # stores, lanes regrouped by ds_bpermute so that a quad covers one 32-byte row piece), the last two steps fetch the NEXT
        # tile's first two stages, and the separate epilogue disappears -- what would that buy?  (profiles/r05/asm1w_fused_tail_timing.txt)
        self.fused_tail = fused_tail

    def L(self, s: str) -> str:
        return f".L{self.name}_{s}"

    # -- emission ------------------------------------------------------------------------------------------------------------
    def e(self, s: str, comment: str = "") -> None:
        self.lines.append(f"\t{s}" + (f"\t; {comment}" if comment else ""))

    def label(self, s: str) -> None:
        self.lines.append(f"{s}:")

    def c(self, s: str) -> None:
        self.lines.append(f"\t; {s}")

    def stamp(self, k: int) -> None:
        """Diagnostic build only: shader clock and the constant 100 MHz clock at this point, kept in SGPRs until the end."""
        if self.probe:
            assert k < NSTAMPS
            self.e(f"s_memtime s[{S_STAMP + 4 * k}:{S_STAMP + 4 * k + 1}]")
            self.e(f"s_memrealtime s[{S_STAMP + 4 * k + 2}:{S_STAMP + 4 * k + 3}]")

    def write_stamps(self) -> None:
        """Wave 0 of every workgroup stores its stamps to probe[tile][NSTAMPS][2] (64-bit each) with scalar stores."""
        if not self.probe:
            return
        e = self.e
        T = S_T
        e("s_waitcnt lgkmcnt(0)")
        e(f"s_cmp_lg_u32 s{S_WAVE}, 0")
        e(f"s_cbranch_scc1 {self.L('noprobe')}")
        e(f"s_cmp_eq_u64 s[{S_PROBE}:{S_PROBE + 1}], 0")
        e(f"s_cbranch_scc1 {self.L('noprobe')}")
        e(f"s_mul_i32 s{T}, s{S_TILE}, {16 * NSTAMPS}")
        e(f"s_add_u32 s{T + 2}, s{S_PROBE}, s{T}")
        e(f"s_addc_u32 s{T + 3}, s{S_PROBE + 1}, 0")
        for k in range(NSTAMPS):
            e(f"s_store_dwordx4 s[{S_STAMP + 4 * k}:{S_STAMP + 4 * k + 3}], s[{T + 2}:{T + 3}], {16 * k}")
        e("s_waitcnt lgkmcnt(0)")
        e("s_dcache_wb")
        self.label(self.L("noprobe"))

    # -- LDS read tracking (lgkmcnt retires LDS operations in order) ---------------------------------------------------------
    def ds_read128(self, dst: int, addr: int, off: int, tag: str) -> None:
        assert 0 <= off < 65536, off
        self.e(f"ds_read_b128 v[{dst}:{dst + 3}], v{addr} offset:{off}", tag)
        self.lgkm.append(tag)

    def wait_tags(self, tags: list[str]) -> None:
        """s_waitcnt lgkmcnt(n) so that every read in `tags` has returned (no instruction if none is outstanding)."""
        idx = max((i for i, t in enumerate(self.lgkm) if t in tags), default=-1)
        if idx < 0:
            return
        after = len(self.lgkm) - 1 - idx
        self.e(f"s_waitcnt lgkmcnt({min(after, 15)})")
        self.lgkm = self.lgkm[idx + 1:] if after <= 15 else self.lgkm[len(self.lgkm) - 15:]

    def wait_all_lds(self) -> None:
        self.e("s_waitcnt lgkmcnt(0)")
        self.lgkm = []

    # -- fragments -------------------------------------------------------------------------------------------------------------
    def read_fa(self, i: int, stage: int, step_tag: str) -> None:
        base = V_FA + 8 * i
        self.ds_read128(base, V_ALO[stage], 2048 * i, f"{step_tag}.A{i}.lo")
        self.ds_read128(base + 4, V_AHI[stage], 2048 * i, f"{step_tag}.A{i}.hi")

    def read_fb_half(self, j: int, stage: int, step_tag: str, half: int) -> None:
        base = V_FB + 8 * fb_slot(j)
        if half == 0:
            self.ds_read128(base, V_BLO[stage], 2048 * j, f"{step_tag}.B{j}.lo")
        else:
            self.ds_read128(base + 4, V_BHI[stage], 2048 * j, f"{step_tag}.B{j}.hi")

    # -- LDS-DMA ----------------------------------------------------------------------------------------------------------------
    def dma_m0(self, idx: int, stage: int) -> None:
        """M0 = LDS destination of piece idx of this wave (0-7: A rows 8 (4 idx + wave) .., 8-18: B n tile idx - 8) in `stage`."""
        if idx < 8:
            self.e(f"s_add_i32 m0, s{S_M0A}, {stage * STAGE + 4096 * idx}")
        else:
            self.e(f"s_add_i32 m0, s{S_M0B}, {stage * STAGE + 2048 * (idx - 8)}")

    def dma_load(self, idx: int, stage: int) -> None:
        """The LDS-DMA of piece idx of the step at S_KOFF (M0 set by dma_m0, at least one instruction earlier)."""
        if idx < 8:
            self.e(f"buffer_load_dwordx4 v{V_DMA + idx}, s[{S_DA}:{S_DA + 3}], s{S_KOFF} offen lds", f"A piece {idx} -> stage {stage}")
        else:
            self.e(f"buffer_load_dwordx4 v{V_DMA + idx}, s[{S_DB}:{S_DB + 3}], s{S_KOFF} offen lds", f"B tile {idx - 8} piece -> stage {stage}")

    def dma_piece(self, idx: int, stage: int) -> None:
        self.dma_m0(idx, stage)
        self.e("s_nop 0")
        self.dma_load(idx, stage)

    def issue_stage(self, stage: int, koff: int) -> None:
        """All 19 pieces of this wave for the K step at byte offset `koff` into `stage` (real descriptors)."""
        e = self.e
        e(f"s_mov_b32 s{S_KOFF}, {koff}")
        e(f"s_mov_b32 s{S_DA + 2}, s{S_NRA}")
        e(f"s_mov_b32 s{S_DB + 2}, s{S_NRB}")
        e("s_nop 3")
        for idx in range(8 + NT):
            self.dma_piece(idx, stage)

    # -- one K step -----------------------------------------------------------------------------------------------------------------
    def step(self, st: int, tag: str, nxt: str, first: bool = False, tail: int = 0) -> None:
        """88 MFMA slots of the step staged in `st`; `tag` / `nxt` name this step's and the next step's fragment reads.
        `first`: the tile's first K step -- the MFMAs take the constant 0 as their addend (no accumulator is zeroed anywhere)."""
        # DMA issue slots: after Z every third slot that carries no fragment read, after X every other one
        last = 8 * (NT - 1)                  # first slot of the last n tile (80)
        busy = {8 * j + h for j in range(NT) for h in (0, 1)} | {X_SLOT + 2, X_SLOT + 3} | set(range(last, SLOTS)) | {0}
        free = [q for q in range(Z_SLOT, X_SLOT) if q not in busy]
        z_slots = free[::max(1, len(free) // PIECES_Z)][:PIECES_Z]
        assert len(z_slots) == PIECES_Z, z_slots
        x_slots = ([X_SLOT, X_SLOT + 1] + list(range(last + 2, SLOTS, 2)) + list(range(last + 1, SLOTS, 2)))[:PIECES_X]   # 76, 77, 82, 84, 86
        assert len(x_slots) == PIECES_X and len(set(x_slots)) == PIECES_X
        dma_at = {q: k for k, q in enumerate(z_slots)}
        dma_at.update({q: PIECES_Z + k for k, q in enumerate(x_slots)})
        for q in range(SLOTS):
            j, ii = divmod(q, MT)
            i = ii if j % 2 == 0 else MT - 1 - ii
            if q == Z_SLOT:
                self.c("---- barrier Z: every wave has read this stage's A part and B tiles 0-5 ----")
                self.wait_all_lds()
                self.e("s_barrier")
            if q == X_SLOT:
                self.c("---- barrier X: the next stage has landed; this stage's B tiles 6-10 are dead ----")
                # (tail step of the timing experiment: the 8-byte stores issued so far are younger than the awaited pieces too)
                # ... and in the step before it the next tile's three scale loads (the experiment does not care that a walk's last
                # tile issues none; the real thing would issue them unconditionally)
                self.e(f"s_waitcnt vmcnt({min(63, PIECES_Z + (max(0, q - 4) if tail == 2 else 3 if tail == 1 else 0))})")
                self.wait_all_lds()
                self.e("s_barrier")
            self.wait_tags([f"{tag}.A{i}.lo", f"{tag}.A{i}.hi", f"{tag}.B{j}.lo", f"{tag}.B{j}.hi"])
            if q in dma_at:
                self.dma_m0(dma_at[q], st)     # (the MFMA below is the wait state between the M0 write and the LDS-DMA that reads it)
            fb = V_FB + 8 * fb_slot(j)
            fa = V_FA + 8 * i
            acc = acc_reg(i, j)
            add = "0" if first else acc
            if self.int8:
                self.e(f"v_mfma_i32_16x16x64_i8 {acc}, v[{fb}:{fb + 3}], v[{fa}:{fa + 3}], {add}", f"slot {q}: m tile {i}, n tile {j}, K bytes 0-63")
                self.e(f"v_mfma_i32_16x16x64_i8 {acc}, v[{fb + 4}:{fb + 7}], v[{fa + 4}:{fa + 7}], {acc}", "K bytes 64-127")
            else:
                self.e(f"v_mfma_f32_16x16x128_f8f6f4 {acc}, v[{fb}:{fb + 7}], v[{fa}:{fa + 7}], {add}", f"slot {q}: m tile {i}, n tile {j}")
            # ---- fillers behind this MFMA ----
            if q == 0:
                self.read_fa(7, st, tag)          # FA[7] of THIS step (its registers were busy until the previous step's last MFMA)
            if ii in (0, 1) and j + 1 <= NT - 2:
                self.read_fb_half(j + 1, st, tag, ii)
            if q in (8 * (NT - 2), 8 * (NT - 2) + 1):            # 72, 73
                self.read_fb_half(NT - 1, st, tag, q - 8 * (NT - 2))
            if q in (X_SLOT + 2, X_SLOT + 3):                    # 78, 79
                self.read_fb_half(0, 1 - st, nxt, q - (X_SLOT + 2))
            if j == NT - 1 and ii >= 1:
                self.read_fa(ii - 1, 1 - st, nxt)   # FA[ii - 1] was last used one slot ago
            if q in dma_at:
                self.dma_load(dma_at[q], st)       # order: A 0-7, B tiles 0-5 | B tiles 6-10
            if q == 2 and tail:
                # the last two steps of a tile (timing experiment): their pieces are the NEXT tile's steps 0 and 1 (v1..v19 already
                # hold its source offsets), or nothing when the walk ends here
                self.e(f"s_mov_b32 s{S_KOFF}, {128 * (tail - 1)}")
                self.e(f"s_cmp_lg_u32 s{S_HASNEXT}, 0")
                self.e(f"s_cselect_b32 s{S_DA + 2}, s{S_NRA}, 0")
                self.e(f"s_cselect_b32 s{S_DB + 2}, s{S_NRB}, 0")
            elif q == 2:
                # the step whose pieces this step issues is t + 2: past the end of K the descriptors are switched to zero
                # records (every lane out of range: no memory traffic, zeros land in a stage nobody reads)
                self.e(f"s_add_u32 s{S_KOFF}, s{S_KOFF}, 128")
                self.e(f"s_cmp_lt_u32 s{S_KOFF}, s{S_K}")
                self.e(f"s_cselect_b32 s{S_DA + 2}, s{S_NRA}, 0")
                self.e(f"s_cselect_b32 s{S_DB + 2}, s{S_NRB}, 0")
            if tail == 2:
                self.tail_convert(q)

    # -- timing experiment: conversion and stores inside the last K step ------------------------------------------------------------
    FT_X, FT_PK, FT_PM, FT_SA, FT_SB, FT_PERM, FT_VOFF = V_E, V_E + 8, V_E + 12, V_E + 18, V_E + 26, V_E + 30, V_E + 31

    def tail_prepare(self) -> None:
        """Before the last pair of steps: row-block store offsets s[S_T : S_T + 8), the lane regrouping and the per-lane store offset."""
        e = self.e
        T = S_T
        e(f"s_lshl_b32 s{T + 8}, s{S_WR}, 7")
        e(f"s_add_u32 s{T + 8}, s{T + 8}, s{S_BM0}", "first row of this wave")
        e(f"s_lshl_b32 s{T + 9}, s{S_LDC}, 1", "row pitch in bytes")
        e(f"s_mul_i32 s{T + 8}, s{T + 8}, s{T + 9}")
        e(f"s_mul_i32 s{T + 10}, s{S_WC}, {16 * NT}")
        e(f"s_add_u32 s{T + 10}, s{T + 10}, s{S_BN0}")
        e(f"s_lshl_b32 s{T + 10}, s{T + 10}, 1")
        e(f"s_add_u32 s{T + 8}, s{T + 8}, s{T + 10}", "byte offset of this wave's sub-tile in C")
        e(f"s_lshl_b32 s{T + 9}, s{T + 9}, 4", "16 rows")
        for i in range(MT):
            e(f"s_mov_b32 s{T + i}, s{T + 8}")
            e(f"s_add_u32 s{T + 8}, s{T + 8}, s{T + 9}")
        vl = V_T
        e(f"v_and_b32 v{vl}, 63, v{V_TID}")
        e(f"v_and_b32 v{self.FT_PERM}, 3, v{vl}", "g' = lane & 3")
        e(f"v_lshlrev_b32 v{self.FT_VOFF}, 3, v{self.FT_PERM}", "8 g' bytes")
        e(f"v_lshlrev_b32 v{self.FT_PERM}, 4, v{self.FT_PERM}")
        e(f"v_lshrrev_b32 v{vl}, 2, v{vl}", "c' = lane >> 2")
        e(f"v_add_u32 v{self.FT_PERM}, v{self.FT_PERM}, v{vl}", "source lane 16 g' + c'")
        e(f"v_lshlrev_b32 v{self.FT_PERM}, 2, v{self.FT_PERM}", "ds_bpermute address")
        e(f"s_lshl_b32 s{T + 9}, s{S_LDC}, 1")
        e(f"v_mul_lo_u32 v{vl}, v{vl}, s{T + 9}")
        e(f"v_add_u32 v{self.FT_VOFF}, v{self.FT_VOFF}, v{vl}", "c' rows + 8 g' bytes")
        for k in range(12):  # (stand-ins for the row / column scales)
            e(f"v_mov_b32 v{self.FT_SA + k}, 1.0")

    def tail_convert_rest(self, q: int) -> None:
        """slots past the end of the step: what tail_convert still owes"""
        self.tail_convert_parts(q, convert=True, store=True)

    def tail_convert(self, q: int) -> None:
        self.tail_convert_parts(q, convert=True, store=True)

    def tail_convert_parts(self, q: int, convert: bool, store: bool) -> None:
        """Fillers of slot q of a tile's last K step: convert the tile of slot q - 2, store the one of slot q - 4."""
        e = self.e
        if convert and 2 <= q < SLOTS + 2:
            c = q - 2
            j, ii = divmod(c, MT)
            i = ii if j % 2 == 0 else MT - 1 - ii
            t = 8 * j + i
            x, pk = self.FT_X + 4 * (c % 2), self.FT_PK + 2 * (c % 2)
            pm = self.FT_PM + 2 * (c % 3)
            if t < 64:
                for k in range(4):
                    e(f"v_accvgpr_read_b32 v{x + k}, a{4 * t + k}")
                src = x
            else:
                src = V_ACC + 4 * (t - 64)
            e(f"v_pk_mul_f32 v[{x}:{x + 1}], v[{src}:{src + 1}], v[{self.FT_SA}:{self.FT_SA + 1}]")
            e(f"v_pk_mul_f32 v[{x + 2}:{x + 3}], v[{src + 2}:{src + 3}], v[{self.FT_SA}:{self.FT_SA + 1}]")
            e(f"v_pk_mul_f32 v[{x}:{x + 1}], v[{x}:{x + 1}], v[{self.FT_SB}:{self.FT_SB + 1}]")
            e(f"v_pk_mul_f32 v[{x + 2}:{x + 3}], v[{x + 2}:{x + 3}], v[{self.FT_SB + 2}:{self.FT_SB + 3}]")
            e(f"v_cvt_pk_bf16_f32 v{pk}, v{x}, v{x + 1}")
            e(f"v_cvt_pk_bf16_f32 v{pk + 1}, v{x + 2}, v{x + 3}")
            for k in range(2):
                e(f"ds_bpermute_b32 v{pm + k}, v{self.FT_PERM}, v{pk + k}", f"ft.{c}.{k}")
                self.lgkm.append(f"ft.{c}.{k}")
        if store and 4 <= q < SLOTS + 4:
            c = q - 4
            j, ii = divmod(c, MT)
            i = ii if j % 2 == 0 else MT - 1 - ii
            pm = self.FT_PM + 2 * (c % 3)
            self.wait_tags([f"ft.{c}.0", f"ft.{c}.1"])
            e(f"buffer_store_dwordx2 v[{pm}:{pm + 1}], v{self.FT_VOFF}, s[{S_DC}:{S_DC + 3}], s{S_T + i} offen offset:{32 * j}" + self.store_policy)

    # -- whole kernel ---------------------------------------------------------------------------------------------------------------
    def setup_once(self) -> None:
        """Arguments, wave roles, descriptors, LDS-DMA destination bases, fragment read bases: everything that does not depend
        on the tile."""
        e = self.e
        T = S_T
        e(f"s_load_dwordx8 s[8:15], s[{S_KARG}:{S_KARG + 1}], 0x0")
        e(f"s_load_dwordx8 s[16:23], s[{S_KARG}:{S_KARG + 1}], 0x20")
        e(f"s_load_dwordx8 s[24:31], s[{S_KARG}:{S_KARG + 1}], 0x40")
        e(f"s_load_dwordx2 s[{S_ACCSCALE}:{S_ACCSCALE + 1}], s[{S_KARG}:{S_KARG + 1}], 0x60")
        if self.probe or self.bias:
            e(f"s_load_dwordx2 s[{S_PROBE}:{S_PROBE + 1}], s[{S_KARG}:{S_KARG + 1}], 0x68", "debug buffer | bias vector")
        # gated launch (the e4m3fnuz fast path, gemm.hpp ScaledGemmArgs::gate): run only if (*gate != 0) == (run_if != 0)
        e(f"s_load_dwordx2 s[{T}:{T + 1}], s[{S_KARG}:{S_KARG + 1}], 0x70")
        e(f"s_load_dword s{T + 2}, s[{S_KARG}:{S_KARG + 1}], 0x78")
        e("s_waitcnt lgkmcnt(0)")
        e(f"s_cmp_eq_u64 s[{T}:{T + 1}], 0")
        e(f"s_cbranch_scc1 {self.L('nogate')}")
        e(f"s_load_dword s{T + 3}, s[{T}:{T + 1}], 0x0")
        e("s_waitcnt lgkmcnt(0)")
        e(f"s_cmp_lg_u32 s{T + 3}, 0")
        e(f"s_cselect_b32 s{T + 3}, 1, 0")
        e(f"s_cmp_lg_u32 s{T + 2}, 0")
        e(f"s_cselect_b32 s{T + 2}, 1, 0")
        e(f"s_cmp_eq_u32 s{T + 2}, s{T + 3}")
        e(f"s_cbranch_scc0 {self.L('exit')}")
        self.label(self.L("nogate"))
        self.stamp(0)
        e(f"s_mov_b32 s{S_TILE}, s{S_WG}")
        e(f"s_mov_b32 s{S_SCOFF}, 0")
        # lane / wave
        vl = V_T
        e(f"v_and_b32 v{vl}, 63, v{V_TID}", "lane")
        e(f"v_lshrrev_b32 v{V_T + 1}, 6, v{V_TID}")
        e("s_nop 0")
        e(f"v_readfirstlane_b32 s{S_WAVE}, v{V_T + 1}")
        e(f"s_lshr_b32 s{S_WR}, s{S_WAVE}, 1")
        e(f"s_and_b32 s{S_WC}, s{S_WAVE}, 1")
        e("s_waitcnt lgkmcnt(0)")
        # ---- descriptors ----
        e(f"s_sub_u32 s{T}, s{S_M}, 1")
        e(f"s_mul_i32 s{T}, s{T}, s{S_LDA}")
        e(f"s_add_u32 s{S_NRA}, s{T}, s{S_K}", "A bytes")
        e(f"s_sub_u32 s{T}, s{S_N}, 1")
        e(f"s_mul_i32 s{T}, s{T}, s{S_LDB}")
        e(f"s_add_u32 s{S_NRB}, s{T}, s{S_K}", "B bytes")
        e(f"s_sub_u32 s{T}, s{S_M}, 1")
        e(f"s_mul_i32 s{T}, s{T}, s{S_LDC}")
        e(f"s_add_u32 s{T}, s{T}, s{S_N}")
        e(f"s_lshl_b32 s{T}, s{T}, 1", "C bytes")
        for d, p, nr in ((S_DA, S_A, f"s{S_NRA}"), (S_DB, S_B, f"s{S_NRB}"), (S_DC, S_C, f"s{T}")):
            e(f"s_mov_b32 s{d}, s{p}")
            e(f"s_and_b32 s{d + 1}, s{p + 1}, 0xffff")
            e(f"s_mov_b32 s{d + 2}, {nr}")
            e(f"s_mov_b32 s{d + 3}, 0x00020000")
        # ---- LDS-DMA destination bases; per-lane constants of the source offsets ----
        e(f"s_lshl_b32 s{S_M0A}, s{S_WAVE}, 10")
        e(f"s_mul_i32 s{T}, s{S_WR}, {B_WAVECOL}", "(wave >> 1): which wave column's B rows this wave stages")
        e(f"s_lshl_b32 s{T + 1}, s{S_WC}, 10", "(wave & 1): first or second 8 rows of the n tile")
        e(f"s_add_u32 s{S_M0B}, s{T}, s{T + 1}")
        e(f"s_add_u32 s{S_M0B}, s{S_M0B}, {A_BYTES}")
        vt = V_E
        e(f"v_lshrrev_b32 v{V_L3}, 3, v{vl}")
        e(f"v_and_b32 v{V_CH16}, 7, v{vl}")
        e(f"v_lshrrev_b32 v{vt}, 1, v{V_L3}")
        e(f"s_lshl_b32 s{T}, s{S_WC}, 2")
        e(f"v_add_u32 v{vt}, s{T}, v{vt}", "swizzle = 4 (wave & 1) + (lane >> 4)")
        e(f"v_xor_b32 v{V_CH16}, v{V_CH16}, v{vt}")
        e(f"v_lshlrev_b32 v{V_CH16}, 4, v{V_CH16}", "source chunk * 16")
        # ---- fragment read bases ----
        r, g, lo = V_E + 4, V_E + 5, V_E + 6
        e(f"v_and_b32 v{r}, 15, v{vl}")
        e(f"v_lshrrev_b32 v{g}, 4, v{vl}")
        e(f"v_lshrrev_b32 v{lo}, 3, v{r}")
        e(f"v_lshlrev_b32 v{lo}, 10, v{lo}", "(r >> 3) * 1024")
        e(f"v_and_b32 v{vt}, 7, v{r}")
        e(f"v_lshlrev_b32 v{vt}, 7, v{vt}")
        e(f"v_add_u32 v{lo}, v{lo}, v{vt}", "+ (r & 7) * 128")
        e(f"v_lshrrev_b32 v{vt}, 1, v{r}")
        e(f"v_and_b32 v{vt}, 7, v{vt}")
        e(f"v_xor_b32 v{vt}, v{vt}, v{g}")
        e(f"v_lshlrev_b32 v{vt}, 4, v{vt}")
        e(f"v_add_u32 v{lo}, v{lo}, v{vt}", "+ (g ^ ((r >> 1) & 7)) * 16")
        e(f"s_lshl_b32 s{T}, s{S_WR}, 14")
        e(f"v_add_u32 v{V_ALO[0]}, s{T}, v{lo}")
        e(f"v_xor_b32 v{V_AHI[0]}, 64, v{V_ALO[0]}")
        e(f"s_mul_i32 s{T}, s{S_WC}, {B_WAVECOL}")
        e(f"s_add_u32 s{T}, s{T}, {A_BYTES}")
        e(f"v_add_u32 v{V_BLO[0]}, s{T}, v{lo}")
        e(f"v_xor_b32 v{V_BHI[0]}, 64, v{V_BLO[0]}")
        e(f"s_mov_b32 s{T}, {STAGE}")
        for pair in (V_ALO, V_AHI, V_BLO, V_BHI):
            e(f"v_add_u32 v{pair[1]}, s{T}, v{pair[0]}")

    def tile_coords(self, tile: int, bm0: int, bn0: int, site: str) -> None:
        """First row / column of tile `tile` (SGPR numbers): XCD-contiguous runs of a GROUP_M = 8 raster over tile ids; the
        persistent walk visits ids wg, wg + grid, ... (grid a multiple of 8 or the whole problem: the XCD of an id stays)."""
        e = self.e
        T = S_T
        e(f"s_and_b32 s{T}, s{tile}, 7", "xcd")
        e(f"s_lshr_b32 s{T + 1}, s{tile}, 3", "index inside the XCD's run")
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
        e(f"s_cbranch_scc0 {self.L('partial_group_' + site)}")
        e(f"s_lshr_b32 s{T + 10}, s{T + 7}, 3", "tn")
        e(f"s_and_b32 s{T + 11}, s{T + 7}, 7", "tm in group")
        e(f"s_branch {self.L('tile_done_' + site)}")
        self.label(self.L("partial_group_" + site))
        e(f"s_mul_hi_u32 s{T + 10}, s{T + 7}, s{S_MAGIC_LAST}", "tn = in_group / height, height 2..7 (magic = ceil(2^32 / height))")
        e(f"s_cmp_eq_u32 s{T + 9}, 1")
        e(f"s_cselect_b32 s{T + 10}, s{T + 7}, s{T + 10}", "height 1: tn = in_group (2^32 / 1 does not fit the magic)")
        e(f"s_mul_i32 s{T + 11}, s{T + 10}, s{T + 9}")
        e(f"s_sub_u32 s{T + 11}, s{T + 7}, s{T + 11}")
        self.label(self.L("tile_done_" + site))
        e(f"s_add_u32 s{T + 11}, s{T + 11}, s{T + 8}", "tm")
        e(f"s_lshl_b32 s{bm0}, s{T + 11}, 8")
        e(f"s_mul_i32 s{bn0}, s{T + 10}, {TILE_N}")

    def tile_sources(self, bm0: int, bn0: int) -> None:
        """Per-lane source byte offsets of this wave's 19 LDS-DMA pieces for the tile at (s[bm0], s[bn0]) -> v1..v19, and the
        tile's scale loads -> v[V_NSC : V_NSC + 3] (parked in LDS later)."""
        e = self.e
        T = S_T
        vr, vt = V_T + 3, V_E
        e(f"s_lshl_b32 s{T}, s{S_WAVE}, 3")
        e(f"s_add_u32 s{T}, s{T}, s{bm0}")
        e(f"v_add_u32 v{vr}, s{T}, v{V_L3}", "A row of piece 0")
        e(f"s_sub_u32 s{T + 1}, s{S_M}, 1")
        for p in range(8):
            e(f"v_add_u32 v{vt}, {32 * p}, v{vr}")
            e(f"v_min_u32 v{vt}, s{T + 1}, v{vt}")
            e(f"v_mul_lo_u32 v{vt}, v{vt}, s{S_LDA}")
            e(f"v_add_u32 v{V_DMA + p}, v{vt}, v{V_CH16}")
        e(f"s_mul_i32 s{T}, s{S_WR}, {16 * NT}")
        e(f"s_lshl_b32 s{T + 2}, s{S_WC}, 3")
        e(f"s_add_u32 s{T}, s{T}, s{T + 2}")
        e(f"s_add_u32 s{T}, s{T}, s{bn0}")
        e(f"v_add_u32 v{vr}, s{T}, v{V_L3}", "B^T row of n tile 0's piece")
        e(f"s_sub_u32 s{T + 1}, s{S_N}, 1")
        for j in range(NT):
            e(f"v_add_u32 v{vt}, {16 * j}, v{vr}")
            e(f"v_min_u32 v{vt}, s{T + 1}, v{vt}")
            e(f"v_mul_lo_u32 v{vt}, v{vt}, s{S_LDB}")
            e(f"v_add_u32 v{V_DMA + 8 + j}, v{vt}, v{V_CH16}")
        # scales
        e(f"v_add_u32 v{vt}, s{bm0}, v{V_TID}")
        e(f"s_sub_u32 s{T}, s{S_M}, 1")
        e(f"v_min_u32 v{vt}, s{T}, v{vt}")
        e(f"s_cmp_lg_u32 s{S_SAVEC}, 0")
        e(f"s_cselect_b32 s{T + 2}, -1, 0")
        e(f"v_and_b32 v{vt}, s{T + 2}, v{vt}", "scalar scale: element 0")
        e(f"v_lshlrev_b32 v{vt}, 2, v{vt}")
        e(f"global_load_dword v{V_NSC}, v{vt}, s[{S_SA}:{S_SA + 1}]")
        e(f"s_sub_u32 s{T}, s{S_N}, 1")
        e(f"s_cmp_lg_u32 s{S_SBVEC}, 0")
        e(f"s_cselect_b32 s{T + 2}, -1, 0")
        for k, dst in ((0, V_NSC + 1), (256, V_NSC + 2)):
            e(f"v_add_u32 v{vt}, s{bn0}, v{V_TID}")
            if k:
                e(f"v_add_u32 v{vt}, {k}, v{vt}")
            e(f"v_min_u32 v{vt}, s{T}, v{vt}")
            e(f"v_and_b32 v{vt}, s{T + 2}, v{vt}")
            e(f"v_lshlrev_b32 v{vt}, 2, v{vt}")
            e(f"global_load_dword v{dst}, v{vt}, s[{S_SB}:{S_SB + 1}]")
        if self.bias:
            for k, dst in ((0, V_NBIAS), (256, V_NBIAS + 1)):
                e(f"v_add_u32 v{vt}, s{bn0}, v{V_TID}")
                if k:
                    e(f"v_add_u32 v{vt}, {k}, v{vt}")
                e(f"v_min_u32 v{vt}, s{T}, v{vt}", "columns past N: element N - 1 (never stored)")
                e(f"v_lshlrev_b32 v{vt}, 1, v{vt}")
                e(f"global_load_ushort v{dst}, v{vt}, s[{S_BIAS}:{S_BIAS + 1}]")

    def park_scales(self) -> None:
        """v[V_NSC..] (landed) -> the scale buffer at s[S_SCOFF]."""
        e = self.e
        vt = V_E
        e(f"v_mul_f32 v{V_NSC}, s{S_ACCSCALE}, v{V_NSC}", "acc_scale: 1, or the exact 1/4 of e4m3fnuz bytes run as OCP e4m3")
        e(f"v_lshlrev_b32 v{vt}, 2, v{V_TID}")
        e(f"v_add_u32 v{vt}, {LDS_SA}, v{vt}", "(DS offsets are 16 bits: the scale area's base goes into the address)")
        e(f"v_add_u32 v{vt}, s{S_SCOFF}, v{vt}")
        e(f"ds_write_b32 v{vt}, v{V_NSC}")
        e(f"ds_write_b32 v{vt}, v{V_NSC + 1} offset:{LDS_SB - LDS_SA}")
        e(f"ds_write_b32 v{vt}, v{V_NSC + 2} offset:{LDS_SB - LDS_SA + 1024}")
        if self.bias:
            T = S_T
            e(f"s_lshr_b32 s{T}, s{S_SCOFF}, 11", "scale buffer 0 / 1 ...")
            e(f"s_lshl_b32 s{T}, s{T}, {BIAS_BUF.bit_length() - 1}", "... -> bias buffer 0 / 1")
            e(f"v_lshlrev_b32 v{vt}, 1, v{V_TID}")
            e(f"v_add_u32 v{vt}, {LDS_BIAS}, v{vt}")
            e(f"v_add_u32 v{vt}, s{T}, v{vt}")
            e(f"ds_write_b16 v{vt}, v{V_NBIAS}")
            e(f"ds_write_b16 v{vt}, v{V_NBIAS + 1} offset:512")

    def first_tile(self) -> None:
        e = self.e
        self.tile_coords(S_TILE, S_BM0, S_BN0, "first")
        self.tile_sources(S_BM0, S_BN0)
        self.issue_stage(0, 0)
        self.issue_stage(1, 128)
        e(f"s_waitcnt vmcnt({2 * (8 + NT)})", "the three scale loads are the oldest vector-memory operations")
        self.park_scales()
        e(f"s_waitcnt vmcnt({8 + NT})", "stage 0 of this wave has landed")
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")

    def k_loop(self, steady: list[str]) -> None:
        """First fragment reads, the peeled first pair of K steps (addend 0), the loop over the remaining pairs."""
        e = self.e
        for t in steady:  # exactly the reads a step leaves outstanding for its successor, in their order of issue
            _, what, half = t.split(".")
            kind, idx = what[0], int(what[1:])
            if kind == "A":
                self.ds_read128(V_FA + 8 * idx + (4 if half == "hi" else 0), (V_AHI if half == "hi" else V_ALO)[0], 2048 * idx, t)
            else:
                self.ds_read128(V_FB + 8 * fb_slot(idx) + (4 if half == "hi" else 0), (V_BHI if half == "hi" else V_BLO)[0], 2048 * idx, t)
        assert self.lgkm == steady
        if self.fused_tail:
            self.label(self.L("go2"))  # (a tile whose predecessor's last step already read these fragments enters here)
            self.stamp(1)
        e(f"s_lshr_b32 s{S_CNT}, s{S_K}, 8")
        e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1", "pairs of K steps behind the peeled first pair (K >= 512: at least one)")
        self.step(0, "t0", "t1", first=True)
        self.step(1, "t1", "t0")
        assert self.lgkm == steady
        if self.fused_tail:
            e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1", "(the last pair is peeled too)")
            e(f"s_cmp_eq_u32 s{S_CNT}, 0")
            e(f"s_cbranch_scc1 {self.L('lastpair')}")
        self.label(self.L("loop"))
        self.step(0, "t0", "t1")
        self.step(1, "t1", "t0")
        e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
        e(f"s_cmp_lg_u32 s{S_CNT}, 0")
        e(f"s_cbranch_scc1 {self.L('loop')}")
        assert self.lgkm == steady, "the loop body must leave the queue as it found it"
        if self.fused_tail:
            self.label(self.L("lastpair"))
            # the next tile of this workgroup: its LDS-DMA source offsets replace this tile's (which the remaining steps no longer
            # request), its scale loads go out
            e(f"s_add_u32 s{S_TNEXT}, s{S_TILE}, s{S_GRID}")
            e(f"s_cmp_lt_u32 s{S_TNEXT}, s{S_NWG}")
            e(f"s_cselect_b32 s{S_HASNEXT}, 1, 0")
            e(f"s_cbranch_scc0 {self.L('ft_no_next')}")
            self.tile_coords(S_TNEXT, S_BM0N, S_BN0N, "next")
            self.tile_sources(S_BM0N, S_BN0N)
            self.label(self.L("ft_no_next"))
            self.tail_prepare()
            self.step(0, "t0", "t1", tail=1)
            self.step(1, "t1", "t0", tail=2)
            for q in range(SLOTS, SLOTS + 4):  # the last tiles' conversions and stores
                self.tail_convert_rest(q)

    @staticmethod
    def bias_pair(j: int) -> int:
        """First of the two registers holding n tile j's four bias values of this lane, packed as stored (epilogue_rows: the
        free even-aligned pairs v112..119 and v136..151)."""
        return 112 + 2 * j if j < 4 else 136 + 2 * (j - 4)

    def bias_ops(self, x: int, b: int, mask: str) -> list[str]:
        """v[x + 4 : x + 5] (four packed 16-bit outputs) += v[b : b + 1] (four packed 16-bit bias values), in the output dtype:
        bf16 -- both sides widened to fp32 (a shift / a mask), v_pk_add_f32, RNE cast (what torch's bf16 add does); fp16 --
        v_pk_add_f16 (the exact sum rounded once; equal to the fp32 route: a sum of two fp16 values that fp32 cannot hold exactly
        is more than 2^13 ulps of the smaller operand away from any fp16 rounding boundary)."""
        if self.out_fp16:
            return [f"v_pk_add_f16 v{x + 4}, v{x + 4}, v{b}", f"v_pk_add_f16 v{x + 5}, v{x + 5}, v{b + 1}"]
        return [f"v_lshlrev_b32 v{x}, 16, v{x + 4}", f"v_and_b32 v{x + 1}, {mask}, v{x + 4}",
                f"v_lshlrev_b32 v{x + 2}, 16, v{x + 5}", f"v_and_b32 v{x + 3}, {mask}, v{x + 5}",
                f"v_lshlrev_b32 v{x + 4}, 16, v{b}", f"v_and_b32 v{x + 5}, {mask}, v{b}",
                f"v_pk_add_f32 v[{x}:{x + 1}], v[{x}:{x + 1}], v[{x + 4}:{x + 5}]",
                f"v_lshlrev_b32 v{x + 4}, 16, v{b + 1}", f"v_and_b32 v{x + 5}, {mask}, v{b + 1}",
                f"v_pk_add_f32 v[{x + 2}:{x + 3}], v[{x + 2}:{x + 3}], v[{x + 4}:{x + 5}]",
                f"v_cvt_pk_bf16_f32 v{x + 4}, v{x}, v{x + 1}", f"v_cvt_pk_bf16_f32 v{x + 5}, v{x + 2}, v{x + 3}"]

    def convert_ops(self, i: int, j: int, sa: int, sb: int, x: int) -> list[str]:
        """Instructions that turn accumulator tile (i, j) into two registers v[x + 4 : x + 5] of packed 16-bit outputs:
        cast(sb * (sa * acc)), both products rounded to fp32 (conch/reference/quantization/scaled_gemm.py:21-23).  `sa` = first of
        the 8 row-scale registers, `sb` = the n tile's 4 column scales.  Returned as a list so that two tiles' dependent chains
        can be interleaved.  self.pk: packed multiplies (v_pk_mul_f32, the same IEEE operation per element) or scalar ones."""
        ops = []
        t = 8 * j + i
        if t < 64:
            ops += [f"v_accvgpr_read_b32 v{x + k}, a{4 * t + k}" for k in range(4)]
            src = x
        else:
            src = V_ACC + 4 * (t - 64)
        if self.int8:  # (float)acc: exact below 2^24, RNE above -- what the reference's fp32 matmul of int8 values holds
            ops += [f"v_cvt_f32_i32 v{x + k}, v{src + k}" for k in range(4)]
            src = x
        if self.pk:
            # 64-bit VGPR operands must be even-aligned: the scale of an odd m tile is the HIGH half of the pair below it
            pair = sa + (i & ~1)
            sel = "op_sel_hi:[1,0]" if i % 2 == 0 else "op_sel:[0,1] op_sel_hi:[1,1]"
            ops.append(f"v_pk_mul_f32 v[{x}:{x + 1}], v[{src}:{src + 1}], v[{pair}:{pair + 1}] {sel}")
            ops.append(f"v_pk_mul_f32 v[{x + 2}:{x + 3}], v[{src + 2}:{src + 3}], v[{pair}:{pair + 1}] {sel}")
            ops.append(f"v_pk_mul_f32 v[{x}:{x + 1}], v[{x}:{x + 1}], v[{sb}:{sb + 1}]")
            ops.append(f"v_pk_mul_f32 v[{x + 2}:{x + 3}], v[{x + 2}:{x + 3}], v[{sb + 2}:{sb + 3}]")
        else:
            ops += [f"v_mul_f32 v{x + k}, v{sa + i}, v{src + k}" for k in range(4)]
            ops += [f"v_mul_f32 v{x + k}, v{sb + k}, v{x + k}" for k in range(4)]
        if self.out_fp16:
            ops += [f"v_cvt_f16_f32 v{x + k}, v{x + k}" for k in range(4)]
            ops.append(f"v_pack_b32_f16 v{x + 4}, v{x}, v{x + 1}")
            ops.append(f"v_pack_b32_f16 v{x + 5}, v{x + 2}, v{x + 3}")
        else:
            ops.append(f"v_cvt_pk_bf16_f32 v{x + 4}, v{x}, v{x + 1}")
            ops.append(f"v_cvt_pk_bf16_f32 v{x + 5}, v{x + 2}, v{x + 3}")
        if self.bias:
            ops += self.bias_ops(x, self.bias_pair(j), f"s{S_T + 7}")
        return ops

    @staticmethod
    def rows_stores() -> int:
        """buffer_store_dwordx4 per wave in epilogue_rows: 8 passes x 2 row groups x column groups of 8 chunks (48 at NT = 11)"""
        return MT * 2 * -(-2 * NT // 8)

    def epilogue_rows(self, vc: int, vg: int) -> None:
        """Whole wave sub-tiles (128 x 176 inside M x N): every wave parks its cast outputs in ITS OWN 19 KiB of the dead STAGE-1
        operand LDS (stage 0 may already be receiving the next tile) as a row-major image -- eight passes of one m tile (16 rows x
        352 bytes at a pitch of 368) through two alternating buffers -- reads the previous pass's image back 16 bytes per lane (a
        wave instruction = 8 rows x 128 contiguous bytes) and stores it with buffer_store_dwordx4 BETWEEN the conversions of the
        current pass: 8 line segments per store instruction instead of 64 (T21 of the programming guide;
        profiles/r03/store_rate.txt).  A wave's LDS operations execute in order, so no barrier separates its writes from its
        reads.  Four tiles' conversion chains are interleaved, the 44 column scales are read once, and every read-back / store
        address is one per-lane base + an immediate (+ a scalar row-group offset)."""
        e = self.e
        T = S_T
        PITCH = 32 * NT + 16         # 368: a wave's row of 176 outputs = 352 bytes, + 16 so that the rows of a tile spread over the banks
        NCG = -(-2 * NT // 8)        # 3: column groups of eight 16-byte chunks in a row
        LASTCG = 2 * NT - 8 * (NCG - 1)   # 6: chunks of the last group
        REGION = STAGE // 4          # 19456 per wave, inside stage 1
        BUF = 16 * PITCH             # 5888: one pass's image
        sa = V_FA                    # v32..39: 8 row scales (even-aligned)
        sb = V_FA + 8                # v40..83: 4 column scales per n tile
        waddr = V_FA + 52            # v84: LDS write address of this lane (row c of the pass's m tile, n tile 0, buffer 0)
        rbase = V_FA + 53            # v85: LDS read-back address of this lane (row group 0, column group 0, buffer 0)
        gbase = V_FA + 54            # v86: C byte offset of this lane's 16 bytes (row group 0, column group 0)
        gmask = V_FA + 55            # v87: the same, or out of range for the lanes past the row's 22nd chunk (column group 2)
        xa = V_FA + 56               # v88..: 6 temporaries per interleaved conversion chain (even-aligned)
        xb = xa + 6
        buf = V_E                    # v120..135: 4 x 4 data registers of the read-back
        assert buf + 16 <= V_NSC and 2 * BUF <= REGION
        # row scales
        e(f"s_lshl_b32 s{T}, s{S_WR}, 7")
        e(f"v_add_u32 v{xa}, s{T}, v{vc}")
        e(f"v_lshlrev_b32 v{xa}, 2, v{xa}")
        e(f"v_add_u32 v{xa}, {LDS_SA}, v{xa}")
        e(f"v_add_u32 v{xa}, s{S_SCOFF}, v{xa}", "this tile's scale buffer")
        for i in range(MT):
            e(f"ds_read_b32 v{sa + i}, v{xa} offset:{64 * i}")
        # column scales of all 11 n tiles
        e(f"s_mul_i32 s{T}, s{S_WC}, {16 * NT}")
        e(f"v_lshlrev_b32 v{xa + 1}, 2, v{vg}")
        e(f"v_add_u32 v{xa + 1}, s{T}, v{xa + 1}")
        e(f"v_lshlrev_b32 v{xa + 1}, 2, v{xa + 1}")
        e(f"v_add_u32 v{xa + 1}, {LDS_SB}, v{xa + 1}")
        e(f"v_add_u32 v{xa + 1}, s{S_SCOFF}, v{xa + 1}")
        for j in range(NT):
            e(f"ds_read_b128 v[{sb + 4 * j}:{sb + 4 * j + 3}], v{xa + 1} offset:{64 * j}")
        if self.bias:
            # the 4 bias values of this lane's columns in every n tile, packed as stored: 8 bytes at 2 (16 NT wc + 16 j + 4 g)
            e(f"s_mul_i32 s{T}, s{S_WC}, {32 * NT}")
            e(f"s_lshr_b32 s{T + 1}, s{S_SCOFF}, 11")
            e(f"s_lshl_b32 s{T + 1}, s{T + 1}, {BIAS_BUF.bit_length() - 1}", "this tile's bias buffer")
            e(f"s_add_u32 s{T}, s{T}, s{T + 1}")
            e(f"v_lshlrev_b32 v{xa + 2}, 3, v{vg}")
            e(f"v_add_u32 v{xa + 2}, s{T}, v{xa + 2}")
            e(f"v_add_u32 v{xa + 2}, {LDS_BIAS}, v{xa + 2}")
            for j in range(NT):
                b = self.bias_pair(j)
                e(f"ds_read_b64 v[{b}:{b + 1}], v{xa + 2} offset:{32 * j}")
            e(f"s_mov_b32 s{T + 7}, 0xffff0000", "high half of a packed pair of bf16")
        # this lane's write address inside the wave's region: row c, byte 8 g
        e(f"s_mul_i32 s{T + 1}, s{S_WAVE}, {REGION}")
        e(f"s_add_u32 s{T + 1}, s{T + 1}, {STAGE}", "the wave's staging region (stage 1)")
        e(f"v_mul_u32_u24 v{waddr}, {PITCH}, v{vc}")
        e(f"v_lshl_add_u32 v{waddr}, v{vg}, 3, v{waddr}")
        e(f"v_add_u32 v{waddr}, s{T + 1}, v{waddr}")
        # read-back: lane L -> row L / 8 of a group of 8 rows, 16-byte chunk L % 8 of a group of 8 chunks (3 groups: 8 + 8 + 6)
        e(f"s_lshl_b32 s{T + 2}, s{S_LDC}, 1", "C row pitch in bytes")
        e(f"s_lshl_b32 s{T + 3}, s{S_WR}, 7")
        e(f"s_add_u32 s{T + 3}, s{T + 3}, s{S_BM0}")
        e(f"s_mul_i32 s{T + 3}, s{T + 3}, s{T + 2}", "first row of the wave, bytes")
        e(f"s_mul_i32 s{T + 4}, s{S_WC}, {16 * NT}")
        e(f"s_add_u32 s{T + 4}, s{T + 4}, s{S_BN0}")
        e(f"s_lshl_b32 s{T + 4}, s{T + 4}, 1")
        e(f"s_add_u32 s{T + 3}, s{T + 3}, s{T + 4}", "+ first column: byte offset of the wave's corner in C")
        e(f"v_and_b32 v{xb}, 63, v{V_TID}", "lane")
        e(f"v_lshrrev_b32 v{xb + 1}, 3, v{xb}", "row in the group")
        e(f"v_and_b32 v{xb + 2}, 7, v{xb}", "chunk in the group")
        e(f"v_mul_u32_u24 v{rbase}, {PITCH}, v{xb + 1}")
        e(f"v_lshl_add_u32 v{rbase}, v{xb + 2}, 4, v{rbase}")
        e(f"v_add_u32 v{rbase}, s{T + 1}, v{rbase}")
        e(f"v_mul_lo_u32 v{gbase}, v{xb + 1}, s{T + 2}")
        e(f"v_lshl_add_u32 v{gbase}, v{xb + 2}, 4, v{gbase}")
        e(f"v_add_u32 v{gbase}, s{T + 3}, v{gbase}")
        e(f"v_cmp_gt_u32 vcc, {LASTCG}, v{xb + 2}", "the last column group of a row: its first LASTCG lanes of eight")
        e(f"v_mov_b32 v{xb + 3}, 0x80000000")
        e(f"v_cndmask_b32 v{gmask}, v{xb + 3}, v{gbase}, vcc")
        e(f"s_lshl_b32 s{T + 5}, s{T + 2}, 3", "8 rows of C in bytes: one row group")
        e(f"s_mov_b32 s{T + 6}, 0", "scalar offset of the row group being stored")
        e("s_waitcnt lgkmcnt(0)")
        units = [(rg, cg) for rg in range(2) for cg in range(NCG)]   # of one pass's image: 2 row groups x NCG column groups
        stores = 0

        def read_unit(p: int, u: int) -> None:
            rg, cg = units[u]
            b = buf + 4 * (u % 4)
            e(f"ds_read_b128 v[{b}:{b + 3}], v{rbase} offset:{(p % 2) * BUF + rg * 8 * PITCH + cg * 128}")

        def store_unit(u: int) -> None:
            nonlocal stores
            rg, cg = units[u]
            b = buf + 4 * (u % 4)
            e(f"buffer_store_dwordx4 v[{b}:{b + 3}], v{gmask if cg == NCG - 1 else gbase}, s[{S_DC}:{S_DC + 3}], s{T + 6} offen offset:{cg * 128}" + self.store_policy)
            stores += 1
            if cg == NCG - 1:
                e(f"s_add_u32 s{T + 6}, s{T + 6}, s{T + 5}", "next row group")

        G = self.chains             # conversion chains interleaved (a lone wave pays the VALU latency of every dependent pair)
        xs = [xa + 6 * c for c in range(G)]
        assert xs[-1] + 6 <= V_E
        for p in range(MT):
            self.c(f"---- pass {p}: m tile {p} -> buffer {p % 2}" + (f"; image of pass {p - 1} read back and stored in between" if p else "") + " ----")
            tiles = [(p, j) for j in range(NT)]
            groups = [tiles[k:k + G] for k in range(0, len(tiles), G)]
            per = -(-len(units) // len(groups))          # read-back units handled per group boundary
            nxt = 0                                      # next unit of pass p - 1 to read
            pending: list[int] = []
            if p:
                for _ in range(per):
                    if nxt < len(units):
                        read_unit(p - 1, nxt)
                        pending.append(nxt)
                        nxt += 1
            for grp in groups:
                ops = [self.convert_ops(i, j, sa, sb + 4 * j, xs[c]) for c, (i, j) in enumerate(grp)]
                for k in range(max(len(o) for o in ops)):
                    for o in ops:
                        if k < len(o):
                            e(o[k])
                if p:
                    e("s_waitcnt lgkmcnt(0)", "the units read one group ago (and this wave's earlier image writes)")
                    for u in pending:
                        store_unit(u)
                    pending = []
                for c, (i, j) in enumerate(grp):
                    e(f"ds_write_b64 v{waddr}, v[{xs[c] + 4}:{xs[c] + 5}] offset:{(p % 2) * BUF + 32 * j}")
                if p:
                    for _ in range(per):
                        if nxt < len(units):
                            read_unit(p - 1, nxt)
                            pending.append(nxt)
                            nxt += 1
            if p:
                e("s_waitcnt lgkmcnt(0)")
                for u in pending:
                    store_unit(u)
                assert nxt == len(units)
        self.c(f"---- image of pass {MT - 1} ----")
        e("s_waitcnt lgkmcnt(0)")
        read_unit(MT - 1, 0)
        for u in range(len(units)):
            if u + 1 < len(units):
                read_unit(MT - 1, u + 1)
                e("s_waitcnt lgkmcnt(1)")
            else:
                e("s_waitcnt lgkmcnt(0)")
            store_unit(u)
        assert stores == self.rows_stores(), stores
        # the next tile's scale loads and stage-0 pieces are OLDER than these stores: all but the stores have landed
        e(f"s_cmp_lg_u32 s{S_HASNEXT}, 0")
        e(f"s_cbranch_scc0 {self.L('tail')}")
        e(f"s_waitcnt vmcnt({self.rows_stores()})")
        e(f"s_branch {self.L('tail')}")

    def epilogue(self) -> None:
        e = self.e
        T = S_T
        e("s_waitcnt vmcnt(0)")
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier", "every wave is past its last fragment read and every LDS-DMA has landed: the operand LDS is dead")
        # ---- the next tile of this workgroup: its scale loads and its whole first stage go out BEFORE this tile's epilogue ----
        e(f"s_add_u32 s{S_TNEXT}, s{S_TILE}, s{S_GRID}")
        e(f"s_cmp_lt_u32 s{S_TNEXT}, s{S_NWG}")
        e(f"s_cselect_b32 s{S_HASNEXT}, 1, 0")
        e(f"s_cbranch_scc0 {self.L('no_next')}")
        self.tile_coords(S_TNEXT, S_BM0N, S_BN0N, "next")
        self.tile_sources(S_BM0N, S_BN0N)
        self.issue_stage(0, 0)
        self.label(self.L("no_next"))
        e("s_nop 15", "last MFMA's result -> VALU")
        e("s_nop 7")
        vl, vc, vg, vt = V_T, V_T + 1, V_T + 2, V_T + 3
        e(f"v_and_b32 v{vl}, 63, v{V_TID}")
        e(f"v_and_b32 v{vc}, 15, v{vl}", "c: m inside a tile")
        e(f"v_lshrrev_b32 v{vg}, 4, v{vl}", "g: n quad inside a tile")
        if self.rows:
            # whole sub-tile of this wave inside M x N?  (wave-uniform; the waves of a workgroup may take different paths)
            e(f"s_lshl_b32 s{T}, s{S_WR}, 7")
            e(f"s_add_u32 s{T}, s{T}, s{S_BM0}")
            e(f"s_add_u32 s{T}, s{T}, 128")
            e(f"s_cmp_le_u32 s{T}, s{S_M}")
            e(f"s_cbranch_scc0 {self.L('direct')}")
            e(f"s_mul_i32 s{T}, s{S_WC}, {16 * NT}")
            e(f"s_add_u32 s{T}, s{T}, s{S_BN0}")
            e(f"s_add_u32 s{T}, s{T}, {16 * NT}")
            e(f"s_cmp_le_u32 s{T}, s{S_N}")
            e(f"s_cbranch_scc0 {self.L('direct')}")
            e(f"s_and_b32 s{T}, s{S_LDC}, 7", "16-byte stores: C rows 16-byte aligned (ldc % 8 == 0; the base is checked by the launcher)")
            e(f"s_cmp_eq_u32 s{T}, 0")
            e(f"s_cbranch_scc0 {self.L('direct')}")
            self.epilogue_rows(vc, vg)
            self.label(self.L("direct"))
        e(f"v_and_b32 v{vl}, 63, v{V_TID}")
        e(f"v_and_b32 v{vc}, 15, v{vl}", "c: m inside a tile")
        e(f"v_lshrrev_b32 v{vg}, 4, v{vl}", "g: n quad inside a tile")
        # row offsets into C (bytes), rows past M -> out of range of the descriptor
        rowoff = V_FA            # 8 registers (the fragments are dead)
        sa = V_FA + 8            # 8 registers
        e(f"s_lshl_b32 s{T}, s{S_WR}, 7")
        e(f"s_add_u32 s{T}, s{T}, s{S_BM0}")
        e(f"v_add_u32 v{vt}, s{T}, v{vc}", "row of m tile 0")
        e(f"s_mul_i32 s{T + 1}, s{S_WC}, {16 * NT}")
        e(f"s_add_u32 s{T + 1}, s{T + 1}, s{S_BN0}", "first column of this wave")
        e(f"v_lshlrev_b32 v{V_E}, 2, v{vg}")
        e(f"v_add_u32 v{V_E}, s{T + 1}, v{V_E}", "column of n tile 0")
        e(f"v_lshlrev_b32 v{V_E}, 1, v{V_E}", "bytes")
        e(f"s_lshl_b32 s{T + 2}, s{S_LDC}, 1", "row pitch in bytes")
        for i in range(MT):
            e(f"v_add_u32 v{V_E + 1}, {16 * i}, v{vt}")
            e(f"v_mul_lo_u32 v{V_E + 2}, v{V_E + 1}, s{T + 2}")
            e(f"v_add_u32 v{V_E + 2}, v{V_E + 2}, v{V_E}")
            e(f"v_cmp_gt_u32 vcc, s{S_M}, v{V_E + 1}")
            e(f"v_mov_b32 v{V_E + 3}, 0x80000000")
            e(f"v_cndmask_b32 v{rowoff + i}, v{V_E + 3}, v{V_E + 2}, vcc")
        # scales from LDS
        e(f"s_lshl_b32 s{T}, s{S_WR}, 7")
        e(f"v_add_u32 v{V_E + 1}, s{T}, v{vc}")
        e(f"v_lshlrev_b32 v{V_E + 1}, 2, v{V_E + 1}")
        e(f"v_add_u32 v{V_E + 1}, {LDS_SA}, v{V_E + 1}", "sa address")
        e(f"v_add_u32 v{V_E + 1}, s{S_SCOFF}, v{V_E + 1}")
        for i in range(MT):
            e(f"ds_read_b32 v{sa + i}, v{V_E + 1} offset:{64 * i}")
        e(f"s_mul_i32 s{T}, s{S_WC}, {16 * NT}")
        e(f"v_lshlrev_b32 v{V_E + 2}, 2, v{vg}")
        e(f"v_add_u32 v{V_E + 2}, s{T}, v{V_E + 2}")
        e(f"v_lshlrev_b32 v{V_E + 2}, 2, v{V_E + 2}")
        e(f"v_add_u32 v{V_E + 2}, {LDS_SB}, v{V_E + 2}", "sb address")
        e(f"v_add_u32 v{V_E + 2}, s{S_SCOFF}, v{V_E + 2}")
        sbreg = V_FA + 16        # 4 registers per n tile, double buffered
        bsreg = V_FA + 20        # (bias variants) the n tile's four packed bias values: 2 registers
        if self.bias:
            e(f"s_mul_i32 s{T + 3}, s{S_WC}, {32 * NT}")
            e(f"s_lshr_b32 s{T + 4}, s{S_SCOFF}, 11")
            e(f"s_lshl_b32 s{T + 4}, s{T + 4}, {BIAS_BUF.bit_length() - 1}")
            e(f"s_add_u32 s{T + 3}, s{T + 3}, s{T + 4}")
            e(f"v_lshlrev_b32 v{V_E + 3}, 3, v{vg}")
            e(f"v_add_u32 v{V_E + 3}, s{T + 3}, v{V_E + 3}")
            e(f"v_add_u32 v{V_E + 3}, {LDS_BIAS}, v{V_E + 3}", "bias address")
            e(f"s_mov_b32 s{T + 7}, 0xffff0000")
        # n tiles of this wave that start below N (N is a multiple of 16: whole tiles)
        e(f"s_sub_u32 s{S_NVALID}, s{S_N}, s{T + 1}", "columns left of N from this wave's first one (may be <= 0)")
        e(f"s_cmp_gt_i32 s{S_NVALID}, 0")
        e(f"s_cbranch_scc0 {self.L('direct_done')}")
        e(f"s_lshr_b32 s{S_NVALID}, s{S_NVALID}, 4")
        for j in range(NT):
            e(f"s_cmp_gt_u32 s{S_NVALID}, {j}")
            e(f"s_cbranch_scc0 {self.L('direct_done')}")
            e(f"ds_read_b128 v[{sbreg}:{sbreg + 3}], v{V_E + 2} offset:{64 * j}")
            if self.bias:
                e(f"ds_read_b64 v[{bsreg}:{bsreg + 1}], v{V_E + 3} offset:{32 * j}")
            e("s_waitcnt lgkmcnt(0)")
            for i in range(MT):
                t = 8 * j + i
                x = V_E + 4      # 4 values
                if t < 64:
                    for k in range(4):
                        e(f"v_accvgpr_read_b32 v{x + k}, a{4 * t + k}")
                    src = [f"v{x + k}" for k in range(4)]
                else:
                    src = [f"v{V_ACC + 4 * (t - 64) + k}" for k in range(4)]
                if self.int8:
                    for k in range(4):
                        e(f"v_cvt_f32_i32 v{x + k}, {src[k]}")
                    src = [f"v{x + k}" for k in range(4)]
                for k in range(4):
                    e(f"v_mul_f32 v{x + k}, v{sa + i}, {src[k]}", "sa * acc" if k == 0 else "")
                for k in range(4):
                    e(f"v_mul_f32 v{x + k}, v{sbreg + k}, v{x + k}", "sb * (sa * acc)" if k == 0 else "")
                if self.out_fp16:
                    for k in range(4):
                        e(f"v_cvt_f16_f32 v{x + k}, v{x + k}")
                    e(f"v_pack_b32_f16 v{x + 4}, v{x}, v{x + 1}")
                    e(f"v_pack_b32_f16 v{x + 5}, v{x + 2}, v{x + 3}")
                else:
                    e(f"v_cvt_pk_bf16_f32 v{x + 4}, v{x}, v{x + 1}")
                    e(f"v_cvt_pk_bf16_f32 v{x + 5}, v{x + 2}, v{x + 3}")
                if self.bias:
                    for op in self.bias_ops(x, bsreg, f"s{T + 7}"):
                        e(op)
                e(f"buffer_store_dwordx2 v[{x + 4}:{x + 5}], v{rowoff + i}, s[{S_DC}:{S_DC + 3}], 0 offen offset:{32 * j}")
                e("s_nop 1")
        self.label(self.L("direct_done"))
        e("s_waitcnt vmcnt(0)", "(an edge wave's store count varies: drain; the next tile's loads are older than the stores)")
        # ---- tail: hand over to the next tile, or end ----
        self.label(self.L("tail"))
        self.stamp(3)
        if self.probe:
            self.stamp(4)
            self.write_stamps()
        e(f"s_cmp_lg_u32 s{S_HASNEXT}, 0")
        e(f"s_cbranch_scc0 {self.L('exit')}")
        if self.probe:
            self.stamp(0)
        e(f"s_xor_b32 s{S_SCOFF}, s{S_SCOFF}, {SCALE_BUF}")
        self.park_scales()
        e(f"s_mov_b32 s{S_TILE}, s{S_TNEXT}")
        e(f"s_mov_b32 s{S_BM0}, s{S_BM0N}")
        e(f"s_mov_b32 s{S_BN0}, s{S_BN0N}")
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier", "every wave is done with its staging region (stage 1) and its stage-0 pieces of the next tile have landed")
        self.issue_stage(1, 128)
        e(f"s_branch {self.L('go')}")
        self.label(self.L("exit"))
        e("s_endpgm")

    def fused_exit(self) -> None:
        """Timing experiment: behind a tile whose stores rode in its last K step.  The next tile's two stages and its scale loads
        are in flight (or nothing is): drain, park the scales, swap, meet, enter the next tile's loop behind its fragment reads."""
        e = self.e
        self.stamp(3)
        if self.probe:
            self.stamp(4)
            self.write_stamps()
        e(f"s_cmp_lg_u32 s{S_HASNEXT}, 0")
        e(f"s_cbranch_scc0 {self.L('exit')}")
        if self.probe:
            self.stamp(0)
        e("s_waitcnt vmcnt(0)", "(the experiment drains here; the real thing would wait for the scale loads and the two stages only)")
        e(f"s_xor_b32 s{S_SCOFF}, s{S_SCOFF}, {SCALE_BUF}")
        self.park_scales()
        e(f"s_mov_b32 s{S_TILE}, s{S_TNEXT}")
        e(f"s_mov_b32 s{S_BM0}, s{S_BM0N}")
        e(f"s_mov_b32 s{S_BN0}, s{S_BN0N}")
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        e(f"s_branch {self.L('go2')}")
        self.label(self.L("exit"))
        e("s_waitcnt vmcnt(0)")
        e("s_endpgm")

    def build(self) -> str:
        self.setup_once()
        self.first_tile()
        # steady-state queue of outstanding fragment reads: generate a pair once to learn what it leaves behind
        scratch = Gen(self.name, self.out_fp16, int8=self.int8)  # (only its final queue state is used: no bias, no probe)
        scratch.step(0, "t0", "t1")
        scratch.step(1, "t1", "t0")
        steady = list(scratch.lgkm)
        self.label(self.L("go"))
        if not self.fused_tail:
            self.stamp(1)
        self.k_loop(steady)
        self.stamp(2)
        if self.fused_tail:
            self.fused_exit()
        else:
            self.epilogue()
        return self.render()

    def render(self) -> str:
        name = self.name
        head = f"""\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
\t.amdhsa_code_object_version 6
\t.text
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
\t\t.amdhsa_group_segment_fixed_size {self.lds_total}
\t\t.amdhsa_private_segment_fixed_size 0
\t\t.amdhsa_kernarg_size 128
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

    @staticmethod
    def metadata(names: list) -> str:
        """names: (kernel name, LDS bytes) pairs"""
        kernels = ""
        for name, lds_total in names:
            kernels += f"""  - .agpr_count:     256
    .args:
      - .offset:         0
        .size:           128
        .value_kind:     by_value
    .group_segment_fixed_size: {lds_total}
    .kernarg_segment_align: 8
    .kernarg_segment_size: 128
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
    text = ""
    names = []
    variants = []
    for nt in (11, 9, 7):
        sfx = "" if nt == 11 else f"_n{nt}"
        variants += [(f"conch_gemm1w_fp8_bf16{sfx}", nt, False, False, True, False), (f"conch_gemm1w_fp8_f16{sfx}", nt, True, False, True, False),
                     (f"conch_gemm1w_i8_bf16{sfx}", nt, False, False, True, True), (f"conch_gemm1w_i8_f16{sfx}", nt, True, False, True, True)]
        # ... and each with the bias add in its epilogue (name + "_bias"; the kernarg slot of the diagnostic buffer holds the vector)
        variants += [(name + "_bias", nt_, f16_, probe_, pk_, i8_) for name, nt_, f16_, probe_, pk_, i8_ in variants[-4:]]
    variants += [("conch_gemm1w_fp8_bf16_probe", 11, False, True, True, False), ("conch_gemm1w_fp8_bf16_alt", 11, False, False, False, False),
                 ("conch_gemm1w_fp8_bf16_alt_probe", 11, False, True, False, False), ("conch_gemm1w_i8_bf16_probe", 11, False, True, True, True)]
    for name, nt, fp16, probe, pk, int8 in variants:
        configure(nt)
        # the A/B twin (pk False): non-temporal C stores (correct results; profiles/r05/asm1w_nt_store_ab.txt; sc1: asm1w_sc1_store_ab.txt).
        # CONCH_GEN_EXPERIMENT=fused_tail at build time makes it the fused-tail TIMING experiment instead (WRONG results: never
        # in a library that ships; profiles/r05/asm1w_fused_tail_timing.txt)
        fused = (not pk) and os.environ.get("CONCH_GEN_EXPERIMENT") == "fused_tail"
        g = Gen(name, out_fp16=fp16, probe=probe, pk=True, int8=int8, store_policy="" if pk or fused else " nt", fused_tail=fused,
                bias=name.endswith("_bias"))
        body = g.build()
        if text:
            body = body.split("\n", 2)[2]  # one target / code-object-version header per file
        text += body
        names.append((name, g.lds_total))
    text += Gen.metadata(names)
    with open(out, "w") as f:
        f.write(text)


if __name__ == "__main__":
    main()
