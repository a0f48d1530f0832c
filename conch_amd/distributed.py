"""N-sharded scaled GEMM across the GPUs of one node (BASELINE config C5).

The reference has no distributed code at all; this is the multi-GPU form of the same hot path.
Output column j depends only on B[:, j], scale_b[j] and bias[j], so N is partitioned into
`world_size` contiguous column blocks: every rank holds all of A / scale_a and its block of
B / scale_b / bias, computes its [M, N/G] block of C with the single-GPU kernel (no reduction, so
every element is bit-identical to the single-GPU result), and ONE kind of collective -- an all-gather of the
blocks over RCCL/xGMI -- assembles C.  One process per GPU; backend "nccl" is RCCL on ROCm, "gloo" is used by
the CPU tests.

Data movement, designed for xGMI (point-to-point links: the gather, not the GEMM, is the long pole at C5 --
58.7 MB per rank against a 0.15 ms GEMM):

* M is cut into `panels` row panels.  The GEMM of panel p writes its [h, N/G] block STRAIGHT into this rank's slot
  of the panel's gather buffer (the kernels take any output row stride / base), so the all-gather is in place and
  nothing is copied before it.
* The all-gather of panel p runs on a side stream while the GEMM of panel p+1 runs on the compute stream.
* An all-gather concatenates along dim 0, so a gathered panel is [G, h, N/G] ("column-block major").
  `gathered_panels()` hands that layout out as is ([panels, G, h, N/G]), for consumers that are themselves
  column-parallel; `gathered_blocks()` is the un-panelised [G, M, N/G] form (one GEMM, one gather).  `__call__`
  returns the row-major [M, N] tensor: each gathered panel is unpacked into it ([G, h, N/G] -> rows) on the side
  stream, behind its own gather and under the next panel's gather -- the one whole-matrix permuting copy of round 1
  is gone.  With world_size == 1 the GEMM writes the row-major result directly (column offset + row stride N) and
  there is no gather and no copy.
* DIRECT mode (round 4; `direct=True`, SURVEY.md H7 / 8(e): "the epilogue writes to peers"): every rank's row-major [M, N]
  result lives in memory every other rank has mapped (torch symmetric memory = HIP IPC handles exchanged once, at
  construction), and the tile kernel stores each of its tiles into ALL G results (`conch_scaled_gemm_multi`: one local
  destination and G - 1 peer-mapped ones) -- the all-gather is the GEMM's epilogue, its stores leave over all xGMI links at
  once, there is no staging buffer, no collective on the data path and no unpack pass (`__call__` issues no `copy_`).  What is
  left of the exchange is completion: a barrier of the group behind the last panel, so that a rank reads its result only after
  every peer's stores have landed (and one ahead of the first panel of the next call, so that nobody overwrites a result a peer
  is still reading).  UNMEASURED ON HARDWARE: no box with two GPUs was available to any round; the layout logic is covered by
  a world-size-2 gloo test with injected shared-memory "peers", the kernel by a one-GPU test with three local destinations.
  The RCCL all-gather path stays the default.
"""

from __future__ import annotations

from collections.abc import Callable

import torch
import torch.distributed as dist


def shard_bounds(n: int, world_size: int, rank: int) -> tuple[int, int]:
    """Column range [lo, hi) of `rank`; N must divide evenly (the all-gather needs equal blocks)."""
    if n % world_size:
        raise ValueError(f"N={n} is not divisible by world_size={world_size}")
    per = n // world_size
    return rank * per, (rank + 1) * per


def _default_gemm_into(out, a, b, scale_a, scale_b, bias):
    """The product path: the public launcher writing into a caller-provided (possibly strided) output view."""
    from conch_amd.kernels.quantization.gemm import scaled_gemm_launcher
    from conch_amd.ops.quantization._metadata import create_scaled_metadata

    meta = create_scaled_metadata(a, b, scale_a, scale_b, out.dtype)
    scaled_gemm_launcher(out, a, b, scale_a, scale_b, meta, bias=bias)


def _default_gemm_multi(outs, a, b, scale_a, scale_b, bias):
    """The product path of direct mode: ONE launch whose epilogue stores the block into every tensor of `outs` (views of the
    G results with one shape and one row stride: outs[0] local, the others peer-mapped)."""
    from conch_amd.kernels.quantization.gemm import scaled_gemm_multi_launcher
    from conch_amd.ops.quantization._metadata import create_scaled_metadata

    meta = create_scaled_metadata(a, b, scale_a, scale_b, outs[0].dtype)
    scaled_gemm_multi_launcher(outs, a, b, scale_a, scale_b, meta, bias=bias)


def symmetric_results(m: int, n: int, dtype: torch.dtype, device: torch.device, group=None):
    """([M, N] result of every rank as seen from THIS rank, barrier callable): torch symmetric memory -- every rank allocates
    its result from the symmetric-memory allocator and the rendezvous exchanges the IPC handles, so `results[r]` is a tensor on
    this device's address space that aliases rank r's buffer (stores to it travel over xGMI)."""
    import torch.distributed._symmetric_memory as symm

    grp = group if group is not None else dist.group.WORLD
    buf = symm.empty((m, n), dtype=dtype, device=device)
    hdl = symm.rendezvous(buf, grp)
    rank, world = dist.get_rank(grp), dist.get_world_size(grp)
    results = [buf if r == rank else hdl.get_buffer(r, (m, n), dtype) for r in range(world)]
    return results, (lambda: hdl.barrier(channel=0))


def ipc_results(m: int, n: int, dtype: torch.dtype, device: torch.device, group=None):
    """The same contract as symmetric_results() on plain HIP IPC handles (torch's tensor-sharing reductions: what
    torch.multiprocessing sends a CUDA tensor with), exchanged through the group's object collectives: the fallback where the
    symmetric-memory allocator is not available -- and the only form two ranks on ONE device (a dry run) can use, since the
    allocator refuses ranks that share a device.  The barrier drains this rank's stream, then meets on the host."""
    from torch.multiprocessing.reductions import reduce_tensor

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    buf = torch.empty((m, n), dtype=dtype, device=device)
    rebuild, args = reduce_tensor(buf)
    shared = [None] * world
    dist.all_gather_object(shared, (rebuild, args), group=group)
    results = [buf if r == rank else shared[r][0](*shared[r][1]) for r in range(world)]

    def barrier() -> None:
        torch.cuda.current_stream(device).synchronize()
        dist.barrier(group)

    return results, barrier


def device_cus(device=None) -> int:
    """Compute units of `device` (256 on MI355X); 256 where there is no GPU to ask (CPU dry runs of the layout logic)."""
    try:
        if torch.cuda.is_available():
            return int(torch.cuda.get_device_properties(device).multi_processor_count)
    except Exception:  # noqa: BLE001, S110 -- a heuristic's input, never a reason to fail
        pass
    return 256


def direct_results(m: int, n: int, dtype: torch.dtype, device: torch.device, group=None, scheme: str | None = None):
    """(results, barrier, scheme name) of direct mode.  EXPERIMENTAL: has only met two ranks on one device, never xGMI.

    `scheme` (or $CONCH_AMD_DIRECT_SCHEME): "symmetric" = torch symmetric memory (device-side barrier), "ipc" = plain HIP IPC
    handles with a host barrier, None / "auto" = symmetric memory where EVERY rank can have it, else IPC.  Both constructors are
    collective, so the choice is made collectively: each rank reports whether symmetric memory is possible for it (the module is
    there and no two ranks share a device -- the allocator refuses that, and a refusal INSIDE its rendezvous would leave the ranks
    on different collective sequences), the flags are reduced with MIN, and all ranks take the same branch.  A failure inside the
    chosen constructor is raised, never papered over; the scheme in use is logged once per process."""
    import logging
    import os

    scheme = (scheme or os.environ.get("CONCH_AMD_DIRECT_SCHEME") or "auto").lower()
    if scheme not in ("auto", "symmetric", "ipc"):
        raise ValueError(f"direct mode: unknown scheme {scheme!r} (want auto, symmetric or ipc)")
    if scheme == "auto":
        world = dist.get_world_size(group)
        # can THIS rank have symmetric memory?  The module must import AND its allocator must hand out a buffer here (a local
        # allocation, no rendezvous: an allocator this ROCm build lacks shows up now, on every rank alike, not inside the
        # collective constructor)
        try:
            import torch.distributed._symmetric_memory as symm

            symm.empty(16, dtype=torch.uint8, device=device)
            can = 1
        except Exception:  # noqa: BLE001 -- any refusal selects the IPC scheme on all ranks
            can = 0
        # which PHYSICAL device: per-rank HIP_VISIBLE_DEVICES makes every rank "cuda:0", so the index says nothing
        mine = -1
        if device.type == "cuda":
            idx = device.index if device.index is not None else torch.cuda.current_device()
            props = torch.cuda.get_device_properties(idx)
            mine = str(getattr(props, "uuid", "")) or f"{getattr(props, 'pci_domain_id', 0)}:{getattr(props, 'pci_bus_id', idx)}:{getattr(props, 'pci_device_id', 0)}"
        seen = [None] * world
        dist.all_gather_object(seen, (can, os.uname().nodename, mine), group=group)
        shared_device = len({(host, idx) for _, host, idx in seen}) < world
        scheme = "symmetric" if all(c for c, _, _ in seen) and not shared_device and device.type == "cuda" else "ipc"
    logging.getLogger("conch_amd.distributed").warning(
        "direct mode (experimental, unmeasured on xGMI): peer results through %s",
        "torch symmetric memory (device-side barrier)" if scheme == "symmetric" else "HIP IPC handles + host barrier (stream.synchronize + dist.barrier per call)")
    if scheme == "symmetric":
        results, barrier = symmetric_results(m, n, dtype, device, group)
    else:
        results, barrier = ipc_results(m, n, dtype, device, group)
    return results, barrier, scheme


def planned_tile_columns(m: int, n_local: int, k: int, in_dtype: torch.dtype) -> int:
    """Tile width (columns) of the kernel the library's dispatcher launches for an [m, k] x [k, n_local] scaled GEMM of
    `in_dtype`: 352 / 288 / 224 on the one-wave-per-SIMD assembly kernel, 128 on the few-tile kernel, else 256 (the plan is a pure
    function of the problem -- conch_debug_scaled_plan, no GPU needed; shapes outside the MFMA kernels' contract: 256)."""
    import ctypes

    from conch_amd import _C

    code = {torch.float8_e4m3fn: 3, torch.int8: 4}.get(in_dtype)
    if code is None or min(m, n_local, k) <= 0:
        return 256
    fn = _C.load().conch_debug_scaled_plan
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_int64] * 3 + [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    out = (ctypes.c_int * 2)()
    if fn(m, n_local, k, code, out) != 0:
        return 256
    return {3: int(out[1]) or 352, 1: 128, 2: 0}.get(out[0], 256)  # (0: the split-K decode kernel has no output tiles to count)


def default_panels(m: int, n_local: int, tile: int = 256, cus: int | None = None, device=None, k: int | None = None,
                   in_dtype: torch.dtype | None = None) -> int:
    """Row panels per call: as many as keep every panel's GEMM at >= ~0.85 of one full round of tiles (a panel below one round
    idles CUs).  Tiles are `tile` rows high; their WIDTH is the one the dispatcher's kernel uses for a panel of that height when
    `k` and `in_dtype` are given (C5 on 8 GPUs: the 3584-column shard runs 256 x 224 assembly tiles, 32 x 16 of them -> 2 panels
    of one full round each), else 256 columns (the HIP tile kernel; round 5's rule).  `cus` defaults to the device's count."""
    if cus is None:
        cus = device_cus(device)
    tiles_m = -(-m // tile)
    best = 1
    for p in range(2, 9):
        # a panel is a whole number of rows AND of tile rows: M = 1001 or 515 rows stay one panel
        if m % p or (m // p) % tile or tiles_m % p:
            continue
        width = planned_tile_columns(m // p, n_local, k, in_dtype) if k is not None and in_dtype is not None else 256
        if width < 224:
            continue  # a panel this small leaves the 256-row tile kernels (128 x 128 tiles, split K): not a full round of anything
        if (tiles_m // p) * -(-n_local // width) >= 0.85 * cus:
            best = p
    return best


class NShardedScaledGemm:
    """scaled_gemm with B, scale_b and bias sharded on N and C all-gathered.

    Buffers are allocated once (288 GB of HBM per GPU: the gathered C of config C5 is 470 MB, its staging another
    470 MB) and reused by every call.  `gemm_fn(a, b, scale_a, scale_b, output_dtype, bias) -> Tensor` may be injected
    (the CPU tests pass the oracle); the default writes through the public launcher into the gather buffer.
    """

    def __init__(
        self,
        m: int,
        n: int,
        output_dtype: torch.dtype,
        device: torch.device,
        group: dist.ProcessGroup | None = None,
        gemm_fn: Callable | None = None,
        panels: int | None = None,
        force_collective: bool = False,
        direct: bool = False,
        peer_results=None,
        gemm_multi_fn: Callable | None = None,
        scheme: str | None = None,
        k: int | None = None,
        in_dtype: torch.dtype | None = None,
    ) -> None:
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.m, self.n = m, n
        self.lo, self.hi = shard_bounds(n, self.world_size, self.rank)
        self.n_local = self.hi - self.lo
        self.output_dtype = output_dtype
        self.device = torch.device(device)
        self.gemm_fn = gemm_fn
        # one rank normally writes the row-major result directly; `force_collective` sends it through the staging buffers, the
        # side stream and the all-gather anyway (a one-rank RCCL group: the GPU test of the exchange path on a one-GPU box)
        self._exchange = self.world_size > 1 or force_collective
        # row panels exist to overlap the gather of one panel with the GEMM of the next: with one rank there is no gather,
        # the heuristic is not consulted and an explicit `panels` only shapes gathered_panels()' view
        if panels is None:
            # (`k`, `in_dtype`: optional hints -- with them the panel rule counts the tiles of the kernel that will run)
            panels = default_panels(m, self.n_local, device=self.device if self.device.type == "cuda" else None, k=k, in_dtype=in_dtype) if self.world_size > 1 else 1
        self.panels = panels
        if self.panels < 1 or m % self.panels:
            raise ValueError(f"M={m} is not divisible into {self.panels} panels")
        self.h = m // self.panels
        g = self.world_size
        self._blocks = None  # [G, M, N/G] staging of gathered_blocks(), allocated on first use
        self._c = None if (direct and self.world_size > 1) else torch.empty((m, n), dtype=output_dtype, device=self.device)  # row-major result
        # gather staging, one [G, h, N/G] buffer per panel (unused when world_size == 1)
        # direct mode: the G row-major results as seen from this rank ([rank] = the local one) + a completion barrier.
        # `peer_results` = (results, barrier) injects them (tests; another IPC scheme); default: torch symmetric memory.
        # `gemm_multi_fn(outs, a, b, scale_a, scale_b, bias)` injects the multi-destination product (the CPU tests' oracle).
        self.direct = bool(direct) and self.world_size > 1
        self.gemm_multi_fn = gemm_multi_fn
        self._results, self._barrier = None, None
        self.direct_scheme = "injected" if peer_results is not None else None
        if self.direct:
            if peer_results is not None:
                self._results, self._barrier = peer_results
            else:
                self._results, self._barrier, self.direct_scheme = direct_results(m, n, output_dtype, self.device, group, scheme)
            if len(self._results) != g or any(tuple(t.shape) != (m, n) or t.dtype != output_dtype for t in self._results):
                raise ValueError("direct mode: one [M, N] result of the output dtype per rank")
            self._c = self._results[self.rank]
        staged = self._exchange and not self.direct
        self._stage = torch.empty((self.panels, g, self.h, self.n_local), dtype=output_dtype, device=self.device) if staged else None
        self._side = torch.cuda.Stream(self.device) if self.device.type == "cuda" and staged else None

    # -- the local product -------------------------------------------------------------------------------------------
    def _gemm_into(self, out: torch.Tensor, a, b_shard, scale_a, scale_b_shard, bias_shard) -> None:
        if self.gemm_fn is None:
            _default_gemm_into(out, a, b_shard, scale_a, scale_b_shard, bias_shard)
        else:
            out.copy_(self.gemm_fn(a, b_shard, scale_a, scale_b_shard, self.output_dtype, bias_shard))

    def _panel_inputs(self, p: int, a, scale_a):
        rows = slice(p * self.h, (p + 1) * self.h)
        return a[rows], (scale_a[rows] if scale_a.numel() > 1 else scale_a)

    def local_gemm(self, a, b_shard, scale_a, scale_b_shard, bias_shard=None) -> torch.Tensor:
        """This rank's [M, N/G] block of C (b_shard: [K, N/G], any strides), as a view of the row-major result buffer."""
        out = self._c[:, self.lo : self.hi]
        self._gemm_into(out, a, b_shard, scale_a, scale_b_shard, bias_shard)
        return out

    # -- the exchange ------------------------------------------------------------------------------------------------
    def _run_direct(self, a, b_shard, scale_a, scale_b_shard, bias_shard) -> None:
        """Direct mode: per row panel ONE launch that stores this rank's [h, N/G] block into all G results; no collective on the
        data path, no staging, no unpack."""
        self._barrier()  # nobody still reads the results of the previous call
        order = [self.rank] + [r for r in range(self.world_size) if r != self.rank]  # the local destination first
        for p in range(self.panels):
            a_p, sa_p = self._panel_inputs(p, a, scale_a)
            rows = slice(p * self.h, (p + 1) * self.h)
            outs = [self._results[r][rows, self.lo : self.hi] for r in order]
            if self.gemm_multi_fn is not None:
                self.gemm_multi_fn(outs, a_p, b_shard, sa_p, scale_b_shard, bias_shard)
            else:
                _default_gemm_multi(outs, a_p, b_shard, sa_p, scale_b_shard, bias_shard)
        self._barrier()  # every peer's stores into this rank's result have landed

    def _run(self, a, b_shard, scale_a, scale_b_shard, bias_shard, unpack: bool) -> None:
        if not self._exchange:
            self.local_gemm(a, b_shard, scale_a, scale_b_shard, bias_shard)
            return
        if self.direct:
            if not unpack:
                raise RuntimeError("direct mode produces the row-major result only (gathered_panels / gathered_blocks are layouts of the all-gather path)")
            self._run_direct(a, b_shard, scale_a, scale_b_shard, bias_shard)
            return
        cuda = self._side is not None
        main = torch.cuda.current_stream(self.device) if cuda else None
        for p in range(self.panels):
            a_p, sa_p = self._panel_inputs(p, a, scale_a)
            slot = self._stage[p, self.rank]  # [h, N/G], contiguous: the in-place input of the panel's all-gather
            self._gemm_into(slot, a_p, b_shard, sa_p, scale_b_shard, bias_shard)
            if cuda:
                ready = torch.cuda.Event()
                ready.record(main)
                self._side.wait_event(ready)
                with torch.cuda.stream(self._side):
                    self._gather_panel(p, slot, unpack)
            else:
                self._gather_panel(p, slot, unpack)
        if cuda:
            done = torch.cuda.Event()
            done.record(self._side)
            main.wait_event(done)

    def _gather_panel(self, p: int, slot: torch.Tensor, unpack: bool) -> None:
        g = self.world_size
        dist.all_gather_into_tensor(self._stage[p].view(g * self.h, self.n_local), slot, group=self.group)
        if unpack:  # [G, h, N/G] -> rows [h, G * N/G] of the row-major result
            rows = self._c[p * self.h : (p + 1) * self.h]
            rows.view(self.h, g, self.n_local).copy_(self._stage[p].permute(1, 0, 2))

    def gathered_panels(self, a, b_shard, scale_a, scale_b_shard, bias_shard=None) -> torch.Tensor:
        """[panels, G, M/panels, N/G]: [p, g] holds rows of panel p, columns [g*N/G, (g+1)*N/G) of C -- the layout the
        panelised all-gathers produce, with no copy at all and every gather but the last under the next panel's GEMM
        (for column-parallel consumers that can walk row panels)."""
        if not self._exchange:
            c = self.local_gemm(a, b_shard, scale_a, scale_b_shard, bias_shard)
            return c.view(self.panels, self.h, 1, self.n_local).permute(0, 2, 1, 3)
        self._run(a, b_shard, scale_a, scale_b_shard, bias_shard, unpack=False)
        return self._stage

    def gathered_blocks(self, a, b_shard, scale_a, scale_b_shard, bias_shard=None) -> torch.Tensor:
        """[G, M, N/G], whatever `panels` is: blocks[g] is rank g's [M, N/G] column block of C (columns
        [g*N/G, (g+1)*N/G)) -- ONE GEMM into this rank's slot and ONE in-place all-gather, no copy.  The panelised,
        overlapped form of the same data is gathered_panels()."""
        if not self._exchange:
            return self.local_gemm(a, b_shard, scale_a, scale_b_shard, bias_shard).view(1, self.m, self.n_local)
        if self.direct:
            raise RuntimeError("direct mode produces the row-major result only")
        if self._blocks is None:
            self._blocks = torch.empty((self.world_size, self.m, self.n_local), dtype=self.output_dtype, device=self.device)
        slot = self._blocks[self.rank]
        self._gemm_into(slot, a, b_shard, scale_a, scale_b_shard, bias_shard)
        dist.all_gather_into_tensor(self._blocks.view(self.world_size * self.m, self.n_local), slot, group=self.group)
        return self._blocks

    def __call__(self, a, b_shard, scale_a, scale_b_shard, bias_shard=None) -> torch.Tensor:
        """Row-major [M, N] C, identical on every rank."""
        self._run(a, b_shard, scale_a, scale_b_shard, bias_shard, unpack=True)
        return self._c
