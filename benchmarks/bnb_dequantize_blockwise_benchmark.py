"""bitsandbytes dequantize_4bit microbenchmark (CLI of the reference's benchmarks/bnb_dequantize_blockwise_benchmark.py:35-113).

Baseline: the same table lookup + scale written with plain torch ops on the GPU."""

import click
import torch

from _common import report_match, run_pair
from conch_amd.ops.quantization.bitsandbytes.functional import dequantize_4bit, quantize_4bit
from conch_amd.third_party.vllm.utils import seed_everything

_DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}
_NF4 = [-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635, -0.18477343022823334,
        -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224,
        0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0]
_FP4 = [0.0, 0.0052083333, 0.6666666, 1.0, 0.333333, 0.5, 0.166666, 0.25, -0.0, -0.0052083333, -0.666666, -1.0, -0.333333, -0.5,
        -0.166666, -0.25]


@click.command()
@click.option("--blocksize", default=64, type=int)
@click.option("--size-multiplier", default=458752, type=int, help="tensor size = blocksize x this (default: 29.4 M elements)")
@click.option("--quant-type", default="nf4", type=click.Choice(["nf4", "fp4"]))
@click.option("--dequant-dtype", default="bf16", type=click.Choice(sorted(_DT)))
@click.option("--compress-statistics", is_flag=True)
@click.option("--iteration-time-ms", default=2000, type=int)
@click.option("--warmup-time-ms", default=500, type=int)
@click.option("--verbose", is_flag=True)
@click.option("--gpu", default="cuda:0")
@click.option("--csv", is_flag=True)
def main(blocksize, size_multiplier, quant_type, dequant_dtype, compress_statistics, iteration_time_ms, warmup_time_ms, verbose, gpu, csv):
    seed_everything(0)
    device = torch.device(gpu)
    dtype = _DT[dequant_dtype]
    n = blocksize * size_multiplier
    x = torch.randn((n,), dtype=dtype, device=device)
    q, state = quantize_4bit(x, blocksize=blocksize, compress_statistics=compress_statistics, quant_type=quant_type)
    table = torch.tensor(_NF4 if quant_type == "nf4" else _FP4, dtype=torch.float32, device=device)

    def ours():
        return dequantize_4bit(q, state, blocksize=blocksize, quant_type=quant_type)

    absmax = state.absmax if not state.nested else None

    def baseline():
        qq = q.view(-1)
        codes = torch.stack([qq >> 4, qq & 0xF], dim=1).view(-1).long()
        am = absmax if absmax is not None else torch.ones(n // blocksize, device=device)
        return (table[codes] * am.repeat_interleave(blocksize)).to(dtype)

    out = ours()
    if not state.nested:
        report_match(torch.equal(out.view(-1), baseline()))
    if verbose:
        print(out)
    params = {"blocksize": blocksize, "size": n, "quant_type": quant_type, "dequant_dtype": dequant_dtype,
              "compress_statistics": compress_statistics}
    run_pair("bnb dequantize_4bit", ours, baseline, params, iteration_time_ms, warmup_time_ms, csv, nbytes=float(n * (0.5 + x.element_size())))


if __name__ == "__main__":
    main()
