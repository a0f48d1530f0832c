"""bitsandbytes quantize_4bit microbenchmark (CLI of the reference's benchmarks/bnb_quantize_blockwise_benchmark.py).

Baseline: absmax per block + nearest-code search written with plain torch ops on the GPU (timing comparison only)."""

import click
import torch

from _common import report_match, run_pair
from conch_amd.ops.quantization.bitsandbytes.functional import dequantize_4bit, quantize_4bit
from conch_amd.third_party.vllm.utils import seed_everything

_DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}


@click.command()
@click.option("--blocksize", default=64, type=int)
@click.option("--size-multiplier", default=458752, type=int)
@click.option("--quant-type", default="nf4", type=click.Choice(["nf4", "fp4"]))
@click.option("--input-dtype", default="bf16", type=click.Choice(sorted(_DT)))
@click.option("--compress-statistics", is_flag=True)
@click.option("--iteration-time-ms", default=2000, type=int)
@click.option("--warmup-time-ms", default=500, type=int)
@click.option("--verbose", is_flag=True)
@click.option("--gpu", default="cuda:0")
@click.option("--csv", is_flag=True)
def main(blocksize, size_multiplier, quant_type, input_dtype, compress_statistics, iteration_time_ms, warmup_time_ms, verbose, gpu, csv):
    seed_everything(0)
    device = torch.device(gpu)
    dtype = _DT[input_dtype]
    n = blocksize * size_multiplier
    x = torch.randn((n,), dtype=dtype, device=device)

    def ours():
        return quantize_4bit(x, blocksize=blocksize, compress_statistics=compress_statistics, quant_type=quant_type)

    def baseline():  # the absmax reduction and the scaling: the part plain torch can express in a few kernels
        blocks = x.view(-1, blocksize).float()
        am = blocks.abs().amax(dim=1, keepdim=True)
        return blocks * am.reciprocal()

    q, state = ours()
    back = dequantize_4bit(q, state, blocksize=blocksize, quant_type=quant_type)
    err = (back.float() - x.float()).abs().mean().item() / x.float().abs().mean().item()
    report_match(err < 0.2, f"(relative round-trip error {err:.3f})")
    if verbose:
        print(q)
    params = {"blocksize": blocksize, "size": n, "quant_type": quant_type, "input_dtype": input_dtype,
              "compress_statistics": compress_statistics}
    run_pair("bnb quantize_4bit", ours, baseline, params, iteration_time_ms, warmup_time_ms, csv, nbytes=float(n * (0.5 + x.element_size())))


if __name__ == "__main__":
    main()
