"""mixed_precision_gemm microbenchmark (CLI of the reference's benchmarks/mixed_precision_gemm_benchmark.py:75-187)."""

import math

import click
import torch

from _common import DTYPES, report_match, run_pair
from conch_amd.ops.quantization.gemm import mixed_precision_gemm
from conch_amd.third_party.vllm.quant_utils import pack_rows, quantize_weights
from conch_amd.third_party.vllm.scalar_type import scalar_types
from conch_amd.third_party.vllm.utils import seed_everything

WEIGHT_TYPES = {"uint4b8": scalar_types.uint4b8, "uint8b128": scalar_types.uint8b128, "uint4": scalar_types.uint4,
                "uint8": scalar_types.uint8}


@click.command()
@click.option("--m-dim", default=4096, type=int)
@click.option("--k-dim", default=8192, type=int)
@click.option("--n-dim", default=4096, type=int)
@click.option("--input-dtype", default="float16", type=click.Choice(["float16", "bfloat16"]))
@click.option("--weight-dtype", default="uint4b8", type=click.Choice(sorted(WEIGHT_TYPES)))
@click.option("--zero-points", is_flag=True)
@click.option("--group-size", default=128, type=int)
@click.option("--prepack", is_flag=True, help="pre-pack the weights for the tile kernel first (one-off, not timed), as the "
              "reference does for its comparator (benchmarks/mixed_precision_gemm_benchmark.py:59-75)")
@click.option("--iteration-time-ms", default=2000, type=int)
@click.option("--warmup-time-ms", default=500, type=int)
@click.option("--verbose", is_flag=True)
@click.option("--gpu", default="cuda:0")
@click.option("--csv", is_flag=True)
def main(m_dim, k_dim, n_dim, input_dtype, weight_dtype, zero_points, group_size, prepack, iteration_time_ms, warmup_time_ms,
         verbose, gpu, csv):
    seed_everything(0)
    device = torch.device(gpu)
    dtype = DTYPES[input_dtype]
    wt = WEIGHT_TYPES[weight_dtype]
    a = (10 * (torch.rand((m_dim, k_dim), dtype=torch.float32) - 0.3)).to(dtype)
    b = (10 * (torch.rand((k_dim, n_dim), dtype=torch.float32) - 0.3)).to(dtype)
    w_ref, w_q, w_s, w_zp = quantize_weights(b, wt, group_size, zero_points=zero_points)  # host side, one-off
    packed = pack_rows(w_q, wt.size_bits, *w_q.shape)
    a, w_ref, packed, w_s = a.to(device), w_ref.to(device), packed.to(device), w_s.to(device)
    w_zp = None if w_zp is None else w_zp.to(device)

    pre = None
    if prepack:
        from conch_amd.ops.quantization.prepack import mixed_precision_gemm_prepacked, prepack_mixed_weights

        pre = prepack_mixed_weights(packed, wt.size_bits, m_hint=m_dim, per_group_zero_points=w_zp is not None and w_zp.numel() > 1)

    def ours():
        if pre is not None:
            return mixed_precision_gemm_prepacked(a, pre, w_s, w_zp, wt.bias, group_size)
        return mixed_precision_gemm(a, packed, w_s, w_zp, wt.size_bits, wt.bias, group_size)

    ref = torch.matmul(a, w_ref)
    out = ours()
    ok = torch.allclose(ref.float(), out.float(), rtol=1e-1, atol=min(5e-2 * math.sqrt(k_dim), 1))
    report_match(ok, f"(max |diff| {(ref.float() - out.float()).abs().max().item():.4g})")
    if verbose:
        print(out)
    params = {"m_dim": m_dim, "k_dim": k_dim, "n_dim": n_dim, "input_dtype": input_dtype, "weight_dtype": weight_dtype,
              "zero_points": zero_points, "group_size": group_size, "prepack": prepack}
    run_pair("mixed_precision_gemm", ours, lambda: torch.matmul(a, w_ref), params, iteration_time_ms, warmup_time_ms,
             csv, flops=2.0 * m_dim * n_dim * k_dim)


if __name__ == "__main__":
    main()
