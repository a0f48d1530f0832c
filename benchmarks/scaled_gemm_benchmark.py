"""scaled_gemm microbenchmark (CLI of the reference's benchmarks/scaled_gemm_benchmark.py:39-151)."""

import click
import torch

from _common import DTYPES, report_match, run_pair, torch_scaled_gemm
from conch_amd.ops.quantization.gemm import scaled_gemm
from conch_amd.platforms import current_platform
from conch_amd.third_party.vllm.utils import seed_everything


@click.command()
@click.option("--m-dim", default=4096, type=int)
@click.option("--k-dim", default=8192, type=int)
@click.option("--n-dim", default=4096, type=int)
@click.option("--input-dtype", default="int8", type=click.Choice(["int8", "fp8"]))
@click.option("--output-dtype", default="bfloat16", type=click.Choice(["float16", "bfloat16"]))
@click.option("--use-scalar-scale-a", is_flag=True)
@click.option("--use-scalar-scale-b", is_flag=True)
@click.option("--use-bias", is_flag=True)
@click.option("--iteration-time-ms", default=2000, type=int)
@click.option("--warmup-time-ms", default=500, type=int)
@click.option("--verbose", is_flag=True)
@click.option("--gpu", default="cuda:0")
@click.option("--csv", is_flag=True)
def main(m_dim, k_dim, n_dim, input_dtype, output_dtype, use_scalar_scale_a, use_scalar_scale_b, use_bias,
         iteration_time_ms, warmup_time_ms, verbose, gpu, csv):
    seed_everything(0)
    device = torch.device(gpu)
    torch.set_default_device(device)
    out_dtype = DTYPES[output_dtype]
    if input_dtype == "fp8":
        fp8 = current_platform.fp8_dtype()
        a = (0.25 * torch.rand((m_dim, k_dim), dtype=torch.float32)).to(fp8)
        b = (0.25 * torch.rand((n_dim, k_dim), dtype=torch.float32)).to(fp8).T
    else:
        a = torch.randint(-32, 32, (m_dim, k_dim), dtype=torch.int8)
        b = torch.randint(-32, 32, (n_dim, k_dim), dtype=torch.int8).T
    scale_a = torch.rand((1, 1)) if use_scalar_scale_a else 0.25 * torch.rand((m_dim, 1))
    scale_b = torch.rand((1, 1)) if use_scalar_scale_b else 0.25 * torch.rand((n_dim, 1))
    bias = torch.rand((n_dim,), dtype=out_dtype) if use_bias else None

    ref = torch_scaled_gemm(a, b, scale_a, scale_b, out_dtype, bias)
    out = scaled_gemm(a, b, scale_a, scale_b, out_dtype, bias)
    ok = torch.allclose(ref.float(), out.float(), rtol=1e-1, atol=1e-1)
    report_match(ok, f"(max |diff| {(ref.float() - out.float()).abs().max().item():.4g})")
    if verbose:
        print(out)
    params = {"m_dim": m_dim, "k_dim": k_dim, "n_dim": n_dim, "input_dtype": input_dtype, "output_dtype": output_dtype,
              "scalar_scale_a": use_scalar_scale_a, "scalar_scale_b": use_scalar_scale_b, "bias": use_bias}
    extra = None
    if input_dtype == "fp8" and not use_scalar_scale_a and not use_scalar_scale_b:
        # the vendor library's row-wise fp8 GEMM (hipBLASLt) -- this platform's counterpart of the reference's vLLM CUTLASS leg
        sb_row = scale_b.T.contiguous()
        extra = {"hipBLASLt (torch._scaled_mm)": lambda: torch._scaled_mm(a, b, scale_a=scale_a, scale_b=sb_row, bias=bias, out_dtype=out_dtype)}
    run_pair("scaled_gemm", lambda: scaled_gemm(a, b, scale_a, scale_b, out_dtype, bias),
             lambda: torch_scaled_gemm(a, b, scale_a, scale_b, out_dtype, bias), params, iteration_time_ms,
             warmup_time_ms, csv, flops=2.0 * m_dim * n_dim * k_dim, extra=extra)


if __name__ == "__main__":
    main()
