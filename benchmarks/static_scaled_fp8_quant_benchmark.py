"""static_scaled_fp8_quant microbenchmark (CLI of the reference's benchmarks/static_scaled_fp8_quant_benchmark.py:22-123)."""

import click
import torch

from _common import DTYPES, report_match, run_pair, torch_fp8_quant
from conch_amd.ops.quantization.fp8 import scaled_fp8_quant
from conch_amd.platforms import current_platform
from conch_amd.third_party.vllm.utils import seed_everything


@click.command()
@click.option("--hidden-size", default=4068, type=int)
@click.option("--num-tokens", default=4096, type=int)
@click.option("--scale", default=2.1, type=float)
@click.option("--dynamic", is_flag=True, help="per-token dynamic quantisation (scale=None) instead of the static per-tensor scale")
@click.option("--dtype", "dtype_name", default="float16", type=click.Choice(sorted(DTYPES)))
@click.option("--fnuz", is_flag=True, help="emit the MI300-era float8_e4m3fnuz instead of OCP float8_e4m3fn")
@click.option("--iteration-time-ms", default=2000, type=int)
@click.option("--warmup-time-ms", default=500, type=int)
@click.option("--verbose", is_flag=True)
@click.option("--gpu", default="cuda:0")
@click.option("--csv", is_flag=True)
@click.option("--compile-ref", is_flag=True, help="torch.compile() the PyTorch baseline (the reference's flag; needs a working Inductor backend)")
@click.option("--compile-conch", is_flag=True, help="torch.compile() the conch_amd op: captured as ONE opaque custom op (ops/quantization/_compile.py)")
def main(hidden_size, num_tokens, scale, dynamic, dtype_name, fnuz, iteration_time_ms, warmup_time_ms, verbose, gpu, csv, compile_ref, compile_conch):
    seed_everything(0)
    device = torch.device(gpu)
    dtype = DTYPES[dtype_name]
    fp8 = torch.float8_e4m3fnuz if fnuz else current_platform.fp8_dtype()
    x = torch.rand(num_tokens, hidden_size, dtype=dtype, device=device)
    s = torch.tensor([scale], dtype=torch.float32, device=device)
    if dynamic:  # per-token: scale[t] = absmax(x[t]) / finfo.max, then the static arithmetic row by row
        lim = torch.finfo(fp8).max
        amax = x.float().abs().amax(dim=-1, keepdim=True)
        srow = torch.where(amax > 0, amax / lim, torch.ones_like(amax))
        out, sout = scaled_fp8_quant(x, None, output_dtype=fp8)
        ref = (x.float() * srow.reciprocal()).clamp(-lim, lim).to(fp8)
        report_match(torch.equal(out.view(torch.uint8), ref.view(torch.uint8)) and torch.equal(sout, srow))
        s = None
    else:
        out, _ = scaled_fp8_quant(x, s, output_dtype=fp8)
        ref = torch_fp8_quant(x, s, fp8)
        report_match(torch.equal(out.view(torch.uint8), ref.view(torch.uint8)))
    if verbose:
        print(out)
    params = {"hidden_size": hidden_size, "num_tokens": num_tokens, "scale": scale, "dtype": dtype_name,
              "fp8": str(fp8)}
    params["dynamic"] = dynamic

    def baseline():
        if not dynamic:
            return torch_fp8_quant(x, s, fp8)
        lim = torch.finfo(fp8).max
        amax = x.float().abs().amax(dim=-1, keepdim=True)
        return (x.float() * (amax / lim).reciprocal()).clamp(-lim, lim).to(fp8)

    ours = lambda: scaled_fp8_quant(x, s, output_dtype=fp8)  # noqa: E731
    if compile_conch and not dynamic:  # (the dynamic op has no custom-op form: it is this build's extension)
        compiled = torch.compile(lambda t, sc: scaled_fp8_quant(t, sc, output_dtype=fp8), fullgraph=True)
        ours = lambda: compiled(x, s)  # noqa: E731
    if compile_ref:
        baseline = torch.compile(baseline)
    run_pair("dynamic_scaled_fp8_quant" if dynamic else "static_scaled_fp8_quant", ours,
             baseline, params, iteration_time_ms, warmup_time_ms, csv,
             nbytes=float(x.numel() * (x.element_size() + 1)))


if __name__ == "__main__":
    main()
