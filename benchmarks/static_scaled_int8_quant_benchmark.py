"""static_scaled_int8_quant microbenchmark (CLI of the reference's benchmarks/static_scaled_int8_quant_benchmark.py:17-114)."""

import click
import torch

from _common import DTYPES, report_match, run_pair, torch_int8_quant
from conch_amd.ops.quantization.int8 import scaled_int8_quant
from conch_amd.third_party.vllm.utils import seed_everything


@click.command()
@click.option("--hidden-size", default=4608, type=int)
@click.option("--num-tokens", default=4096, type=int)
@click.option("--scale", default=2.1, type=float)
@click.option("--dynamic", is_flag=True, help="per-token dynamic quantisation (scale=None) instead of the static per-tensor scale")
@click.option("--dtype", "dtype_name", default="float16", type=click.Choice(sorted(DTYPES)))
@click.option("--iteration-time-ms", default=2000, type=int)
@click.option("--warmup-time-ms", default=500, type=int)
@click.option("--verbose", is_flag=True)
@click.option("--gpu", default="cuda:0")
@click.option("--csv", is_flag=True)
@click.option("--compile-ref", is_flag=True, help="torch.compile() the PyTorch baseline (the reference's flag; needs a working Inductor backend)")
@click.option("--compile-conch", is_flag=True, help="torch.compile() the conch_amd op: captured as ONE opaque custom op (ops/quantization/_compile.py)")
def main(hidden_size, num_tokens, scale, dynamic, dtype_name, iteration_time_ms, warmup_time_ms, verbose, gpu, csv, compile_ref, compile_conch):
    seed_everything(0)
    device = torch.device(gpu)
    dtype = DTYPES[dtype_name]
    x = torch.rand(num_tokens, hidden_size, dtype=dtype, device=device) * 1000
    s = torch.tensor([scale], dtype=torch.float32, device=device)
    if dynamic:  # per-token: scale[t] = absmax(x[t]) / 127, then the static arithmetic row by row
        amax = x.float().abs().amax(dim=-1, keepdim=True)
        srow = torch.where(amax > 0, amax / 127.0, torch.ones_like(amax))
        out, sout = scaled_int8_quant(x, None)
        ref = (x * srow.reciprocal()).clamp(-128, 127).to(torch.int8)
        report_match(torch.equal(out, ref) and torch.equal(sout, srow))
        s = None
    else:
        out, _ = scaled_int8_quant(x, s)
        ref = torch_int8_quant(x, s)
        report_match(bool((out.int() - ref.int()).abs().max().item() <= 1))
    if verbose:
        print(out)
    params = {"hidden_size": hidden_size, "num_tokens": num_tokens, "scale": scale, "dtype": dtype_name}
    params["dynamic"] = dynamic

    def baseline():
        if not dynamic:
            return torch_int8_quant(x, s)
        amax = x.float().abs().amax(dim=-1, keepdim=True)
        return (x * (amax / 127.0).reciprocal()).clamp(-128, 127).to(torch.int8)

    ours = lambda: scaled_int8_quant(x, s)  # noqa: E731
    if compile_conch and not dynamic:  # (the dynamic op has no custom-op form: it is this build's extension)
        compiled = torch.compile(lambda t, sc: scaled_int8_quant(t, sc), fullgraph=True)
        ours = lambda: compiled(x, s)  # noqa: E731
    if compile_ref:
        baseline = torch.compile(baseline)
    run_pair("dynamic_scaled_int8_quant" if dynamic else "static_scaled_int8_quant", ours, baseline, params,
             iteration_time_ms, warmup_time_ms, csv, nbytes=float(x.numel() * (x.element_size() + 1)))


if __name__ == "__main__":
    main()
