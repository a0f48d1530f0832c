"""Host time of a decode-size op call (development aid): wall time per call with the GPU queue never empty, and cProfile's top
entries -- scaled_gemm 14.4 us (9 us of it the ctypes call: marshalling + hipLaunchKernel), mixed_precision_gemm 12.3 us,
scaled_int8_quant 7.8 us on the round-3 box.  usage: python tools/prof_host.py"""
import cProfile, pstats, time, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from conch_amd.ops.quantization.gemm import scaled_gemm, mixed_precision_gemm, static_quant_scaled_gemm
from conch_amd.ops.quantization.fp8 import scaled_fp8_quant
from conch_amd.ops.quantization.int8 import scaled_int8_quant
m,k,n=16,4096,4096
a=torch.randint(-32,32,(m,k),dtype=torch.int8,device='cuda'); bt=torch.randint(-32,32,(n,k),dtype=torch.int8,device='cuda')
sa=0.25*torch.rand((m,1),device='cuda'); sb=0.25*torch.rand((n,1),device='cuda')
x=(torch.rand((m,k),device='cuda')-0.3).to(torch.float16)
wq=torch.randint(-2**31,2**31-1,(k//8,n),dtype=torch.int32,device='cuda'); ws=(0.05*torch.rand((k//128,n),device='cuda')+0.01).to(torch.float16)
s1=torch.tensor([2.1],device='cuda')
def f1(): return scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
def f2(): return mixed_precision_gemm(x, wq, ws, None, 4, 8, 128)
def f3(): return scaled_int8_quant(x, s1)
def f4(): return scaled_int8_quant(x, None)
def f5(): return scaled_fp8_quant(x, s1)
def f6(): return static_quant_scaled_gemm(x, bt.T, s1, sb, torch.bfloat16)
PROFILE = '--profile' in sys.argv
for name,f in (('scaled_gemm',f1),('mixed_precision_gemm',f2),('scaled_int8_quant',f3),('scaled_int8_quant dynamic',f4),('scaled_fp8_quant',f5),('static_quant_scaled_gemm',f6)):
    for _ in range(2000): f()
    torch.cuda.synchronize()
    t=time.perf_counter()
    N=20000
    for _ in range(N): f()
    host=(time.perf_counter()-t)/N*1e6
    torch.cuda.synchronize()
    print(f'{name}: host time per call {host:.2f} us (GPU queue never empty: pure host cost if > GPU time)')
    if not PROFILE: continue
    pr=cProfile.Profile(); pr.enable()
    for _ in range(5000): f()
    pr.disable(); torch.cuda.synchronize()
    st=pstats.Stats(pr); st.sort_stats('tottime'); st.print_stats(12)
