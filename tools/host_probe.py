import time, sys, ctypes
sys.path.insert(0, '/root/repo')
import torch
from conch_amd import _C
from conch_amd.ops.quantization.gemm import scaled_gemm, create_scaled_metadata
from conch_amd.kernels.quantization import gemm as kg
m,k,n=16,4096,4096
a=torch.randint(-32,32,(m,k),dtype=torch.int8,device='cuda'); bt=torch.randint(-32,32,(n,k),dtype=torch.int8,device='cuda'); b=bt.T
sa=0.25*torch.rand((m,1),device='cuda'); sb=0.25*torch.rand((n,1),device='cuda')
out=torch.empty((m,n),dtype=torch.bfloat16,device='cuda')
lib=_C.load(); fn=lib.conch_scaled_gemm
stream=_C.current_stream_handle(a.device)
args=(out.data_ptr(),a.data_ptr(),b.data_ptr(),sa.data_ptr(),sb.data_ptr(),None,m,n,k,a.stride(0),a.stride(1),b.stride(0),b.stride(1),out.stride(0),out.stride(1),sa.numel(),sb.numel(),_C.dtype_id(a.dtype),_C.dtype_id(out.dtype),stream)
def t(name,f,N=20000):
    for _ in range(2000): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(N): f()
    dt=(time.perf_counter()-t0)/N*1e6; torch.cuda.synchronize()
    print(f'{name:50s} {dt:6.2f} us/call', flush=True)
t('op scaled_gemm (full path)', lambda: scaled_gemm(a,b,sa,sb,torch.bfloat16))
t('raw ctypes call, precomputed args', lambda: fn(*args))
t('torch.empty((m,n))', lambda: torch.empty((m,n),dtype=torch.bfloat16,device=a.device))
t('a.new_empty((m,n))', lambda: a.new_empty((m,n),dtype=torch.bfloat16))
t('create_scaled_metadata', lambda: create_scaled_metadata(a,b,sa,sb,torch.bfloat16))
t('6x data_ptr', lambda: (out.data_ptr(),a.data_ptr(),b.data_ptr(),sa.data_ptr(),sb.data_ptr()))
t('current_stream_handle', lambda: _C.current_stream_handle(a.device))
t('reshape(-1) x2', lambda: (sa.reshape(-1), sb.reshape(-1)))
t('empty python call', lambda: None)
meta=create_scaled_metadata(a,b,sa,sb,torch.bfloat16)
t('launcher only (out given)', lambda: kg.scaled_gemm_launcher(out,a,b,sa,sb,meta))
# tiny kernel launch via torch for comparison
t('torch: out.zero_() (one tiny kernel)', lambda: sa.zero_())
