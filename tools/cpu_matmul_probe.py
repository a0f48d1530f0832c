import torch,time,os
print("cpus", os.cpu_count(), "threads", torch.get_num_threads())
for nt in (None, 32, 8):
    if nt: torch.set_num_threads(nt)
    for dt in (torch.float16, torch.bfloat16):
        for m in (16, 64):
            a=torch.randn(m,11008).to(dt); b=torch.randn(11008,4096).to(dt)
            t=time.time(); c=a@b; print("threads",nt,dt,m, round(time.time()-t,3), flush=True)
    a=torch.randn(256,4096); b=torch.randn(4096,11008)
    t=time.time(); c=a@b; print("threads",nt,"fp32 256x4096x11008", round(time.time()-t,3), flush=True)
    a=torch.randn(1024,4096).double(); b=torch.randn(4096,4096).double()
    t=time.time(); c=a@b; print("threads",nt,"fp64 1024x4096x4096", round(time.time()-t,3), flush=True)
