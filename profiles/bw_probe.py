import torch, time
dev=torch.device("cuda",0)
for mb in (77, 154, 308, 1024):
    n = mb*1024*1024//2
    x=torch.empty(n,dtype=torch.float16,device=dev).normal_(); y=torch.empty_like(x)
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    print(f"copy {mb} MB: {ms*1e3:.1f} us  {2*mb/1024/ms*1e3/1e3:.2f} TB/s (r+w)")
