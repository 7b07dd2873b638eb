import torch
dev='cuda:0'
M=47360
for N,K in [(2304,768),(768,768),(3072,768),(768,3072)]:
    X=(torch.randn(M,K,device=dev)*0.5).half(); W=(torch.randn(N,K,device=dev)*0.05).half()
    for _ in range(3): Y=X@W.t()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): Y=X@W.t()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    print(f'torch.matmul f16 M={M} N={N} K={K}: {ms*1000:.1f} us {2.0*M*N*K/ms/1e9:.1f} TF')
