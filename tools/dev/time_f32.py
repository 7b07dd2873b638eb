import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vilgod_amd import clip_weights as cw
from vilgod_amd.clip_wrapper import VitEncoder
dev = torch.device('cuda:0')
enc = VitEncoder(cw.synthetic_vit_weights(0, **cw.VIT_B16), dtype='f32', device=dev)
x = torch.randn(64, 3, 224, 224, device=dev)
for mode in ('1', '0', '1', '0'):
    os.environ['VG_GEMM_F32_MFMA'] = mode
    enc.encode(x); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): f = enc.encode(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f'fp32 tower, 64 crops, MFMA={mode}: {1000*dt:.1f} ms = {1e6*dt/64:.0f} us per crop ({64*35.127e9/dt/1e12:.1f} TFLOP/s over the whole tower)', flush=True)
