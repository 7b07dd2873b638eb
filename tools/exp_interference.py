"""What does one clustering pass cost a projection GEMM that runs beside it?  Stream A: back-to-back c_fc-shaped launches (M = 66 560,
N = 3072, K = 768, QuickGELU epilogue); stream B (another thread): the MST of one 150k-point frame's non-ground points, again and
again.  Prints the time of 300 GEMM launches alone and with B running, the MSTs B finished meanwhile, and the GEMM time lost per MST --
with both streams unrestricted, and with B restricted to r CUs of every XCD (vilgod_amd/streams.py) while A runs on the other CUs
('complement') or on all of them ('all'):    python tools/exp_interference.py [r ...]      (default sweep 1 2 3 4)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vilgod_amd import synthetic
from vilgod_amd._lib import lib, ptr, check
from vilgod_amd.pipeline import PseudoLabelPipeline
dev = torch.device('cuda:0')
pipe = PseudoLabelPipeline(device=dev, max_points=160_000, clip_model_path='/nonexistent')
pts = pipe.upload(synthetic.make_frame(1, 150_000))
mask = pipe.ground(pts)
X = pipe.to_ref(pts, np.eye(4))[mask == 0].contiguous()
M, N, K = 66_560, 3072, 768
A = (torch.randn(M, K, device=dev) * 0.5).half()
W = (torch.randn(N, K, device=dev) * 0.05).half()
b = torch.randn(N, device=dev)
C = torch.empty(M, N, dtype=torch.float16, device=dev)
NG = 300
from vilgod_amd.streams import make_streams
reserves = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
configs = [('unrestricted', torch.cuda.Stream(), torch.cuda.Stream())]
for r in reserves:
    for tower in ('complement', 'all'):
        front, tow = make_streams(dev, r, tower)
        configs.append((f'reserve {r}/XCD, GEMMs on {tower}', tow(0), front(0)))
sa = sb = None


def gemms():
    with torch.cuda.stream(sa):
        for _ in range(NG):
            check(lib.vg_gemm(1, 1, ptr(A), ptr(W), ptr(b), ptr(C), None, M, N, K, sa.cuda_stream))
        sa.synchronize()


def timed(with_b):
    stop, done = threading.Event(), [0]

    def loop():
        with torch.cuda.stream(sb):
            while not stop.is_set():
                pipe.cluster_model.mst(X)
                done[0] += 1
    th = threading.Thread(target=loop)
    if with_b:
        th.start()
        time.sleep(0.05)
    torch.cuda.synchronize()
    n0 = done[0]
    t0 = time.perf_counter()
    gemms()
    dt = time.perf_counter() - t0
    n1 = done[0]
    if with_b:
        stop.set(); th.join()
    torch.cuda.synchronize()
    return dt, n1 - n0


for rep in range(2):
  for name, sa, sb in configs:
    gemms()
    t_alone, _ = timed(False)
    t_with, n = timed(True)
    with torch.cuda.stream(sb):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            pipe.cluster_model.mst(X)
        torch.cuda.synchronize(); t_mst = (time.perf_counter() - t0) / 5
    print(f'{name}: MST alone {1e3 * t_mst:.2f} ms; {NG} GEMMs alone {1e3 * t_alone:.1f} ms ({1e6 * t_alone / NG:.1f} us each), beside the MST loop {1e3 * t_with:.1f} ms; '
          f'{n} MSTs finished meanwhile ({1e3 * t_with / max(n, 1):.2f} ms each): {1e3 * (t_with - t_alone) / max(n, 1):.3f} ms of GEMM time lost per MST', flush=True)
