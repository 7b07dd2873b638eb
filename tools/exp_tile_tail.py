"""Does the partial last round of GEMM tiles cost time?  ViT-B/16 encode of n crops (n * 197 tokens -> ceil / 256 row tiles; the residual
GEMMs have 3 column tiles: 256 row tiles = 768 tiles = exactly 3 rounds of 256 CUs, 257 row tiles start a 4th round with 3 tiles):
time per crop for crop counts around the boundary (332 crops = 256 row tiles, 333 = 257), with 1 and 2 encodes in flight (the pipeline
runs at most two ViT passes at a time)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch
from vilgod_amd import clip_weights as cw
from vilgod_amd.clip_wrapper import VitEncoder
dev = torch.device('cuda:0')
enc = VitEncoder(cw.synthetic_vit_weights(0, **cw.VIT_B16), dtype='f16', device=dev)
views = [enc.view(), enc.view()]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
REP = int(os.environ.get('REP', '12'))
counts = [int(c) for c in os.environ.get('COUNTS', '300,320,326,332,333,338,345,360,384,390').split(',')]
def run(n, inflight):
    rows = (n * 196 + 255) // 256 * 256
    p = [(torch.randint(0, 256, (rows, 256), device=dev).float() / 256).half() for _ in range(inflight)]
    def loop(k):
        with torch.cuda.stream(streams[k]):
            for _ in range(REP):
                views[k].encode_patches(p[k], n)
            streams[k].synchronize()
    for k in range(inflight): loop(k)          # warm-up (workspaces)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=loop, args=(k,)) for k in range(inflight)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (REP * inflight)
for rnd in range(2):
    for n in counts:
        a, b = run(n, 1), run(n, 2)
        tiles = (n * 197 + 255) // 256
        print(f'round {rnd} crops {n:4d} row tiles {tiles:4d} (x3 = {3*tiles:4d} = {3*tiles/256:.2f} rounds): 1 in flight {1000*a:7.2f} ms = {1e6*a/n:6.2f} us/crop;  2 in flight {1000*b:7.2f} ms = {1e6*b/n:6.2f} us/crop', flush=True)
