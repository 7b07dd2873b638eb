"""Is it the other stream's KERNEL BOUNDARIES that slow the projection GEMMs down?  (round 5)
tools/exp_interference.py: one frame's MST beside a stream of c_fc-shaped GEMMs costs the GEMMs 1.35 ms, 0.75 ms of it even when the MST is
confined to CUs the GEMMs never use -- and, round 4, "while ANY kernel of the MST is running the GEMM stream advances at ~63 % of its speed,
whatever that kernel's workgroup count".  An MST is 124 dependent launches; every dependent launch of a stream ends / begins with cache
maintenance (release / acquire at agent scope; the eight XCD L2s are not coherent with each other).  This tool runs the same 300 GEMMs
beside a second stream that does NOTHING but launch trivial kernels back to back:
    a  one-workgroup kernels that touch 4 bytes (vg_cluster_rot with one angle), dependent (same stream)
    b  the same, from TWO extra streams
    c  79k-thread kernels that touch 0.3 MB (torch fill of an int32 array), dependent
and prints the GEMM time lost per 100 launches of the other stream.    python tools/exp_launch_noise.py"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import numpy as np, torch
from vilgod_amd._lib import lib, ptr, check
dev = torch.device('cuda:0')
M, N, K = 66_560, 3072, 768
A = (torch.randn(M, K, device=dev) * 0.5).half()
W = (torch.randn(N, K, device=dev) * 0.05).half()
b = torch.randn(N, device=dev)
C = torch.empty(M, N, dtype=torch.float16, device=dev)
NG = 300
sa = torch.cuda.Stream()
others = [torch.cuda.Stream() for _ in range(2)]
angle = torch.zeros(1, device=dev)
rot = torch.zeros(1, 6, dtype=torch.float64, device=dev)
big = torch.zeros(79_000, dtype=torch.int32, device=dev)


def gemms():
    with torch.cuda.stream(sa):
        for _ in range(NG):
            check(lib.vg_gemm(1, 1, ptr(A), ptr(W), ptr(b), ptr(C), None, M, N, K, sa.cuda_stream))
        sa.synchronize()


def noise(kind, stream, stop, count):
    with torch.cuda.stream(stream):
        while not stop.is_set():
            for _ in range(50):
                if kind == 'tiny':
                    lib.vg_cluster_rot(ptr(angle), 1, ptr(rot), stream.cuda_stream)
                else:
                    big.fill_(1)
            count[0] += 50
            stream.synchronize()


def timed(kind, n_streams):
    stop = threading.Event()
    counts = [[0] for _ in range(n_streams)]
    ths = [threading.Thread(target=noise, args=(kind, others[i], stop, counts[i])) for i in range(n_streams)]
    for t in ths:
        t.start()
    time.sleep(0.05)
    torch.cuda.synchronize()
    n0 = sum(c[0] for c in counts)
    t0 = time.perf_counter()
    gemms()
    dt = time.perf_counter() - t0
    n1 = sum(c[0] for c in counts)
    stop.set()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    return dt, n1 - n0


gemms()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); gemms(); alone = time.perf_counter() - t0
    print(f'alone: {NG} GEMMs {1e3 * alone:.1f} ms ({1e6 * alone / NG:.1f} us each)')
    for name, kind, ns in (('a: one-workgroup kernels, one stream', 'tiny', 1), ('b: one-workgroup kernels, two streams', 'tiny', 2),
                           ('c: 79k-thread fills, one stream', 'fill', 1)):
        dt, n = timed(kind, ns)
        print(f'{name}: {1e3 * dt:.1f} ms beside {n} launches of the other stream(s) ({1e6 * dt / max(n, 1):.2f} us of wall per launch): '
              f'{1e3 * (dt - alone) / max(n, 1) * 100:.3f} ms of GEMM time lost per 100 launches ({100 * (dt / alone - 1):+.1f} %)', flush=True)
