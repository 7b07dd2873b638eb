"""Two encodes in flight: does it matter HOW FAR APART they are?  (round 5)  The pipeline lets two ViT passes run at a time in arrival order;
their relative phase is whatever the frames' front stages produce.  Here two streams each run REP back-to-back encodes of n crops and the second
stream starts `offset` of an encode after the first (both loops have the same length, so the offset persists): time per encode for offsets
0, 1/8 ... 1/2 of an encode, and for THREE streams a third apart."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch
from vilgod_amd import clip_weights as cw
from vilgod_amd.clip_wrapper import VitEncoder
dev = torch.device('cuda:0')
enc = VitEncoder(cw.synthetic_vit_weights(0, **cw.VIT_B16), dtype='f16', device=dev)
views = [enc.view() for _ in range(3)]
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
REP = int(os.environ.get('REP', '10'))
n = int(os.environ.get('CROPS', '333'))
rows = (n * 196 + 255) // 256 * 256
p = [(torch.randint(0, 256, (rows, 256), device=dev).float() / 256).half() for _ in range(3)]


def loop(k, reps, delay):
    if delay:
        time.sleep(delay)
    with torch.cuda.stream(streams[k]):
        for _ in range(reps):
            views[k].encode_patches(p[k], n)
        streams[k].synchronize()


def run(delays):
    for k in range(len(delays)):
        loop(k, 1, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=loop, args=(k, REP, d)) for k, d in enumerate(delays)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0 - max(delays)) / (REP * len(delays))     # (the tail of the delayed stream runs alone for `delay`: a small bias against offsets)


one = run([0.0])
print(f'{n} crops: one encode at a time {1e3 * one:.2f} ms')
for rnd in range(2):
    for frac in (0.0, 0.125, 0.25, 0.375, 0.5):
        t = run([0.0, frac * one])
        print(f'round {rnd}: two in flight, second stream {frac:5.3f} of an encode behind: {1e3 * t:.2f} ms per encode = {1e6 * t / n:.2f} us per crop', flush=True)
    t = run([0.0, one / 3, 2 * one / 3])
    print(f'round {rnd}: three in flight, a third apart: {1e3 * t:.2f} ms per encode = {1e6 * t / n:.2f} us per crop', flush=True)
