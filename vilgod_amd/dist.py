"""Frame-level data parallelism over the GPUs of one node: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU tests), frames of a sequence in
contiguous blocks per rank, ONE collective on the data path: the all-gather of the per-crop score matrices."""
import os

import numpy as np
import torch


def world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend=None):
    """Initialise the process group from RANK / WORLD_SIZE / MASTER_* if launched by torch.distributed.run."""
    import torch.distributed as dist
    ws = int(os.environ.get('WORLD_SIZE', 1))
    if ws <= 1 or dist.is_initialized():
        return world()
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
    kw = {}
    if backend == 'nccl':
        kw['device_id'] = torch.device(f"cuda:{int(os.environ.get('LOCAL_RANK', 0))}")
    dist.init_process_group(backend=backend, rank=int(os.environ['RANK']), world_size=ws, **kw)
    return world()


def shard_frames(n_frames, rank, world_size):
    """Contiguous block of frame numbers for `rank` (blocks keep a rank's frames adjacent in time)."""
    base, rem = divmod(n_frames, world_size)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def gather_scores(local, n_classes=24, device=None):
    """All-gather of per-frame score matrices.  local: {fnr: tensor [n_crops_f, K] float32}.
    Returns {fnr: tensor} with every rank's frames (on `device`).  Two collectives on a padded slab:
    counts, then scores (north_star: "RCCL all-gather ... only for the final cosine scores")."""
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1:
        return dict(local)
    frames = sorted(local)
    device = device or (next(iter(local.values())).device if local else torch.device('cpu'))
    out_device = device
    if dist.get_backend() == 'gloo':
        device = torch.device('cpu')           # CPU tests / single-GPU multi-process runs
    meta = torch.tensor([[f, local[f].shape[0]] for f in frames], dtype=torch.int64, device=device).reshape(-1, 2)
    n_loc = torch.tensor([meta.shape[0], int(meta[:, 1].sum()) if len(frames) else 0], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n_loc) for _ in range(ws)]
    dist.all_gather(sizes, n_loc)
    max_f = max(int(s[0]) for s in sizes)
    max_c = max(int(s[1]) for s in sizes)
    meta_pad = torch.full((max(max_f, 1), 2), -1, dtype=torch.int64, device=device)
    meta_pad[:meta.shape[0]] = meta
    metas = [torch.empty_like(meta_pad) for _ in range(ws)]
    dist.all_gather(metas, meta_pad)
    slab = torch.zeros((max(max_c, 1), n_classes), dtype=torch.float32, device=device)
    if frames:
        cat = torch.cat([local[f].to(device=device, dtype=torch.float32) for f in frames])
        slab[:cat.shape[0]] = cat
    slabs = [torch.empty_like(slab) for _ in range(ws)]
    dist.all_gather(slabs, slab)
    out = {}
    for r in range(ws):
        off = 0
        for f, c in metas[r].tolist():
            if f < 0:
                continue
            out[int(f)] = slabs[r][off:off + c].to(out_device)
            off += c
    return out


def gather_objects(obj):
    """Small python objects (per-frame result dicts / serialised frame states) to every rank."""
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1:
        return [obj]
    out = [None] * ws
    dist.all_gather_object(out, obj)
    return out


def _state_device(device=None):
    import torch.distributed as dist
    return torch.device('cpu') if dist.get_backend() == 'gloo' else (device or torch.device('cuda', torch.cuda.current_device()))


def recv_ground_state(ground_model, device=None):
    """Rank r > 0: wait for the Patchwork++ state after the last frame of rank r - 1's block and set it (point to point, ~131 KB)."""
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1 or rank == 0:
        return
    from ._lib import lib
    buf = torch.empty(int(lib.vg_ground_state_bytes()), dtype=torch.uint8, device=_state_device(device))
    dist.recv(buf, src=rank - 1)
    ground_model.set_state(buf.cpu().numpy().tobytes())


def send_ground_state(ground_model, device=None):
    """Rank r < N - 1: export the state after this rank's last ground pass (synchronises the caller's stream) and send it on."""
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1 or rank == ws - 1:
        return
    blob = torch.frombuffer(bytearray(ground_model.export_state()), dtype=torch.uint8).to(_state_device(device))
    dist.send(blob, dst=rank + 1)


def chain_ground_state(ground_model, run_my_block, device=None):
    """Patchwork++'s adaptive state runs through the whole sequence (SURVEY 8e exception 1).  With contiguous frame blocks the
    state is HANDED from rank to rank instead of every rank replaying the frames before its block: rank r waits for the state
    after frame start_r - 1 from rank r - 1, sets it, runs the ground stage over its OWN block (`run_my_block()`), and passes
    the state on to rank r + 1 before it starts the heavy stages.  Same masks and same final state as one sequential pass
    (tests/test_cli.py, tests/test_ground.py)."""
    recv_ground_state(ground_model, device)
    out = run_my_block()
    send_ground_state(ground_model, device)
    return out


def relay_recv_state(ground_model, g, device=None):
    """Round-robin frames with the ground state RELAYED frame by frame (bench.py --ground-handoff relay; SURVEY 8e exception 1): frame
    g belongs to rank g % N, and before its ground pass that rank takes the Patchwork++ state behind frame g - 1 from rank (g - 1) % N
    (point to point, ~131 KB).  Every rank runs ONLY its own ground passes -- against `replicate`, where every rank runs all N K of them:
    measured by one rank on one GPU (bench.py multi_gpu_model.measured_single_rank_emulation) that costs 1.3 / 2.7 / 6.2 % at N = 2 / 4 /
    8 -- at the price of a chain of N hand-offs per round of frames (0.46 ms pass + transfer each, far below a 14 ms frame)."""
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1 or g == 0:
        return
    from ._lib import lib
    buf = torch.empty(int(lib.vg_ground_state_bytes()), dtype=torch.uint8, device=_state_device(device))
    dist.recv(buf, src=(g - 1) % ws)
    ground_model.set_state(buf.cpu().numpy().tobytes())


def relay_send_state(ground_model, g, n_total, device=None):
    """... and behind its ground pass of frame g (synchronises the caller's stream) it sends the state on to the owner of frame g + 1."""
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1 or g >= n_total - 1:
        return
    blob = torch.frombuffer(bytearray(ground_model.export_state()), dtype=torch.uint8).to(_state_device(device))
    dist.send(blob, dst=(g + 1) % ws)
