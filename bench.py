#!/usr/bin/env python3
"""Benchmark of the pseudo-label hot path (BASELINE.json metric: pseudo-labeled LiDAR frames/sec,
150k-pt frames, ~60 clusters, at 1/2/4/8 MI355X).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one synthetic 150k-point frame through the whole per-frame path on every rank
(ground removal -> ref transform -> HDBSCAN -> filters -> multi-view render -> CLIP ViT-B/16 fp16 encode ->
scores -> vote -> boxes -> result dict).  Frames are sharded across ranks (weak scaling: K frames per GPU);
the only collective is ONE all-gather of the per-crop score matrices after the K frames (north_star).
Inputs are resident in HBM before the timed region starts.  Each rank keeps `--inflight` frames (default 6) in flight on worker
threads with their own streams and handles; the warm-up runs at least that many frames so that every worker handle exists before
the timed region, and the K timed frames include filling and draining that pipeline (small K therefore reads a little lower).
  default_config_mode  (N=1 only, information) the reference's default stage order -- entropy scores + two-frame clustering -- on a
                coherent synthetic sequence

The JSON line also carries
  roofline      the dominant kernel (ViT projection GEMM, k_gemm_f16_pp64): algorithmic FLOPs / launch duration, measured
                live with HIP event pairs on the launch stream (csrc/vit.hip vg_vit_profile), vs the dense fp16
                MFMA peak of /opt/skills/guides/MI355X_MICROARCH.md
  cpu_baseline  the CPU oracle (oracle/pipeline_oracle.py, kind "port": the reference cannot travel to the GPU box)
                timed on a bounded sample on the host cores of the same box (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VIT_FLOP_PER_CROP = 2 * 17_563_453_440
PEAK_F16_MFMA_TFLOPS = 2500.0                   # MI355X_MICROARCH.md: ~2.5 PF dense fp16/bf16


def cpu_baseline(n_points=20_000, n_objects=8):
    """Whole path on the host cores for one bounded frame (BASELINE config 0 shape)."""
    from oracle.pipeline_oracle import OraclePipeline
    from vilgod_amd import synthetic, clip_weights as cw
    from vilgod_amd.pipeline import default_preprocessor_cfg
    n_threads = min(16, os.cpu_count())        # more threads make the small per-cluster ops slower (measured on the 256-core box)
    torch.set_num_threads(n_threads)
    cfg = default_preprocessor_cfg()
    wd = cw.synthetic_vit_weights(0, **cw.VIT_B16)
    text = cw.synthetic_text_features(0, 24, 512)
    orc = OraclePipeline(wd, text, cfg['clip']['class_list'], cfg['clip']['class_mapping'], clusterer='sklearn')
    n_frames = 3
    frames = [synthetic.make_frame(1000 + i, n_points, n_objects=n_objects) for i in range(n_frames)]
    poses = synthetic.make_poses(n_frames + 1)
    t0 = time.perf_counter()
    crops = valid = 0
    tsum = {}
    for i in range(n_frames):
        o = orc.process_frame(frames[i], poses[i + 1], poses[0])
        crops += len(o['u8'])
        valid += int(o['valid'].sum())
        for k, v in orc.timings.items():
            tsum[k] = tsum.get(k, 0.0) + v
    dt = time.perf_counter() - t0
    return {
        'value': round(n_frames / dt, 5), 'unit': 'frames/s (20k-pt frames)', 'cores': n_threads, 'kind': 'port',
        'sample': (f'{n_frames} consecutive synthetic frames of {n_points} points ({valid} valid clusters, {crops} crops in total; '
                   f'BASELINE config 0 shape -- a 150k-pt frame carries ~5x the crops and ~8x the points), all stages, '
                   f'{dt:.1f} s: ' + ', '.join(f'{k} {v:.2f}s' for k, v in tsum.items()) +
                   '; clustering = sklearn.cluster.HDBSCAN stand-in (the reference\'s hdbscan package is absent), '
                   'ViT = torch-CPU fp32, same synthetic weights'),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=96)
    ap.add_argument('--warmup', type=int, default=12)
    ap.add_argument('--points', type=int, default=150_000)
    ap.add_argument('--objects', type=int, default=60)
    ap.add_argument('--views', type=int, default=4)
    ap.add_argument('--dtype', default='f16', choices=['f16', 'f32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--stage-times', action='store_true', help='print per-stage ms (adds synchronisation; not for the metric)')
    ap.add_argument('--no-roofline-pass', action='store_true', help='skip the sequential GEMM-timing pass (profiling runs)')
    ap.add_argument('--no-sequence-pass', action='store_true', help='skip the extra (untimed-for-the-metric) pass in the reference\'s default stage order')
    ap.add_argument('--inflight', type=int, default=6, help='frames in flight per GPU (worker streams); 1 = strictly sequential')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('VILGOD_DIST_BACKEND', 'nccl')       # 'gloo' only for the 2-ranks-on-one-GPU self test
        kw = {'device_id': torch.device(f'cuda:{local_rank}')} if backend == 'nccl' else {}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    dev = torch.device(f'cuda:{local_rank}')
    torch.cuda.set_device(dev)

    from vilgod_amd import synthetic
    from vilgod_amd.pipeline import PseudoLabelPipeline
    pipe = PseudoLabelPipeline(device=dev, vit_dtype=args.dtype, n_views=args.views, max_points=args.points + 1024,
                               clip_model_path='/nonexistent')
    # a short synthetic sequence per rank: frames differ, poses follow a smooth trajectory; resident in HBM
    n_distinct = 4
    poses = synthetic.make_poses(args.steps + max(args.warmup, args.inflight) + 8, seed=rank)   # warm-up covers the worker handles
    frames = [pipe.upload(synthetic.make_frame(1 + rank * 100 + i, args.points, n_objects=args.objects)) for i in range(n_distinct)]
    torch.cuda.synchronize()

    inflight = 1 if args.stage_times else max(1, args.inflight)

    def run_steps(first, count):
        """`count` frames (steps) starting at step index `first`; returns [(FrameState, result, probs)]."""
        idx = list(range(first, first + count))
        if inflight == 1:
            out = []
            for i in idx:
                fs, res = pipe.process_frame(frames[i % n_distinct], poses[i + 1], poses[0], fnr=i, timing=args.stage_times)
                out.append((fs, res, pipe.last_probs))
                for k, v in pipe.timings.items():
                    stage[k] = stage.get(k, 0.0) + v
            return out
        return pipe.process_frames([frames[i % n_distinct] for i in idx], [poses[i + 1] for i in idx], poses[0],
                                   n_workers=inflight, first_fnr=first)

    stage = {}
    pipe.new_sequence()
    run_steps(0, max(args.warmup, inflight if inflight > 1 else 0))        # also builds the worker handles
    stage = {}
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    crops = clusters = labelled = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    score_mats = []
    for fs, res, probs in run_steps(args.warmup, args.steps):
        score_mats.append(probs)
        crops += probs.shape[0]
        clusters += fs.n_detections
        labelled += len(res['name'])
    # the one collective of the path: all-gather of the per-crop score matrices (padded to a common length)
    scores = torch.cat(score_mats) if score_mats else torch.zeros((0, 24), device=dev)
    if dist is not None:
        cdev = 'cpu' if dist.get_backend() == 'gloo' else dev
        n_loc = torch.tensor([scores.shape[0]], device=cdev, dtype=torch.int64)
        n_all = [torch.zeros_like(n_loc) for _ in range(world)]
        dist.all_gather(n_all, n_loc)
        mx = int(max(int(x.item()) for x in n_all))
        pad = torch.zeros((mx, scores.shape[1]), device=dev, dtype=scores.dtype)
        pad[:scores.shape[0]] = scores
        gathered = torch.empty((world * mx, scores.shape[1]), device=dev, dtype=scores.dtype)
        if dist.get_backend() == 'gloo':                  # host-staged in the self test; RCCL gathers device tensors directly
            parts = [torch.empty_like(pad.cpu()) for _ in range(world)]
            dist.all_gather(parts, pad.cpu())
            gathered = torch.cat(parts).to(dev)
        else:
            dist.all_gather_into_tensor(gathered, pad)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device='cpu' if (dist is not None and dist.get_backend() == 'gloo') else dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # roofline of the dominant kernel: HIP event pairs around every k_gemm_f16_pp64 launch on its launch stream, over a
    # SEQUENTIAL pass (1 frame in flight) of the same workload right after the timed region -- with several frames in
    # flight the pairs would also span other streams' kernels and stop measuring this kernel.
    launches = gemm_ms = gemm_flops = all_launches = all_ms = all_flops = 0
    if not args.no_roofline_pass:
        n_pass = min(4, args.steps)
        pipe.clip.encoder.profile(True)
        for i in range(n_pass):
            pipe.process_frame(frames[i % n_distinct], poses[i + 1], poses[0], fnr=i)
        launches, gemm_ms, gemm_flops = pipe.clip.encoder.profile_read(kind=1)        # the dominant kernel alone
        all_launches, all_ms, all_flops = pipe.clip.encoder.profile_read(kind=-1)
        pipe.clip.encoder.profile(False)

    if rank == 0:
        frames_total = world * args.steps
        value = frames_total / elapsed
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'gemm_traffic.json')       # tools/collect_profiles.sh + summarize_profiles.py
        if os.path.exists(tpath):
            traffic = round(json.load(open(tpath))['hbm_bytes_per_launch'])
        out = {
            'metric': 'pseudo-labeled LiDAR frames/sec (150k pts, ~60 clusters)',
            'value': round(value, 3), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1000.0 * elapsed / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f16' if args.dtype == 'f16' else 'f32', 'data': 'synthetic',
            'config': {
                'workload': (f'full per-frame path (ground removal, HDBSCAN, filters, {args.views}-view render, CLIP ViT-B/16 '
                             f'{args.dtype} encode, scores, vote, boxes) on synthetic {args.points}-pt frames, '
                             f'{args.objects} objects (BASELINE config 3 shape), frames sharded {world}-way, '
                             'one all-gather of the score matrices'),
                'points_per_frame': args.points, 'views': args.views, 'frames_per_gpu': args.steps,
                'clusters_per_frame': round(clusters / max(args.steps, 1), 1),
                'crops_per_frame': round(crops / max(args.steps, 1), 1),
                'labelled_per_frame': round(labelled / max(args.steps, 1), 1),
                'weights': pipe.clip.weights_source, 'parallelism': f'frame-sharded x{world}', 'frames_in_flight_per_gpu': inflight,
            },
            'roofline': {
                'kernel': 'k_gemm_f16_pp64 (every ViT projection GEMM: in_proj, out_proj, c_fc, c_proj, patch embedding)',
                'bound': 'mfma', 'achieved': round(achieved, 1), 'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(achieved / PEAK_F16_MFMA_TFLOPS, 4), 'traffic': traffic, 'traffic_unit': 'HBM bytes per launch (PMC)',
                'method': 'HIP event pairs on the launch stream around every k_gemm_f16_pp64 launch, sequential pass (1 frame in flight) of the same frames right after the timed region',
                'launches': launches, 'avg_launch_us': round(1000.0 * gemm_ms / max(launches, 1), 2),
                'algorithmic_flops_per_launch': round(gemm_flops / max(launches, 1)),
                'gemm_ms_per_frame': round(gemm_ms / max(min(4, args.steps), 1), 3),
                'all_projection_gemms': {'kernels': 'k_gemm_f16_pp64 (+ k_gemm_f16_pp16 / k_gemm_f16 for shapes it does not take: none in ViT-B/16)', 'launches': all_launches,
                                         'achieved': round(all_flops / (all_ms * 1e-3) / 1e12, 1) if all_ms > 0 else 0.0,
                                         'ms_per_frame': round(all_ms / max(min(4, args.steps), 1), 3)},
            },
        }
        if args.stage_times:
            out['stage_ms_per_frame'] = {k: round(1000.0 * v / args.steps, 3) for k, v in stage.items()}
        # the two extra passes must never cost the metric line: failures are reported inside the JSON
        try:
            if world == 1 and not args.no_sequence_pass and not args.stage_times:
                # additional information, not the metric: the reference's DEFAULT stage order (preprocessing.yaml:50-68 -- entropy
                # scores over a 15-frame window + two-frame 5-D clustering) on one coherent synthetic sequence
                n_seq = max(48, args.steps)
                sframes, sposes = synthetic.make_sequence(seed=0, n_frames=n_seq, n_points=args.points, n_objects=args.objects)
                sframes = [pipe.upload(f) for f in sframes]
                pipe.process_sequence(sframes[:4], sposes[:4], sposes[0], n_workers=inflight)
                torch.cuda.synchronize()
                ts = time.perf_counter()
                sres = pipe.process_sequence(sframes, sposes, sposes[0], n_workers=inflight)
                torch.cuda.synchronize()
                ts = time.perf_counter() - ts
                out['default_config_mode'] = {
                    'value': round(n_seq / ts, 3), 'unit': 'frames/s', 'frames': n_seq,
                    'workload': ('mask_ground_points -> calculate_entropy_scores (15-frame window, skip 1) -> spatial_clustering n_frames=2 '
                                 '(5-D HDBSCAN + nearest-label transfer) -> filter -> classification -> boxes on one coherent '
                                 f'synthetic sequence of {n_seq} frames x {args.points} points'),
                    'labelled_per_frame': round(sum(len(r[1]['name']) for r in sres) / n_seq, 1),
                    'moving_clusters_per_frame': round(sum(int((~r[0].static).sum()) for r in sres) / n_seq, 1)}
            if world == 1 and not args.no_cpu_baseline:
                out['cpu_baseline'] = cpu_baseline()
        except Exception as e:          # noqa: BLE001
            out.setdefault('extras_error', f'{type(e).__name__}: {e}')
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
