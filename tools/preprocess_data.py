#!/usr/bin/env python3
"""Pseudo-label generation entry point (MI355X build) -- same command line as the reference's
tools/preprocess_data.py (README.md:128-144):

    cd tools && python preprocess_data.py preprocessor=waymo [key=value ...]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 preprocess_data.py preprocessor=waymo

Per sequence it writes the reference's two pickle families:
    <paths.sequence_data>/<seq>.pkl                                   list of per-frame state dicts
    <paths.results>/<results_folder>/<'_'.join(pipeline_active)>/<seq>.pkl  and  <seq>_indices.pkl
With several processes the frames of each sequence are sharded over the GPUs (vilgod_amd/dist.py); rank 0 writes.
"""
import gc
import logging
import os
import pickle
import random
import sys
import time
from pathlib import Path

os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')      # one hardware queue per stream in flight (vilgod_amd/__init__.py), before the runtime starts

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from vilgod_amd import config as vconfig          # noqa: E402
from vilgod_amd import dist as vdist              # noqa: E402


# timings of the last main() call (bench.py's cli_mode block reads them): per sequence wall seconds from "sequence selected" to
# "both pickle families written" and the per-stage ms per frame the dispatcher measured
LAST_RUN = {'sequences': []}


def set_random_seed(seed):
    """src/utils/common_utils.py:13-19."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    LAST_RUN['sequences'] = []
    LAST_RUN['tail_seconds_under_generator'] = 0.0
    logging.basicConfig(level=logging.INFO, format='[%(asctime)s][%(levelname)s] - %(message)s', stream=sys.stdout)
    logger = logging.getLogger('preprocess_data')
    cfg = vconfig.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'configs'), 'preprocessing', argv)
    logger.info('Working directory: {}'.format(os.getcwd()))
    if cfg.get('random_seed', False):
        set_random_seed(cfg.random_seed)

    # device.processes_per_gpu = P > 1 (whole sequences per rank only): P ranks share one GPU, rank r works on cuda:(LOCAL_RANK // P).  A
    # sequence's host-only stages (track boxes, label propagation, result dicts: ~2.5 of ~18 ms per frame, under one interpreter lock)
    # then run while the OTHER process's frame pass keeps the GPU busy.  Such ranks never exchange device memory: the process group is gloo.
    ppg = max(1, int(cfg.get('device', {}).get('processes_per_gpu', 1)))
    if ppg > 1 and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        # started without a launcher: this process (which has not touched the GPU) starts the P x n_gpus ranks as children and waits
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(('127.0.0.1', 0))
            port = so.getsockname()[1]
        n_ranks = ppg * max(1, torch.cuda.device_count())
        logger.info(f'device.processes_per_gpu={ppg}: starting {n_ranks} ranks through torch.distributed.run')
        rc = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n_ranks}', '--master-addr', '127.0.0.1',
                             '--master-port', str(port), os.path.abspath(__file__)] + list(argv)).returncode
        if rc != 0:
            raise SystemExit(rc)
        return None
    rank, world = vdist.init_from_env(os.environ.get('VILGOD_DIST_BACKEND') or ('gloo' if ppg > 1 else None))
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)) // ppg)

    dataset = vconfig.instantiate(cfg.dataset_class, logger=logger, training=True,
                                  start_sequence=cfg.start_sequence, end_sequence=cfg.end_sequence)
    if cfg.split != 'train':
        dataset.set_split(cfg.split)
    dataset.training = False

    from vilgod_amd.zero_shot_detector import ZeroShotDetector
    from vilgod_amd.pipeline import PseudoLabelPipeline
    dev = cfg.get('device', {})
    ga = [t for t in cfg.pipeline if t['name'] == 'mask_ground_points'][0]['args']
    # the cluster model, the projection and the CLIP tower are built once and shared by all sequences
    # (tools/preprocess_data.py:42-48 of the reference builds cluster_model and clip_model the same way)
    pipeline = PseudoLabelPipeline(cfg.preprocessor, device=f'cuda:{torch.cuda.current_device()}',
                                   vit_dtype=dev.get('vit_dtype', 'f16'), n_views=dev.get('n_views', 4),
                                   max_points=dev.get('max_points', 300_000), clip_model_path=cfg.paths.clip_model,
                                   min_range=ga['min_range'], z_offset=ga['z_offset'], plane_seed=dev.get('plane_seed', 666),
                                   box_mode=dev.get('box_mode', 'reference'), box_workers=dev.get('box_workers', 4),
                                   angle_mode=dev.get('angle_mode', 'reference'), vit_graph=bool(dev.get('vit_graph', False)),
                                   cu_reserve=int(dev.get('cu_reserve', 0)), cu_tower=dev.get('cu_tower', 'complement'))
    logger.info(f'CLIP weights: {pipeline.clip.weights_source}')

    result_path = Path(cfg.paths.results) / cfg.results_folder / '_'.join(cfg.pipeline_active)
    if rank == 0:
        result_path.mkdir(parents=True, exist_ok=True)
    logger.info('_' * 40)
    logger.info('Pipeline:')
    for i, name in enumerate([t['name'] for t in cfg.pipeline if t['name'] in cfg.pipeline_active], 1):
        logger.info(f'[{i}] {name}')
    logger.info('_' * 40)

    indices, detection_results = [], []
    result_data = None
    if cfg.load_detection_results and Path(cfg.result_path).exists():       # preprocess_data.py:66-70: evaluate stored results only
        with Path(cfg.result_path).open('rb') as f:
            result_data = pickle.load(f)
    # several ranks: device.shard = frames (default; north_star's configuration: every sequence's frames in contiguous blocks over the
    # ranks, Patchwork++ state handed down the rank chain, one all-gather of the scores per sequence) or sequences (rank r takes
    # sequences r, r + N, ...: no exchange at all until the final evaluation -- the better choice for data sets of many sequences)
    # auto (default): sequences when there are at least as many sequences as ranks, else frames.  Sharding the frames of one sequence
    # leaves its sequence-level stages -- track_clusters, the tracked box fit, propagate_labels, the state pickle: ~5 of ~25 ms per
    # frame on one GPU -- replicated on every rank (Amdahl: ~3.5x at 8 GPUs); whole sequences per rank have none.
    shard = dev.get('shard', 'auto')
    if shard == 'auto':
        names = getattr(dataset, 'sequence_names', None)
        shard = 'sequences' if (world > 1 and names is not None and len(names) >= world) else 'frames'
    # the stage dispatcher reads the RESOLVED choice: written back into the config itself (cfg.get('device', {}) is a detached dict
    # when the config has no device section, and the dispatcher's own default is 'auto', which it must never resolve on its own)
    try:
        dev['shard'] = shard
        cfg['device'] = dev
    except Exception:           # noqa: BLE001  (a read-only config object)
        pass
    by_sequence = world > 1 and shard == 'sequences'
    if ppg > 1 and world > 1 and not by_sequence:
        raise ValueError(f'device.processes_per_gpu={ppg} shares a GPU between ranks that work on DIFFERENT sequences; it needs device.shard=sequences '
                         f'(resolved: {shard}) and at least as many sequences as ranks')
    logger.info(f'ranks: {world}; sharding: {shard}')
    if by_sequence:
        result_path.mkdir(parents=True, exist_ok=True)      # every rank writes the pickles of its own sequences
    starts = []                                             # (sequence number, offsets into detection_results / indices) of this rank's sequences
    t_loop = time.perf_counter()
    LAST_RUN['loop_started_at'] = time.time()
    gen_seconds = 0.0
    if os.environ.get('VILGOD_SWITCH_INTERVAL'):
        sys.setswitchinterval(float(os.environ['VILGOD_SWITCH_INTERVAL']))
    # Sequences overlap (device.overlap_sequences, default on; one rank per sequence): the host-only tail of a sequence's stage list --
    # track boxes, label propagation, result dicts, the pickles -- runs on a background thread while the NEXT sequence's GPU stages
    # run on this one (ZeroShotDetector.split_stages).  The order of everything that is written or returned is the sequences' order.
    pending = []                                            # at most one: (thread, box with the error, finish())

    def finish_pending():
        while pending:
            th, err, fin, _ = pending.pop(0)
            th.join()
            if err:
                raise err[0]
            fin()

    for seq_no, sequence_name in enumerate(dataset.next_sequence()):
        if by_sequence and seq_no % world != rank:
            continue
        if result_data is not None:
            starts.append((seq_no, len(detection_results), len(indices)))
            indices.extend(dataset.sequence_indices)        # (upstream leaves this empty and falls back to dataset.index_mapping)
            continue
        result_file = result_path / f'{sequence_name}.pkl'
        indices_file = result_path / f'{sequence_name}_indices.pkl'
        if cfg.use_cached_results and 'evaluate_sequence' in cfg.pipeline_active and result_file.exists():
            finish_pending()
            starts.append((seq_no, len(detection_results), len(indices)))
            with result_file.open('rb') as f:
                detection_results.extend(pickle.load(f))
            with indices_file.open('rb') as f:
                indices.extend(pickle.load(f))
            continue
        if hasattr(dataset, 'prefetch_sequence'):
            t_gen = time.perf_counter()
            dataset.prefetch_sequence()                 # synthetic data: generate the sequence before the clock starts (stands for disk IO)
            t_gen_end = time.perf_counter()
            # only generator time during which no tail thread was working comes off the clock: the previous sequence's tail (boxes, label
            # propagation, pickles) may be running under it, and that work belongs to the loop
            hidden = 0.0
            for th, _, _, clock in pending:
                hidden = max(hidden, min(t_gen_end, clock.get('ended_at', t_gen_end) if not th.is_alive() else t_gen_end) - t_gen)
            gen_seconds += (t_gen_end - t_gen) - max(hidden, 0.0)
            LAST_RUN['tail_seconds_under_generator'] = LAST_RUN.get('tail_seconds_under_generator', 0.0) + max(hidden, 0.0)
        t_seq = time.perf_counter()
        zsd = ZeroShotDetector(dataset, sequence_name, cfg=cfg, logger=logger, pipeline=pipeline)
        seq_indices = list(dataset.sequence_indices)
        seq_len = dataset.sequence_length
        frame_ids = [info.get('frame_id', f'{sequence_name}_{i:03d}') for i, info in enumerate(dataset.sequence_infos)]
        front, back = zsd.split_stages()
        zsd.process(part='front' if back else 'all')
        torch.cuda.synchronize()
        t_front = time.perf_counter() - t_seq
        finish_pending()                                    # the previous sequence's tail ran under this sequence's GPU stages

        def finish(zsd=zsd, seq_no=seq_no, sequence_name=sequence_name, seq_indices=seq_indices, seq_len=seq_len, frame_ids=frame_ids,
                   result_file=result_file, indices_file=indices_file, t_front=t_front, clock=None):
            starts.append((seq_no, len(detection_results), len(indices)))
            detection_results.extend(zsd.detection_3d_result_list)
            indices.extend(seq_indices)
            seconds = t_front + (clock['back'] if clock else 0.0)
            LAST_RUN['sequences'].append({'name': sequence_name, 'frames': seq_len, 'world_size': world,
                                          'seconds': seconds, 'front_seconds': t_front, 'back_seconds': clock['back'] if clock else 0.0,
                                          'stage_ms_per_frame': dict(zsd.stage_ms), 'tail_on_thread': bool(clock and clock.get('thread', False)),
                                          # the stages every rank repeats over ALL frames when the frames of one sequence are sharded (ms per frame of the
                                          # sequence; with device.shard=sequences nothing is replicated)
                                          'replicated_ms_per_frame': 0.0 if (world == 1 or by_sequence) else round(sum(
                                              zsd.stage_ms.get(k, 0.0) * max(len(zsd.my_frames), 1) for k in ('track_clusters', 'propagate_labels', 'write_sequence_state')
                                          ) / max(seq_len, 1), 3),
                                          'detail_ms': dict(zsd.detail_ms)})
            if LAST_RUN['sequences'][-1]['replicated_ms_per_frame']:
                logger.info(f"  sequence-level stages repeated on every rank: {LAST_RUN['sequences'][-1]['replicated_ms_per_frame']:.2f} ms per frame of the sequence "
                            f"(of {1000.0 * seconds / max(seq_len, 1):.2f}); device.shard=sequences has none")

        def tail(zsd=zsd, sequence_name=sequence_name, seq_indices=seq_indices, frame_ids=frame_ids, result_file=result_file,
                 indices_file=indices_file, with_back=bool(back)):
            if with_back:
                zsd.process(part='back')
            if 'evaluate_sequence' in cfg.pipeline_active and (rank == 0 or by_sequence):
                with open(result_file, 'wb') as f:
                    pickle.dump(zsd.detection_3d_result_list, f)
                with open(indices_file, 'wb') as f:
                    pickle.dump(seq_indices, f)
                if cfg.get('export_pseudo_labels', False):
                    # OpenPCDet `infos`-style pickle + NPZ under paths.pseudo_label (an addition: upstream declares the path, never writes it)
                    from vilgod_amd import export
                    export.write_sequence(cfg.paths.pseudo_label, sequence_name, zsd.detection_3d_result_list, frame_ids,
                                          seq_indices, class_names=dataset.class_names)

        if back and zsd.back_is_host_only():
            import threading
            err, clock = [], {'back': 0.0, 'thread': True}

            def run(tail=tail, err=err, clock=clock, device=zsd.pipe.device):
                t0 = time.perf_counter()
                try:
                    torch.cuda.set_device(device)   # a new thread starts on device 0: a safety net should the tail ever touch the GPU
                    tail()
                except BaseException as e:      # noqa: BLE001  (re-raised on the main thread by finish_pending)
                    err.append(e)
                clock['ended_at'] = time.perf_counter()
                clock['back'] = clock['ended_at'] - t0
            th = threading.Thread(target=run, name=f'vilgod-tail-{sequence_name}', daemon=True)
            th.start()
            pending.append((th, err, lambda finish=finish, clock=clock: finish(clock=clock), clock))
        else:
            t0 = time.perf_counter()
            tail()
            torch.cuda.synchronize()
            finish(clock={'back': time.perf_counter() - t0})
        del zsd
        gc.collect()
        if dev.get('empty_cache_between_sequences', False):
            torch.cuda.empty_cache()                        # (off by default: the next sequence re-uses the allocator's blocks instead of
                                                            # paying ~0.1 s of hipMalloc for them again; 288 GB of HBM hold both)
    finish_pending()

    # the last sequence's state pickle may still be on its way to disk (background writer, vilgod_amd/zero_shot_detector.py): the
    # sequence loop -- and its clock -- ends when it has landed
    from vilgod_amd import zero_shot_detector as _zsd
    t_wait = time.perf_counter()
    _zsd.wait_state_writes()
    LAST_RUN['state_write_wait_seconds'] = time.perf_counter() - t_wait
    LAST_RUN['loop_seconds'] = time.perf_counter() - t_loop - gen_seconds      # (without the synthetic generator, which stands for disk IO)
    LAST_RUN['loop_ended_at'] = time.time()
    LAST_RUN['generator_seconds'] = gen_seconds
    if result_data is not None:
        detection_results = result_data
    elif by_sequence:
        # the per-frame result dicts of every rank's sequences, back in sequence order, for the one evaluation on rank 0
        ends = starts[1:] + [(None, len(detection_results), len(indices))]
        local_sequences = [(s[0], detection_results[s[1]:e[1]], indices[s[2]:e[2]]) for s, e in zip(starts, ends)]
        merged = sorted((item for part in vdist.gather_objects(local_sequences) for item in part), key=lambda it: it[0])
        detection_results = [d for _, det, _ in merged for d in det]
        indices = [i for _, _, idx in merged for i in idx]
    if len(detection_results) > 0 and rank == 0:
        # tools/preprocess_data.py:112-131 of the reference: one evaluation over all sequences with the evaluate_sequence arguments
        det3d_args = [pp for pp in cfg.pipeline if pp['name'] == 'evaluate_sequence'][0]['args']
        det3d_cfg = det3d_args['detection_3d']
        logger.info('_' * 100)
        logger.info('Evaluate all Sequences - Detection 3D')
        logger.info('_' * 100)
        ap_dict = dataset.evaluation(detection_results, class_names=dataset.class_names, indices=indices, eval_cfg=cfg.eval_cfg,
                                     class_agnostic=det3d_cfg['class_agnostic'], eval_range=det3d_args['eval_range'], bev=det3d_cfg['bev'],
                                     moving=det3d_args['moving'], static=det3d_args['static'], score_thresh=det3d_cfg['score_thresh'],
                                     sampling_rate=det3d_cfg['sampling_rate'])
        if isinstance(ap_dict, dict):
            from vilgod_amd.evaluation import print_eval_log
            print_eval_log(ap_dict, logger)
            logger.info(f"Summary over all sequences: { {k: v for k, v in ap_dict.items() if '/' not in k} }")
        else:
            logger.info(str(ap_dict))
        logger.info('_' * 100)
    return detection_results


if __name__ == '__main__':
    main()
