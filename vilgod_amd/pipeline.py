"""The per-frame pseudo-label hot path on one GPU (SURVEY §3.2 [A]-[F]), every per-point / per-crop
computation in the HIP kernels of csrc/, data resident in HBM from raw points to class scores.

Stage functions are separate so that the reference-shaped stage dispatcher (zero_shot_detector.py) and the
benchmark can call them one by one; `process_frame` chains them.

    ground(points)                 A1-A5   csrc/ground.hip
    to_ref + non-ground gather     B1      csrc/segment.hip (+ torch index plumbing)
    cluster(X)                     B2-B3   csrc/cluster.hip (GPU) + csrc/hdbscan_tree.cpp (host)
    ground plane + filters         C1-C2   csrc/segment.hip
    render + encode + score        D1-D9   csrc/render.hip, csrc/vit.hip
    vote                           D10     host numpy (<= ~150 clusters)
    boxes                          E1      csrc/segment.hip
    results                        F1      host numpy
"""
import contextlib
import os
import threading
import time

import numpy as np
import torch
from scipy.spatial.transform import Rotation

from ._lib import lib, ptr, stream_ptr, check
from .frame_state import FrameState, pack_clusters, vote, static_from_entropy
from . import patchworkpp as gpw
from .hdbscan import HDBSCAN
from .projection import RealisticProjection, VIEWS_4, VIEWS_6
from .clip_wrapper import ClipWrapper

# tools/configs/preprocessor/waymo.yaml:104-140
DEFAULT_CLASS_LIST = ['car', 'truck', 'bus', 'van', 'minivan', 'pickup truck', 'school bus', 'fire truck', 'ambulance',
                      'pedestrian', 'human body', 'human', 'cyclist', 'rider', 'bicycle', 'bike',
                      'traffic light', 'traffic sign', 'fence', 'pole', 'clutter', 'tree', 'house', 'wall']
DEFAULT_CLASS_MAPPING = {**{k: 'Vehicle' for k in DEFAULT_CLASS_LIST[:9]}, **{k: 'Pedestrian' for k in DEFAULT_CLASS_LIST[9:12]},
                         **{k: 'Cyclist' for k in DEFAULT_CLASS_LIST[12:16]}, **{k: 'Background' for k in DEFAULT_CLASS_LIST[16:]}}


def default_preprocessor_cfg():
    """The subset of tools/configs/preprocessor/waymo.yaml the hot path reads."""
    return dict(
        name='waymo', class_names=['Vehicle', 'Pedestrian', 'Cyclist'],
        clustering=dict(
            model=dict(cluster_selection_epsilon=0.15, min_cluster_size=15, metric='euclidean', core_dist_n_jobs=-1),
            filters_active=['filter_by_number_points', 'filter_by_plane_distance', 'filter_by_height'],
            filters=[dict(name='filter_by_number_points', args=dict(logic='and', required=True, min_points=10)),
                     dict(name='filter_by_height', args=dict(logic='and', required=True, min_height=0.3, max_height=6)),
                     dict(name='filter_by_plane_distance', args=dict(logic='and', required=True, max_min_height=1.0, min_max_height=0.5))],
            propability_threshold=0.3),
        lidar_image_projection=dict(depth_bias=0.2, obj_ratio=0.8, bg_clr=0.0, resolution=112, depth=8,
                                    gaussian_kernel=dict(sigma=3, zsigma=1)),
        clip=dict(name='clip', model_name='ViT-B-16.pt', top_k=1, split_size=50,
                  prompt_template='a point representation of a {}', class_list=DEFAULT_CLASS_LIST,
                  class_mapping=DEFAULT_CLASS_MAPPING))


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default) if hasattr(cfg, key) else default


class PseudoLabelPipeline:
    def __init__(self, preprocessor_cfg=None, device='cuda:0', vit_dtype='f16', n_views=4, max_points=300_000,
                 clip_model_path='../models/clip/', min_range=1.5, z_offset=1.723, plane_seed=666, clip=None,
                 box_mode='reference', box_workers=4, vit_graph=False, angle_mode='reference', cu_reserve=None, cu_tower=None,
                 hierarchy=None):
        cfg = preprocessor_cfg if preprocessor_cfg is not None else default_preprocessor_cfg()
        self.cfg = cfg
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.z_offset = float(z_offset)
        self.max_points = int(max_points)
        params = gpw.Parameters()
        params.min_range = float(min_range)
        self._pw_params = params
        self.ground_model = gpw.patchworkpp(params, max_points=self.max_points, device=self.device)
        ccfg = _get(cfg, 'clustering')
        mcfg = dict(_get(ccfg, 'model'))
        mcfg.pop('_target_', None)
        # hierarchy stage: 'device' (csrc/hdbscan_device.hip: the tree never leaves the GPU; default) or 'host' (csrc/hdbscan_tree.cpp in
        # the frame's thread) -- the same labels and probabilities bit for bit (tests/test_hierarchy.py); the device stage takes 0.9 ms of
        # a 150k-point frame's latency instead of 2.2 and costs the stream of frames nothing (LAB_NOTES.md section 0).  The device stage
        # holds min_cluster_size <= 32 and <= 2^20 points; beyond that the host stage runs (HDBSCAN decides).
        hierarchy = str(os.environ.get('VILGOD_HIERARCHY', 'device') if hierarchy is None else hierarchy)
        if hierarchy not in ('host', 'device'):
            raise ValueError("hierarchy: 'host' or 'device'")
        mcfg['hierarchy'] = hierarchy
        self.cluster_model = HDBSCAN(max_points=self.max_points, device=self.device, **mcfg)
        self.hierarchy = self.cluster_model.hierarchy
        self.prob_threshold = float(_get(ccfg, 'propability_threshold', 0.3))
        self._filters = self._parse_filters(ccfg)
        self.angle_mode = angle_mode             # view direction angle: 'device' | 'reference' (this host's numpy; projection.py)
        self.projection = RealisticProjection(_get(cfg, 'lidar_image_projection'), device=self.device,
                                              views=VIEWS_4 if n_views == 4 else VIEWS_6, angle_mode=angle_mode)
        clip_cfg = _get(cfg, 'clip')
        self.clip = clip if clip is not None else ClipWrapper(clip_cfg, clip_model_path, device=self.device, dtype=vit_dtype)
        self.vit_dtype = vit_dtype
        self.class_list = list(_get(clip_cfg, 'class_list'))
        mapping = _get(clip_cfg, 'class_mapping')
        self.mapped_names = sorted(set(mapping[c] for c in self.class_list))          # alphabetical = np.unique order
        self.fine_to_mapped = np.array([self.mapped_names.index(mapping[c]) for c in self.class_list])
        self.class_names = list(_get(cfg, 'class_names', ['Vehicle', 'Pedestrian', 'Cyclist']))
        self.cls_key = f"{_get(clip_cfg, 'name', 'clip')}_" + '_'.join(str(_get(clip_cfg, 'prompt_template')).format('').split(' ')[:-1])
        self.plane_seed = int(plane_seed)
        if box_mode not in ('reference', 'fast'):
            raise ValueError("box_mode: 'reference' (the reference's boxes: qhull vertex order, closing edge dropped) or 'fast' "
                             "(GPU hull + rectangle over all edges)")
        self.box_mode = box_mode
        # hipGraph-captured classification (BASELINE config 5): one graph per crop-count bucket, per worker, LRU-bounded
        # (clip_wrapper.GraphClassifier).  Off by default: with several frames in flight the launch overhead is already hidden
        # (bench.py `hipgraph_loop`: captured == plain launches within noise) and a real stream keeps meeting new crop counts.
        self.vit_graph = bool(vit_graph)
        self.patch_1ch = os.environ.get('VILGOD_PATCH_1CH', '1') != '0'      # renderer -> tower hand-over as single-channel patch rows
        self._graph_cls = None
        self._ground_stream = None
        # frames in flight: the ViT passes of the workers take turns in arrival order (see classify); shared by the worker clones
        import threading
        self._vit_turn = {'lock': threading.Lock(), 'events': [], 'depth': int(os.environ.get('VILGOD_VIT_CONCURRENCY', '3'))}
        self.box_workers = int(box_workers)      # helper processes for the host part of the reference box mode (0 = in the frame's thread)
        if self.box_mode == 'reference' and self.box_workers > 0:
            from . import boxes as _boxes
            _boxes._pool(self.box_workers)       # the helper processes import numpy / scipy (~1 s) while the tower's weights are set up
        self._xy_pinned = None
        self._ransac_work = torch.zeros(100 * 36 + 64, dtype=torch.uint8, device=self.device)
        self.timings = {}
        # CU-masked streams for the frames in flight (vilgod_amd/streams.py): the worker streams that carry a frame's front stage
        # (clustering, filters, render) may use `cu_reserve` CUs of every XCD, its ViT pass runs on a second stream restricted to the
        # other CUs ('complement') or unrestricted ('all').  0 = one unrestricted stream per worker (the default; sweep in LAB_NOTES.md).
        self.cu_reserve = int(os.environ.get('VILGOD_CU_RESERVE', '0') if cu_reserve is None else cu_reserve)
        self.cu_tower = str(os.environ.get('VILGOD_CU_TOWER', 'complement') if cu_tower is None else cu_tower)
        if self.cu_tower not in ('complement', 'all'):
            raise ValueError("cu_tower: 'complement' or 'all'")
        self._stream_factories = None
        self.vit_stream = None
        self._clip_cfg, self._clip_model_path, self._mcfg, self._n_views = clip_cfg, clip_model_path, mcfg, n_views
        self._workers = None

    # ---- several frames in flight ------------------------------------------------------------------------------
    def _clone_for_worker(self):
        """A shallow copy that shares configuration and constants but owns its stream-bound handles (cluster buffers,
        ViT workspace + weights, RANSAC scratch): handles are not thread-safe, like the reference objects."""
        import copy
        w = copy.copy(self)
        w.cluster_model = HDBSCAN(max_points=self.max_points, device=self.device, **self._mcfg)
        w.projection = RealisticProjection(_get(self.cfg, 'lidar_image_projection'), device=self.device,
                                           views=VIEWS_4 if self._n_views == 4 else VIEWS_6, angle_mode=self.angle_mode)
        w.clip = self.clip.view()                # shared read-only weights, own workspace
        w._ransac_work = torch.zeros(100 * 36 + 64, dtype=torch.uint8, device=self.device)
        w._xy_pinned = None
        w._graph_cls = None
        w._ground_stream = None
        w.timings = {}
        if self.cu_reserve > 0:
            if self._stream_factories is None:
                from .streams import make_streams
                self._stream_factories = make_streams(self.device, self.cu_reserve, self.cu_tower)
            k = len(self._workers or [])                 # this worker's position: the pooled masked streams are handed out by it
            w.stream, w.vit_stream = self._stream_factories[0](1 + k), self._stream_factories[1](k)
        else:
            w.stream, w.vit_stream = torch.cuda.Stream(device=self.device), None
        from concurrent.futures import ThreadPoolExecutor
        w.thread = ThreadPoolExecutor(max_workers=1)       # a worker's frames run one after the other on ITS thread
        return w

    def _ensure_workers(self, n_workers):
        if self._workers is None or len(self._workers) < n_workers:
            for _ in range(n_workers - len(self._workers or [])):
                self._workers = (self._workers or []) + [self._clone_for_worker()]
        return self._workers[:n_workers]

    def map_workers(self, items, fn, n_workers):
        """fn(worker, item) for every item, item i on worker i % n_workers (its thread, its stream, its handles) after everything
        queued so far on the caller's stream; returns the results in order once all workers have drained their streams."""
        items = list(items)
        n_workers = min(n_workers, len(items))
        if n_workers <= 1:
            return [fn(self, it) for it in items]
        workers = self._ensure_workers(n_workers)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))

        def run(worker, mine):
            with torch.cuda.stream(worker.stream):
                worker.stream.wait_event(ev)
                out = [fn(worker, it) for it in mine]
                worker.stream.synchronize()
            return out
        futs = [workers[k].thread.submit(run, workers[k], items[k::n_workers]) for k in range(n_workers)]
        parts = [f.result() for f in futs]
        return [parts[i % n_workers][i // n_workers] for i in range(len(items))]

    def process_frames(self, frames, poses, ref_pose, n_workers=3, first_fnr=0, after_ground=None, own=None, relay=None):
        """Throughput mode: frames (list of CUDA/numpy point arrays) are processed with `n_workers` frames in flight,
        each on its own HIP stream with its own handles.  Ground segmentation is stateful across frames and runs in
        frame order on the caller's stream; everything else of a frame runs on a worker stream after an event wait.
        `after_ground()` is called once all ground passes are queued, before the results are awaited.
        own: indices into `frames` this caller processes in full (default: all).  The other frames only go through upload + ground
        segmentation, in their place in the order: with several ranks every rank runs the cheap, stateful ground pass over the WHOLE
        sequence itself (0.36 ms per 150k-point scan on a high-priority stream, under its own frames' ViT work) and the frames are
        dealt round-robin, so rank r's first frame waits for r ground passes instead of r whole blocks and no state travels
        (bench.py --ground-handoff replicate).
        relay: (recv(i), send(i)) with `own`: the other ranks' frames are NOT touched at all; recv(i) is called before the ground pass of own
        frame i (it installs the ground state behind frame i - 1, taken from that frame's owner) and send(i) behind it (it hands the state
        on to the owner of frame i + 1) -- vilgod_amd/dist.py relay_recv_state / relay_send_state.
        Returns [(FrameState, result dict, probs tensor)] of the own frames, in frame order."""
        own_set = None if own is None else set(int(i) for i in own)
        own_idx = list(range(len(frames))) if own_set is None else sorted(own_set)
        slot = {i: k for k, i in enumerate(own_idx)}             # frame index -> position among the own frames
        workers = self._ensure_workers(n_workers)
        # the ground passes chain the frames (and, with several ranks, the ranks: the next rank waits for the state after this block):
        # they run on a HIGH-PRIORITY stream of their own so that their small kernels are dispatched ahead of the workers' heavy ones
        # instead of queueing behind every GEMM tile
        if getattr(self, '_ground_stream', None) is None:
            if self.cu_reserve > 0 and os.environ.get('VILGOD_CU_GROUND', 'masked') == 'masked':
                self._ground_stream = self._stream_factories[0](0)       # (a CU-masked stream cannot also carry a priority)
            else:
                self._ground_stream = torch.cuda.Stream(device=self.device, priority=-1)
        main = self._ground_stream
        main.wait_stream(torch.cuda.current_stream(self.device))

        # filling the pipeline: the first frame of every worker starts its clustering only when the frame before it has queued its
        # crops.  Started together, the n_workers clustering stages slow each other down and the first GEMM -- the work everything
        # else hides behind -- begins after ~25 ms instead of ~10; a ViT pass (~15 ms) is longer than a lone front stage, so the chain
        # never starves the encoder.  Frames after the first round are not gated.
        ramp_depth = int(os.environ.get('VILGOD_FILL_RAMP', '1'))          # front stages of a block's first round that may run at a time (0: all)
        ramp = ramp_depth > 0 and n_workers > 1
        front_done = [threading.Event() for _ in range(min(n_workers, len(own_idx)))]

        frame_log = [] if os.environ.get('VILGOD_FRAME_LOG') else None      # development aid: per-frame host timeline of the block
        t_block = time.perf_counter()

        def run(worker, i, d_pts, mask, ev):
            k_ = slot[i]
            gate = front_done[k_] if ramp and k_ < len(front_done) else None
            t_start = time.perf_counter()
            t_front = [None]

            def front():
                t_front[0] = time.perf_counter()
                if gate is not None:
                    gate.set()
            try:
                if gate is not None and k_ >= ramp_depth:
                    front_done[k_ - ramp_depth].wait()
                t_go = time.perf_counter()
                with torch.cuda.stream(worker.stream):
                    worker.stream.wait_event(ev)
                    fs, res = worker.process_frame(d_pts, poses[i], ref_pose, fnr=first_fnr + i, mask=mask,
                                                   before_classify=front if (gate is not None or frame_log is not None) else None)
                    probs = getattr(worker, 'last_probs', None)
                    worker.stream.synchronize()
            finally:
                if gate is not None:
                    gate.set()                   # frames without valid clusters, errors: never leave the next worker waiting
            if frame_log is not None:
                frame_log.append((i, workers.index(worker), t_start - t_block, t_go - t_block, (t_front[0] or t_go) - t_block, time.perf_counter() - t_block))
            return fs, res, probs

        # Frames are handed out in order to whichever worker is free (one shared queue), not dealt round-robin: the workers do not
        # run in step -- the ones the ramp starts last meet a GPU already full of GEMM tiles and need 30-40 ms for a front stage that
        # takes 12 alone -- and with a fixed deal the block ended when the slowest worker had worked off its share while the others
        # sat idle (20-frame block: worker 5 began its second frame at 200 ms and its third at 286 of 340; same box, interleaved:
        # 52.1 -> 60.3 frames/s in 20-frame blocks, 96-frame blocks unchanged).
        from concurrent.futures import Future
        import queue
        futures = {i: Future() for i in own_idx}
        jobs = queue.Queue()

        # (the block's first n_workers frames still go one to each worker: a short warm-up block then exercises every worker's
        # stream, allocator pool and ViT workspace)
        n_active = min(n_workers, len(own_idx))
        first = [queue.Queue() for _ in range(n_active)]

        def drain(worker, k):
            job = first[k].get()
            while job is not None:
                i, d_pts, mask, ev = job
                if futures[i].set_running_or_notify_cancel():
                    try:
                        futures[i].set_result(run(worker, i, d_pts, mask, ev))
                    except BaseException as e:     # noqa: BLE001  (delivered through the frame's future)
                        futures[i].set_exception(e)
                job = jobs.get()

        drains = [workers[k].thread.submit(drain, workers[k], k) for k in range(n_active)]
        fed = 0
        try:
            with torch.cuda.stream(main):
                for i, pts in enumerate(frames):
                    mine = own_set is None or i in own_set
                    if relay is not None and not mine:
                        continue                 # another rank's frame: its owner runs the ground pass and relays the state
                    if relay is not None:
                        relay[0](i)
                    d_pts = self.upload(pts)
                    mask = self.ground(d_pts)
                    if relay is not None:
                        relay[1](i)
                    if not mine:
                        continue                 # another rank's frame: only the ground state moves on
                    ev = torch.cuda.Event()
                    ev.record(main)
                    (first[fed] if fed < n_active else jobs).put((i, d_pts, mask, ev))
                    fed += 1
                if after_ground is not None:
                    after_ground()         # every ground pass of the block is queued: e.g. hand the ground state to the next rank
        finally:
            for k in range(fed, n_active):
                first[k].put(None)             # (only when an upload / ground pass failed before the first round was out)
            for _ in drains:
                jobs.put(None)                 # one stop mark per draining worker, behind the last frame
            if fed < len(own_idx):
                for d in drains:               # ... let the frames already handed out finish before the error goes up
                    d.result()
        out, first_error = [], None
        try:
            for f in (futures[i] for i in own_idx):     # drain every worker even when one frame failed: nothing keeps running behind the caller's back
                try:
                    out.append(f.result())
                except BaseException as e:     # noqa: BLE001
                    first_error = first_error or e
        finally:
            for d in drains:
                d.result()                     # the workers' threads are free again before the caller goes on
            torch.cuda.current_stream(self.device).wait_stream(main)
        if first_error is not None:
            raise first_error
        if frame_log is not None:
            import sys
            print('[frame log] frame worker: submitted  started  crops queued  done (ms since the block began)', file=sys.stderr)
            for i, w, a, b, c, d in sorted(frame_log):
                print(f'[frame log] {i:4d} {w:2d}: {1e3 * a:8.1f} {1e3 * b:8.1f} {1e3 * c:8.1f} {1e3 * d:8.1f}', file=sys.stderr)
        return out

    @staticmethod
    def _parse_filters(ccfg):
        active = list(_get(ccfg, 'filters_active', []))
        f = dict(min_points=0, max_points=999999, max_min_height=np.inf, min_max_height=-np.inf, min_height=-np.inf,
                 max_height=np.inf, use_plane=False)
        known = {'filter_by_number_points', 'filter_by_plane_distance', 'filter_by_height'}
        for flt in _get(ccfg, 'filters', []):
            name, args = _get(flt, 'name'), dict(_get(flt, 'args', {}))
            if name not in active:
                continue
            if name not in known:
                # zero_shot_detector.py:283: filters that do not exist in cluster_utils are skipped silently
                # (filter_by_density); the others of cluster_utils are not used by the shipped configs
                if name in ('filter_by_aspect_ratio', 'filter_by_volume', 'filter_by_area', 'filter_by_ephemeral_score'):
                    raise NotImplementedError(f'{name} is not active in the reference configs and has no GPU kernel')
                continue
            if not (args.get('logic') == 'and' and args.get('required', False)):
                raise NotImplementedError('only `logic: and, required: True` filters (the shipped configuration)')
            if name == 'filter_by_number_points':
                f['min_points'], f['max_points'] = args.get('min_points', 0), args.get('max_points', 999999)
            elif name == 'filter_by_height':
                f['min_height'], f['max_height'] = args['min_height'], args['max_height']
            else:
                f['max_min_height'], f['min_max_height'], f['use_plane'] = args['max_min_height'], args['min_max_height'], True
        return f

    # ---------------------------------------------------------------------------------------------
    def _mark(self, name):
        """Latency accounting of ONE frame (process_frame(timing=True) only): device-synchronise and book the time since the previous
        mark under `name` (bench.py `frame_latency_ms`).  A no-op in the throughput paths."""
        lat = getattr(self, '_lat', None)
        if lat is None:
            return
        torch.cuda.synchronize(self.device)
        now = time.perf_counter()
        lat[name] = lat.get(name, 0.0) + (now - lat['_t'])
        lat['_t'] = now

    def new_sequence(self):
        """A fresh Patchwork++ state per sequence (zero_shot_detector.py:137-140)."""
        self.ground_model.reset()

    def upload(self, points):
        if isinstance(points, torch.Tensor):
            return points.to(self.device, non_blocking=True)
        return torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(self.device, non_blocking=True)

    # [A]
    def ground(self, d_points):
        return self.ground_model.estimate_mask(d_points, self.z_offset)

    # [B1]
    def to_ref(self, d_points, transform_to_ref):
        T = torch.from_numpy(np.ascontiguousarray(transform_to_ref, dtype=np.float64)).to(self.device)
        out = torch.empty_like(d_points)
        if d_points.shape[0] == 0:
            return out
        check(lib.vg_ref_transform(ptr(d_points), d_points.shape[0], d_points.stride(0), ptr(T), ptr(out), stream_ptr()),
              'vg_ref_transform')
        return out

    # [B2] + [B3]
    def cluster(self, d_X):
        """d_X: [M,>=3] CUDA float32 (points_ref_wo_ground).  -> labels, probs (host)."""
        n = d_X.shape[0]
        if n < 2:
            return np.full(n, -1, np.int64), np.zeros(n)
        lo, hi, w2 = self.cluster_model.mst(d_X)
        self._mark('mst_kernels')
        if self.cluster_model.hierarchy == 'device':
            d_labels, d_probs, _ = self.cluster_model.tree_device(lo, hi, w2, n)
            self._mark('hierarchy_device')
            labels, probs = d_labels.cpu().numpy(), d_probs.cpu().numpy()
            self._mark('labels_d2h')
            return labels, probs
        h_lo, h_hi, h_w2 = lo.cpu().numpy(), hi.cpu().numpy(), w2.cpu().numpy()
        self._mark('mst_d2h')
        labels, probs, _ = self.cluster_model.tree(h_lo, h_hi, h_w2, n)
        self._mark('hierarchy_host')
        return labels, probs

    # [C2]
    def ground_plane(self, d_points_ref, d_ground_idx):
        """fit_plane(points_ref[ground_mask]) (lidar_frame.py:96-109; pointcloud_utils.py:375-387): two RANSAC stages."""
        n = int(d_ground_idx.numel())
        if n < 3:
            return np.array([0.0, 0.0, 1.0, 0.0])
        dev = self.device
        plane = torch.empty(4, dtype=torch.float64, device=dev)
        flags = torch.empty(n, dtype=torch.uint8, device=dev)
        cnt = torch.empty(1, dtype=torch.int32, device=dev)
        idx = d_ground_idx.to(torch.int32)
        check(lib.vg_plane_ransac(ptr(d_points_ref), d_points_ref.stride(0), ptr(idx), n, 0.1, 100, self.plane_seed,
                                  ptr(self._ransac_work), ptr(plane), ptr(flags), ptr(cnt), stream_ptr()), 'vg_plane_ransac')
        idx2 = idx[flags.bool()].contiguous()
        n2 = int(idx2.numel())
        if n2 >= 3:
            flags2 = torch.empty(n2, dtype=torch.uint8, device=dev)
            check(lib.vg_plane_ransac(ptr(d_points_ref), d_points_ref.stride(0), ptr(idx2), n2, 0.1, 100, self.plane_seed + 1,
                                      ptr(self._ransac_work), ptr(plane), ptr(flags2), ptr(cnt), stream_ptr()), 'vg_plane_ransac')
        p = plane.cpu().numpy()
        if p[2] < 0:
            p = p * -1
        return p

    # [B4] + [C1]
    def filter(self, d_X, d_index, d_seg, plane):
        C = d_seg.numel() - 1
        stats = torch.empty((C, 6), dtype=torch.float32, device=self.device)
        valid = torch.empty(C, dtype=torch.uint8, device=self.device)
        f = self._filters
        d_plane = torch.from_numpy(np.ascontiguousarray(plane, dtype=np.float64)).to(self.device)
        big = 1e300
        check(lib.vg_cluster_filter(ptr(d_X), d_X.stride(0), ptr(d_index), ptr(d_seg), C, ptr(d_plane), int(f['min_points']),
                                    int(f['max_points']), float(min(f['max_min_height'], big)), float(max(f['min_max_height'], -big)),
                                    float(max(f['min_height'], -big)), float(min(f['max_height'], big)), ptr(stats), ptr(valid),
                                    stream_ptr()), 'vg_cluster_filter')
        return valid, stats

    # [D1]-[D9]
    def classify(self, d_X, d_index, d_seg, transform_to_ego):
        n = (d_seg.numel() - 1) * self.projection.num_views
        enc = self.clip.encoder
        if self.vit_dtype == 'f16' and enc.cfg['patch'] == 16 and enc.cfg['resolution'] == 224:
            # the renderer writes the patch-embedding GEMM's A operand directly (no CHW crops, no im2col pass)
            # (single-channel rows by default: the crop's three channels are one image, the tower folds their normalisation into a
            # K = 256 patch embedding -- a third of the bytes the renderer writes and the GEMM reads; VILGOD_PATCH_1CH=0: 768-wide rows)
            from .clip_wrapper import clip_scores, GraphClassifier
            pmode = 'patch16c1' if self.patch_1ch else 'patch16'
            if self.vit_graph and n > 0 and torch.cuda.current_stream(self.device).cuda_stream != 0:
                # captured loop: the ViT encode + scores replay as one hipGraph (per crop count) on this worker's persistent buffers;
                # the outputs are cloned because the buffers are rewritten by the worker's next frame
                if self._graph_cls is None:
                    self._graph_cls = GraphClassifier(enc, self.clip.text_features, max_crops=max(512, n), patch_width=256 if self.patch_1ch else 768)
                g = self._graph_cls
                self.projection.render_frame(d_X, d_index, d_seg, transform_to_ego, out=pmode, out_buf=g.patch_buffer(n))
                with self._vit_in_turn():
                    probs, top1, score = g.classify(n, self.clip.text_features)
                return probs.clone(), top1.clone(), score.clone()
            patches = self.projection.render_frame(d_X, d_index, d_seg, transform_to_ego, out=pmode)
            self._mark('render')
            vs = self.vit_stream
            if vs is None or n == 0:
                with self._vit_in_turn():
                    return clip_scores(enc.encode_patches(patches, n), self.clip.text_features)
            # CU-masked streams: the tower runs on this worker's ViT stream (its own CU set), ordered after the render and in front of
            # whatever this frame's stream does next
            cur = torch.cuda.current_stream(self.device)
            vs.wait_stream(cur)
            patches.record_stream(vs)
            with torch.cuda.stream(vs):
                with self._vit_in_turn():
                    out = clip_scores(enc.encode_patches(patches, n), self.clip.text_features)
            cur.wait_stream(vs)
            for t_ in out:
                t_.record_stream(cur)
            return out
        crops = self.projection.render_frame(d_X, d_index, d_seg, transform_to_ego, out='f16' if self.vit_dtype == 'f16' else 'f32')
        return self.clip.predict_probs(crops)

    @contextlib.contextmanager
    def _vit_in_turn(self):
        """Frames in flight: at most `depth` (3; env VILGOD_VIT_CONCURRENCY, 0 = unlimited) ViT passes of the worker streams run at
        a time, in arrival order (a pass waits for the event recorded after the pass `depth` before it).  Left alone, the GPU
        interleaves the GEMM tiles of all queued passes: every pass takes n_workers times as long, all finish together, and the
        workers then move in lock step (all clustering, then all encoding; kernel trace: tools/analyze_fill.py) instead of one
        frame's clustering and rendering running underneath another frame's GEMMs.  Strictly one pass at a time staggers the
        workers but leaves the tails of a pass (LayerNorm, attention, head: few workgroups) uncovered; two cover each other.
        Measured, 20-frame blocks: unlimited 18.55, one 18.5, two 17.9, four 18.3 ms per frame; 96-frame blocks: 16.8 for all but
        one (17.65) -- round 2, one GEMM tile per workgroup.  Round 6 (persistent GEMM workgroups: a launch holds its CUs until its tiles
        are done, queued launches no longer interleave tile by tile): one 63.3, two 72.2-72.6, THREE 72.6-74.1, four 72.9-73.6,
        unlimited 72.8-73.8 frames/s (20- and 48-frame blocks, two rounds, one process each on one box): three since."""
        turn = self._vit_turn
        st = torch.cuda.current_stream(self.device)
        depth = turn['depth']
        if depth <= 0 or st.cuda_stream == 0:
            yield
            return
        with turn['lock']:
            if len(turn['events']) >= depth:
                st.wait_event(turn['events'][-depth])
            yield
            ev = torch.cuda.Event()
            ev.record(st)
            turn['events'] = (turn['events'] + [ev])[-depth:]

    # [E1]
    def boxes(self, d_X, d_index, d_seg):
        """'fast' mode kernel: exact hull + rectangle over ALL hull edges, float64 (csrc/segment.hip k_cluster_box)."""
        C = d_seg.numel() - 1
        box = torch.empty((C, 7), dtype=torch.float64, device=self.device)
        aux = torch.empty((C, 3), dtype=torch.float32, device=self.device)
        check(lib.vg_cluster_boxes(ptr(d_X), d_X.stride(0), ptr(d_index), ptr(d_seg), C, ptr(box), ptr(aux), stream_ptr()),
              'vg_cluster_boxes')
        return box, aux

    def cluster_medians(self, d_X, d_index, d_seg):
        """Detection.cluster_mass_center (objects.py:121-123) of the packed clusters over ALL columns of d_X -> CUDA [C, cols] f32."""
        C = d_seg.numel() - 1
        out = torch.empty((C, d_X.shape[1]), dtype=torch.float32, device=self.device)
        check(lib.vg_cluster_medians(ptr(d_X), d_X.stride(0), d_X.shape[1], ptr(d_index), ptr(d_seg), C, ptr(out), stream_ptr()),
              'vg_cluster_medians')
        return out

    def xy_to_host_async(self, d_X):
        """Start the D2H copy of points_ref_wo_ground[:, :2] into this worker's pinned buffer (reference box mode reads each
        cluster's xy points once on the host, vilgod_amd/boxes.py).  -> (host array view, event to wait for)."""
        n = d_X.shape[0]
        if self._xy_pinned is None or self._xy_pinned.shape[0] < n:
            self._xy_pinned = torch.empty((max(n, self.max_points), 2), dtype=torch.float32, pin_memory=True)
        dst = self._xy_pinned[:n]
        dst.copy_(d_X[:, :2], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return dst.numpy(), ev

    def z_extent(self, d_X, d_index, d_seg):
        """float32 (zmin, zmax) per packed cluster from the statistics kernel (vg_cluster_filter, thresholds open)."""
        C = d_seg.numel() - 1
        stats = torch.empty((C, 6), dtype=torch.float32, device=self.device)
        valid = torch.empty(C, dtype=torch.uint8, device=self.device)
        d_plane = torch.tensor([0.0, 0.0, 1.0, 0.0], dtype=torch.float64, device=self.device)
        check(lib.vg_cluster_filter(ptr(d_X), d_X.stride(0), ptr(d_index), ptr(d_seg), C, ptr(d_plane), 0, 2 ** 31 - 1, 1e300, -1e300,
                                    -1e300, 1e300, ptr(stats), ptr(valid), stream_ptr()), 'vg_cluster_filter')
        st = stats.cpu().numpy()
        return st[:, 1], st[:, 2]

    def fit_boxes(self, d_X, index, seg, d_index=None, d_seg=None, xy_host=None, zmin=None, zmax=None):
        """fit_bounding_boxes_simple, static branch (zero_shot_detector.py:444-462), for the packed clusters (index, seg).
        -> [C,7] float64 numpy boxes in the reference frame, by `self.box_mode` (vilgod_amd/boxes.py)."""
        C = len(seg) - 1
        if C == 0:
            return np.zeros((0, 7))
        if d_index is None:
            d_index = torch.from_numpy(np.ascontiguousarray(index, dtype=np.int32)).to(self.device)
            d_seg = torch.from_numpy(np.ascontiguousarray(seg, dtype=np.int32)).to(self.device)
        if self.box_mode == 'fast':
            return self.boxes(d_X, d_index, d_seg)[0].cpu().numpy()
        return self.fit_boxes_async(d_X, index, seg, d_index, d_seg, xy_host, zmin, zmax).result()

    def fit_boxes_async(self, d_X, index, seg, d_index=None, d_seg=None, xy_host=None, zmin=None, zmax=None):
        """Reference mode: start the host part (vilgod_amd/boxes.py) in a helper process; .result() -> [C,7] boxes."""
        from .boxes import submit_reference_boxes
        if d_index is None:
            d_index = torch.from_numpy(np.ascontiguousarray(index, dtype=np.int32)).to(self.device)
            d_seg = torch.from_numpy(np.ascontiguousarray(seg, dtype=np.int32)).to(self.device)
        if zmin is None:
            zmin, zmax = self.z_extent(d_X, d_index, d_seg)
        if xy_host is None:
            xy_host, ev = self.xy_to_host_async(d_X)
            ev.synchronize()
        return submit_reference_boxes(xy_host, index, seg, zmin, zmax, self.box_workers)

    # [F1]
    @staticmethod
    def boxes_to_ego(boxes_ref, transform_to_ego):
        """apply_transform(boxes, transform_to_ego, box=True) (pointcloud_utils.py:21-46; zero_shot_detector.py:847)."""
        if len(boxes_ref) == 0:
            return np.zeros((0, 7))
        out = np.array(boxes_ref, dtype=np.float64, copy=True)
        h = np.hstack((out[:, :3], np.ones((len(out), 1))))
        out[:, :3] = np.einsum('ij,kj->ki', transform_to_ego, h)[:, :3]
        out[:, 6] += Rotation.from_matrix(transform_to_ego[:3, :3]).as_euler('xyz')[-1]
        return out

    # ---------------------------------------------------------------------------------------------
    def prepare(self, points, pose, ref_pose, fnr=0, state=None, mask=None):
        """[A] + [B1]: ground mask, reference-frame transform, non-ground gather.
        -> (FrameState, points_ref, points_ref_wo_ground, ground indices), all CUDA."""
        fs = state if state is not None else FrameState(fnr, pose, ref_pose)
        d_pts = self.upload(points)
        fs.n_points = d_pts.shape[0]
        if mask is None:
            mask = self.ground(d_pts)
        d_ref = self.to_ref(d_pts, fs.transform_to_ref)
        ng = torch.nonzero(mask == 0).squeeze(1)
        gidx = torch.nonzero(mask).squeeze(1)
        d_X = d_ref.index_select(0, ng).contiguous()
        fs.ground_point_indices = gidx.cpu().numpy()
        fs.n_nonground = d_X.shape[0]
        return fs, d_ref, d_X, gidx

    def process_frame(self, points, pose, ref_pose, fnr=0, state=None, timing=False, mask=None, before_classify=None):
        """One frame through [A]-[F].  points: (N,>=4) float32 numpy/CUDA [x,y,z,intensity,...].
        Returns (FrameState, result dict {'boxes_lidar','name','score','moving'})."""
        t = {}
        sync = torch.cuda.synchronize if timing else (lambda: None)

        def tick(name, t0):
            sync()
            t[name] = time.perf_counter() - t0
            return time.perf_counter()

        t0 = time.perf_counter()
        self._lat = {'_t': t0} if timing else None
        try:
            if mask is None:
                d_pts = self.upload(points)
                self._mark('upload')
                mask = self.ground(d_pts)
                self._mark('ground')
                t0 = tick('ground', t0)
                points = d_pts
            else:
                t['ground'] = 0.0
            fs, d_ref, d_X, gidx = self.prepare(points, pose, ref_pose, fnr=fnr, state=state, mask=mask)
            self._mark('to_ref+gather')
            t0 = tick('to_ref', t0)
            labels, probs = self.cluster(d_X)
            t0 = tick('cluster', t0)
            out = self.label(fs, d_ref, d_X, gidx, labels, probs, t=t, t0=t0, tick=tick, before_classify=before_classify)
            if timing:
                self.latency = {k: v for k, v in self._lat.items() if k != '_t'}
            return out
        finally:
            self._lat = None

    def process_sequence(self, frames, poses, ref_pose, entropy_args=None, n_frames=2, seed=0, first_fnr=0, n_workers=1):
        """The reference's DEFAULT stage order over a whole sequence (preprocessing.yaml:50-68):
        mask_ground_points -> calculate_entropy_scores (sliding window over the neighbouring frames) ->
        spatial_clustering with n_frames frames (5-D HDBSCAN + nearest-label transfer) -> filters / classification /
        boxes per frame as in `process_frame`.  `Detection.static` comes from the clusters' entropy percentile.
        Returns [(FrameState, result dict)] in frame order."""
        from .entropy import EntropyScorer, TwoFrameClusterer, full_scores
        self.new_sequence()
        prepared = []
        for i, pts in enumerate(frames):
            prepared.append(self.prepare(pts, poses[i], ref_pose, fnr=first_fnr + i))
        X_list = [p[2] for p in prepared]
        scorer = EntropyScorer(self.cluster_model, **(entropy_args or {}))
        mapper = (lambda items, fn: self.map_workers(items, lambda w, it: fn(w.cluster_model, it), n_workers)) if n_workers > 1 else None
        H_list = scorer.score_sequence(X_list, mapper=mapper)
        ent_list = []
        for (fs, _, d_X, _), H in zip(prepared, H_list):
            fs.entropy_scores, fs.entropy_indices = scorer.reduce(H)
            ent_list.append(full_scores(d_X.shape[0], fs.entropy_scores, fs.entropy_indices, device=self.device))
        use_two = n_frames > 1 and len(frames) >= n_frames
        ent_host = [e.cpu().numpy() for e in ent_list]
        if n_workers <= 1:
            two = TwoFrameClusterer(self.cluster_model, n_frames=n_frames, seed=seed) if use_two else None
            out = []
            for i, (fs, d_ref, d_X, gidx) in enumerate(prepared):
                labels, probs = two.labels(i, X_list, ent_list) if use_two else self.cluster(d_X)
                out.append(self.label(fs, d_ref, d_X, gidx, labels, probs, entropy=ent_host[i]))
            return out
        # several frames in flight: ground / entropy / the per-frame clustering rows are sequence-level work on the caller's
        # stream; clustering + label transfer + filters + classification + boxes of a frame run on a worker stream
        workers = self._ensure_workers(n_workers)
        parts = TwoFrameClusterer(self.cluster_model, n_frames=n_frames, seed=seed).precompute_parts(X_list, ent_list, mapper=mapper) if use_two else None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))

        def run(worker, i):
            fs, d_ref, d_X, gidx = prepared[i]
            with torch.cuda.stream(worker.stream):
                worker.stream.wait_event(ev)
                if use_two:
                    labels, probs = TwoFrameClusterer(worker.cluster_model, n_frames=n_frames, seed=seed, parts=parts).labels(i, X_list, ent_list)
                else:
                    labels, probs = worker.cluster(d_X)
                r = worker.label(fs, d_ref, d_X, gidx, labels, probs, entropy=ent_host[i])
                worker.stream.synchronize()
            return r

        futures = [workers[i % n_workers].thread.submit(run, workers[i % n_workers], i) for i in range(len(prepared))]
        return [f.result() for f in futures]

    def label(self, fs, d_ref, d_X, gidx, labels, probs, entropy=None, t=None, t0=None, tick=None, before_classify=None):
        """Everything after clustering: detections, static flags, filters, classification, boxes, results."""
        t = {} if t is None else t
        if tick is None:
            tick = lambda name, t0: time.perf_counter()
            t0 = time.perf_counter()
        # per-crop score matrix of THIS frame (empty unless the frame reaches classification): never a previous frame's
        self.last_probs = torch.zeros((0, len(self.class_list)), dtype=torch.float32, device=self.device)
        ids, index, seg = pack_clusters(labels, probs, self.prob_threshold)
        fs.set_clusters(ids, index, seg)
        C = len(ids)
        if entropy is not None and C:
            ecfg = _get(_get(self.cfg, 'clustering'), 'entropy_score_filter', None)
            fs.static = static_from_entropy(entropy, index, seg, percentile=float(_get(ecfg, 'percentile', 30) if ecfg else 30),
                                            min_percentile_pp_score=float(_get(ecfg, 'min_percentile_pp_score', 0.5) if ecfg else 0.5))
        result = {'boxes_lidar': np.zeros((0, 7)), 'name': np.array([]), 'score': np.array([]), 'moving': np.array([])}
        if C == 0:
            self.timings = t
            return fs, result
        d_index = torch.from_numpy(index).to(self.device)
        d_seg = torch.from_numpy(seg).to(self.device)
        self._mark('pack_clusters+h2d')
        xy_host = xy_ev = None
        if self.box_mode == 'reference':
            xy_host, xy_ev = self.xy_to_host_async(d_X)            # lands while the plane fit / filters / classification run
        plane = self.ground_plane(d_ref, gidx) if self._filters['use_plane'] else np.array([0.0, 0.0, 1.0, 0.0])
        fs.ground_plane_model_ref = plane
        valid, stats = self.filter(d_X, d_index, d_seg, plane)
        fs.valid = valid.cpu().numpy().astype(bool)
        st = stats.cpu().numpy() if self.box_mode == 'reference' else None     # (the same kernel wrote it: no further wait)
        fs.filtered = True
        self._mark('plane+filter')
        t0 = tick('filter', t0)
        vrows = np.flatnonzero(fs.valid)
        if len(vrows) == 0:
            self.timings = t
            return fs, result
        # packed sub-list of the valid clusters (classification and boxes are `valid_only`, preprocessing.yaml:81,89)
        parts = [index[seg[c]:seg[c + 1]] for c in vrows]
        v_index = np.concatenate(parts)
        v_seg = np.r_[0, np.cumsum([len(p) for p in parts])].astype(np.int32)
        d_vindex = torch.from_numpy(v_index).to(self.device)
        d_vseg = torch.from_numpy(v_seg).to(self.device)
        if before_classify is not None:
            before_classify()                    # the frame's clustering / filtering is done, its crops are about to be queued
        self._mark('valid_lists')
        probs_d, top1, score = self.classify(d_X, d_vindex, d_vseg, fs.transform_to_ego)
        box_fut = None
        if self.box_mode == 'reference':
            # host part of the reference-exact box fit (vilgod_amd/boxes.py) in a helper process.  The boxes do not depend on the classes;
            # the request (gathering and pickling the clusters' xy points: ~1 ms of this thread) goes out AFTER the frame's crops are
            # queued -- the GPU renders and encodes meanwhile, the helpers have the ViT pass's ~13 ms for their ~2.5 ms (round 5: the
            # request used to sit in front of the render, on the frame's critical path)
            xy_ev.synchronize()
            box_fut = self.fit_boxes_async(d_X, v_index, v_seg, d_vindex, d_vseg, xy_host=xy_host, zmin=st[vrows, 1], zmax=st[vrows, 2])
        self._mark('encode+scores')              # (incl. the box request sent while the GPU encodes)
        if box_fut is None:
            box = self.fit_boxes(d_X, v_index, v_seg, d_vindex, d_vseg)
        top1 = top1.cpu().numpy()
        score = score.cpu().numpy()
        if box_fut is not None:
            box = box_fut.result()
        self._mark('scores_d2h+box_wait')
        t0 = tick('classify+boxes', t0)
        V = self.projection.num_views
        nv = len(vrows)
        fine = top1.reshape(nv, V)
        sc = score.reshape(nv, V).astype(np.float32)
        mapped = self.fine_to_mapped[fine]
        win, final = vote(mapped, sc, self.mapped_names)
        names = np.array(self.mapped_names, dtype=object)
        fine_names = np.array(self.class_list, dtype=object)
        fs.set_classes(self.cls_key, fs.valid.copy(), names[mapped], fine_names[fine], sc, names[win], final)
        fs.boxes = np.full((C, 7), np.nan)
        fs.boxes[vrows] = box
        # [F1] evaluate_sequence (zero_shot_detector.py:832-857)
        keep = np.array([names[w] in self.class_names for w in win], dtype=bool)
        result = {'boxes_lidar': self.boxes_to_ego(box[keep], fs.transform_to_ego),
                  'name': np.array([str(names[w]) for w in win[keep]]),
                  'score': np.array(final[keep]),
                  'moving': np.zeros(int(keep.sum()), dtype=bool)}
        self._mark('vote+results')
        t0 = tick('vote+results', t0)
        self.timings = t
        self.last_probs = probs_d
        return fs, result
