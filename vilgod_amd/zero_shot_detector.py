"""Sequence-level stage dispatcher: the MI355X counterpart of the reference's
`src/vilgod/zero_shot_detector.py::ZeroShotDetector` for the hot-path stages.

Same contract (SURVEY §8b "Stage dispatch"): `process()` walks `cfg.pipeline_active`, looks each name up in
`cfg.pipeline` and calls `getattr(self, name)(**args)`; unknown names only warn (zero_shot_detector.py:62-68).
Stage names, keyword arguments, skip-if-already-done rules and the two pickle families are the reference's:
  mask_ground_points(min_range, z_offset)                          :129-151
  calculate_entropy_scores(n_neighbouring_frames, skip_frames, ...) :153-195
  spatial_clustering(force, n_frames)            (n_frames >= 1)    :197-259
  filter_detections(force)                                          :261-297
  classification(image_size, key, aggregation, valid_only, ...)     :329-420
  fit_bounding_boxes_simple(method, force, valid_only, ...)         :422-462 (static branch)
  evaluate_sequence(modes, classification_key, ...)                 :826-857
  sync_lidar_frames(mode)                                           :105-123
SURVEY §8f rows N1 (entropy scores, two-frame clustering) and N2 (`track_clusters`, the track branch of
`fit_bounding_boxes_simple`, `propagate_labels`; host logic in vilgod_amd/tracking.py) are built: the shipped default
`pipeline_active` runs end to end.

Execution differs on purpose: per-frame device data (points, ref-frame points, non-ground subset, cluster lists)
stays resident in HBM across stages (a 199-frame Waymo segment is ~0.6 GB), every stage calls the HIP kernels
through vilgod_amd.pipeline, and with torch.distributed initialised the frames of the sequence are sharded over
the ranks (contiguous blocks): every rank replays the cheap, stateful ground stage over all frames (exact
parity with the sequential reference), the other stages run on the rank's own block, and
`evaluate_sequence` all-gathers scores and per-frame results so that every rank holds the sequence result.
"""
import contextlib
import os
import pickle
import threading
import time
from pathlib import Path

import numpy as np
import torch

from . import dist as vdist
from .frame_state import FrameState, pack_clusters, vote
from .pipeline import PseudoLabelPipeline


# The sequence-state pickle (zero_shot_detector.py:105-114 of the reference) of a 199-frame Waymo-shape sequence is ~300 MB in ~18 000
# per-detection dicts of numpy objects: ~0.45 s of pickle.dump that holds the interpreter lock, plus ~0.08 s to build the dicts.  With
# device.async_state_write (default on) the frames go, as FrameState.compact() arrays, to ONE helper process
# (python -m vilgod_amd.state_writer: numpy only) that builds the dicts and writes the file, fed by one background thread, so that the
# next sequence's GPU pass -- whose worker threads need the interpreter lock to launch kernels -- is not held up.  At most one write is
# outstanding; tools/preprocess_data.py waits for the last one before it returns, and a detector that is about to LOAD a file waits
# for a write of that file first.  If the helper cannot be started or dies, the file is written in the background thread instead.
_STATE_WRITER = {'pool': None, 'pending': None, 'path': None, 'proc': None}
_STATE_LOCK = threading.RLock()      # the main thread, a sequence's tail thread and the writer's pool thread all touch _STATE_WRITER


def _write_state_file(path, compacts):
    from .frame_state import FrameState
    data = [FrameState.from_compact(c).serialize for c in compacts]
    tmp = str(path) + '.tmp'
    with open(tmp, 'wb') as fp:
        pickle.dump(data, fp, protocol=pickle.HIGHEST_PROTOCOL)
    os.replace(tmp, path)                                # readers never see a half-written file


def start_state_writer():
    """Starts the helper process (idempotent; one at a time: the lock keeps two threads from each starting one)."""
    with _STATE_LOCK:
        if _STATE_WRITER['proc'] is not None and _STATE_WRITER['proc'].poll() is None:
            return _STATE_WRITER['proc']
        import subprocess
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
        env['PYTHONPATH'] = root + os.pathsep + env.get('PYTHONPATH', '')
        try:
            _STATE_WRITER['proc'] = subprocess.Popen([sys.executable, '-m', 'vilgod_amd.state_writer'], stdin=subprocess.PIPE,
                                                     stdout=subprocess.PIPE, env=env, cwd=root)
        except OSError:
            _STATE_WRITER['proc'] = None
        return _STATE_WRITER['proc']


def _write_state_via_helper(path, compacts):
    import struct
    p = start_state_writer()
    if p is not None:
        try:
            blob = pickle.dumps((str(path), compacts), protocol=pickle.HIGHEST_PROTOCOL)
            p.stdin.write(struct.pack('<q', len(blob)))
            p.stdin.write(blob)
            p.stdin.flush()
            del blob
            head = p.stdout.read(8)
            if len(head) == 8:
                (n,) = struct.unpack('<q', head)
                status, val = pickle.loads(p.stdout.read(n))
                if status == 'ok':
                    return
                raise RuntimeError(f'state writer process, writing {path}: {val}')
        except (BrokenPipeError, EOFError, OSError, ValueError):
            pass
        import logging
        logging.getLogger('vilgod_amd.state_writer').warning('state writer process ended; writing %s in-process', path)
        try:
            p.kill()
        except Exception:               # noqa: BLE001
            pass
        with _STATE_LOCK:
            if _STATE_WRITER['proc'] is p:
                _STATE_WRITER['proc'] = None
    _write_state_file(path, compacts)


def wait_state_writes(path=None):
    """Blocks until the outstanding background state write (if any; with `path`: only a write of that file) is on disk; re-raises
    its error."""
    with _STATE_LOCK:
        if path is not None and _STATE_WRITER['path'] != str(path):
            return
        fut, _STATE_WRITER['pending'] = _STATE_WRITER['pending'], None
        written = _STATE_WRITER['path']
    if fut is not None:
        try:
            fut.result()
        except Exception as e:          # noqa: BLE001  (name the file: the error surfaces in whichever thread waits next; KeyboardInterrupt / SystemExit pass through)
            raise RuntimeError(f'background write of the sequence state {written} failed: {e}') from e


def shutdown_state_writer():
    wait_state_writes()
    with _STATE_LOCK:
        p, _STATE_WRITER['proc'] = _STATE_WRITER['proc'], None
    if p is not None:
        try:
            p.stdin.close()
            p.wait(timeout=5)
        except Exception:               # noqa: BLE001
            p.kill()


def _submit_state_write(path, compacts):
    from concurrent.futures import ThreadPoolExecutor
    while True:
        wait_state_writes()                 # takes the pending future out under the lock and waits for it outside
        with _STATE_LOCK:
            if _STATE_WRITER['pending'] is not None:
                continue                    # another thread submitted meanwhile: wait for that one too, so no future is ever dropped unawaited
            if _STATE_WRITER['pool'] is None:
                import atexit
                _STATE_WRITER['pool'] = ThreadPoolExecutor(max_workers=1, thread_name_prefix='vilgod-state-writer')
                atexit.register(shutdown_state_writer)
            _STATE_WRITER['path'] = str(path)
            _STATE_WRITER['pending'] = _STATE_WRITER['pool'].submit(_write_state_via_helper, path, compacts)
            return


class ZeroShotDetector:
    def __init__(self, dataset, name, cfg, logger, cluster_model=None, clip_model=None, pipeline=None):
        self.cfg, self.name, self.dataset, self.logger = cfg, name, dataset, logger
        self.lenght = dataset.sequence_length          # (sic) attribute name of the reference, :31
        self.rank, self.world_size = vdist.world()
        dev = cfg.get('device', {}) if hasattr(cfg, 'get') else {}
        shard = dev.get('shard', 'auto')
        if self.world_size > 1 and shard not in ('frames', 'sequences'):
            # `auto` depends on the number of sequences of the run, which only the caller knows (tools/preprocess_data.py resolves it
            # and writes it back into cfg.device).  Guessing here could leave this rank sharding frames -- and waiting in collectives --
            # while the caller deals whole sequences to the ranks: a hang.  Fail loudly instead.
            raise ValueError(f"device.shard={shard!r} is unresolved with {self.world_size} ranks: pass 'frames' or 'sequences' "
                             "(tools/preprocess_data.py resolves 'auto' before it builds the detector)")
        if shard == 'sequences':
            self.rank, self.world_size = 0, 1            # whole sequences per rank (tools/preprocess_data.py): nothing is exchanged inside one
        if pipeline is None:
            margs = [t for t in cfg.pipeline if t['name'] == 'mask_ground_points']
            ga = margs[0]['args'] if margs else {'min_range': 1.5, 'z_offset': 1.723}
            device = f"cuda:{torch.cuda.current_device()}"
            pipeline = PseudoLabelPipeline(cfg.preprocessor, device=device, vit_dtype=dev.get('vit_dtype', 'f16'),
                                           n_views=dev.get('n_views', 4), max_points=dev.get('max_points', 300_000),
                                           clip_model_path=cfg.paths.clip_model, min_range=ga['min_range'],
                                           z_offset=ga['z_offset'], plane_seed=dev.get('plane_seed', 666), clip=clip_model,
                                           box_mode=dev.get('box_mode', 'reference'), box_workers=dev.get('box_workers', 4),
                                   angle_mode=dev.get('angle_mode', 'reference'), vit_graph=bool(dev.get('vit_graph', False)),
                                           hierarchy=dev.get('hierarchy', None))
        self.pipe = pipeline
        self.sequence_data_dir_path = Path(cfg.paths.sequence_data)
        self.my_frames = vdist.shard_frames(self.lenght, self.rank, self.world_size)
        self.lidar_frame_list = []
        self._dev = {}                                   # fnr -> dict of device tensors kept across stages
        self._scores = {}                                # fnr -> [n_crops, K] class probabilities
        self._ent = {}                                   # fnr -> (kept entropy scores, indices), own + halo frames
        self.n_workers = int(dev.get('frames_in_flight', 6))
        self.sync_every_stage = bool(dev.get('sync_every_stage', False))
        self.async_state_write = bool(dev.get('async_state_write', True)) and not self.sync_every_stage
        if self.async_state_write and self.rank == 0:
            start_state_writer()                         # (numpy import in the helper: ready long before the first write)
        self.stage_ms = {}                               # stage name -> ms per (own) frame of the last process()
        self.detail_ms = {}                              # VILGOD_STAGE_DETAIL=1: wall ms of the parts of the host-heavy stages (whole sequence)
        self._detail_on = os.environ.get('VILGOD_STAGE_DETAIL', '0') == '1'
        self._dirty = False
        self._snapshot, self._frozen = None, False       # serialised frames as of the last stage that synchronises (propagate_labels)
        self._written_early = False                      # the frozen snapshot is already with the background writer
        self.tracker = None                              # vilgod_amd.tracking.Tracker after track_clusters (in memory only, like upstream)
        self._tab = None                                 # tracking.DetectionTable shared by fit_bounding_boxes_simple / propagate_labels
        self._host_X = {}                                # fnr -> points_ref_wo_ground on the host (tracking stages are host logic)
        self._host_X3 = {}                               # fnr -> its x, y, z columns, contiguous (the track branch of the box stage)
        self._box_prefetch = {}                          # fnr -> (rows, future of their static boxes), filled by `classification`
        self.init_lidar_frames()
        try:
            self.sync_lidar_frames(mode='load')
        except Exception:                                # the reference swallows load errors too, :45-48
            pass
        self.logger.info(f'Loaded {len(self.lidar_frame_list)} lidar frames')
        self.detection_3d_result_list = []

    @contextlib.contextmanager
    def _part(self, name):
        if not self._detail_on:
            yield
            return
        t0 = time.perf_counter()
        try:
            yield
        finally:
            self.detail_ms[name] = self.detail_ms.get(name, 0.0) + 1000.0 * (time.perf_counter() - t0)

    # ------------------------------------------------------------------------------------------------
    def init_lidar_frames(self):
        self.sequence_data_dir_path.mkdir(parents=True, exist_ok=True)
        ref_pose = self.dataset.sequence_infos[0]['pose']
        for fnr in range(self.lenght):
            self.lidar_frame_list.append(FrameState(fnr, self.dataset.sequence_infos[fnr]['pose'], ref_pose))

    def _points(self, fnr):
        d = self._dev.setdefault(fnr, {})
        if 'pts' not in d:
            d['pts'] = self.pipe.upload(self.dataset.get_lidar_points(fnr))
        return d['pts']

    def _ref_and_nonground(self, fnr):
        """points_ref and points_ref_wo_ground (lidar_frame.py:66-79) on the device."""
        d = self._dev.setdefault(fnr, {})
        if 'X' not in d:
            fs = self.lidar_frame_list[fnr]
            pts = self._points(fnr)
            d['ref'] = self.pipe.to_ref(pts, fs.transform_to_ref)
            mask = torch.ones(pts.shape[0], dtype=torch.bool, device=pts.device)
            mask[torch.from_numpy(np.asarray(fs.ground_point_indices)).to(pts.device)] = False
            d['X'] = d['ref'][mask].contiguous()
            d.pop('pts', None)
        return d['ref'], d['X']

    def _cluster_lists(self, fnr, rows=None):
        fs = self.lidar_frame_list[fnr]
        if rows is None:
            index, seg = fs.index, fs.seg_off
        else:
            parts = [fs.cluster_index(c) for c in rows]
            index = np.concatenate(parts) if parts else np.zeros(0, np.int32)
            seg = np.r_[0, np.cumsum([len(p) for p in parts])].astype(np.int32)
        dev = self.pipe.device
        return torch.from_numpy(np.ascontiguousarray(index, dtype=np.int32)).to(dev), torch.from_numpy(seg).to(dev)

    def _for_frames(self, frames, body, prepare=None, in_order=None):
        """Run `body(pipe, fnr)` for every frame -- sequentially on the caller's stream, or, with device.frames_in_flight > 1,
        on worker threads that own a stream and their own handles (cluster buffers, ViT workspace): `prepare(fnr)` is then
        called for all frames first on the caller's stream (device data the bodies share), followed by one event the workers
        wait for.  Bodies of different frames touch different FrameState objects.
        in_order(fnr): called on the CALLER's thread for every frame, in the order of `frames`, as soon as that frame and all frames
        before it are done -- sequential consumers of the frame pass (the tracker) run there while the workers are busy with later
        frames, instead of as a stage of their own afterwards."""
        frames = list(frames)
        nw = min(self.n_workers, len(frames))
        if prepare is not None:
            for f in frames:
                prepare(f)
        if nw <= 1:
            for f in frames:
                body(self.pipe, f)
                if in_order is not None:
                    in_order(f)
            return
        p = self.pipe
        workers = p._ensure_workers(nw)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(p.device))

        # The frames go, in order, to whichever worker is free (one shared queue; a worker's handles are only ever used on its own
        # thread).  A fixed deal (frame i -> worker i % nw) ends when the slowest worker has worked off its share: the workers do not
        # run in step (PseudoLabelPipeline.process_frames).
        import queue
        todo, done = queue.Queue(), queue.Queue()
        for f in frames:
            todo.put(f)

        def drain(worker):
            errors = []
            with torch.cuda.stream(worker.stream):
                worker.stream.wait_event(ev)
                while True:
                    try:
                        f = todo.get_nowait()
                    except queue.Empty:
                        break
                    try:
                        body(worker, f)
                        worker.stream.synchronize()
                        done.put((f, True))
                    except BaseException as e:      # noqa: BLE001  (the other frames still run; the first error is raised below)
                        errors.append(e)
                        done.put((f, False))
            return errors

        futs = [w.thread.submit(drain, w) for w in workers[:nw]]
        errs = []
        if in_order is not None:
            # frames finish out of order (whichever worker is free takes the next one): hand them on in the order of `frames`
            finished, nxt, broken = {}, 0, False
            for _ in range(len(frames)):
                f, ok = done.get()
                finished[f] = ok
                while nxt < len(frames) and frames[nxt] in finished:
                    broken = broken or not finished[frames[nxt]]
                    if not broken:
                        try:
                            in_order(frames[nxt])
                        except BaseException as e:      # noqa: BLE001
                            errs.append(e)
                            broken = True
                    nxt += 1
        for fut in futs:
            errs = fut.result() + errs
        if errs:
            raise errs[0]

    def _exchange_states(self):
        """Every rank receives the serialised states of the frames it does not own (small pickled objects)."""
        if self.world_size == 1:
            return
        states = {}
        for part in vdist.gather_objects({f: self.lidar_frame_list[f].serialize for f in self.my_frames}):
            states.update(part)
        for f, data in states.items():
            if f not in self.my_frames:
                self.lidar_frame_list[f].clear_detections()
                self.lidar_frame_list[f].sync(data)

    def _points_host(self, fnr):
        if fnr not in self._host_X:
            self._host_X[fnr] = self._ref_and_nonground(fnr)[1].cpu().numpy()
        return self._host_X[fnr]

    def sync_lidar_frames(self, mode='save', final=False):
        path = self.sequence_data_dir_path / f'{self.name}{self.cfg.postfix.sequence_data}'
        if mode == 'save':
            if self.world_size > 1 and not final:
                return                                   # several ranks: written once per run, after the gather (process())
            if not final:
                self._snapshot, self._frozen = None, False   # a stage synchronised: the live state is what the file holds from here on
                self._written_early = False                  # (and a file written from an earlier freeze is out of date)
            if not self.sync_every_stage and not final:
                self._dirty = True                       # one write at the end of process() instead of one per stage
                return
            snapshot = self._snapshot if final else None
            if self.world_size > 1:
                if not (final and self._frozen):         # (the same decision on every rank: the exchange is a collective)
                    self._exchange_states()
                if self.rank != 0:
                    self._dirty = False
                    return
            if final and snapshot is not None and getattr(self, '_written_early', False):
                self._dirty = False                      # propagate_labels handed exactly this snapshot to the background writer already
                return
            with self._part('state.serialize'):
                data = snapshot if snapshot is not None else [f.compact() for f in self.lidar_frame_list]
            with self._part('state.pickle'):
                if final and self.async_state_write:
                    _submit_state_write(path, data)      # (the one write at the end of process(): in the background, see _STATE_WRITER)
                else:
                    wait_state_writes()
                    _write_state_file(path, data)
            self._dirty = False
        elif mode == 'load':
            wait_state_writes(path)                      # (a background write of this very file may still be running)
            if path.exists():
                with open(path, 'rb') as fp:
                    data = pickle.load(fp)
                for fnr, frame_data in enumerate(data):
                    self.lidar_frame_list[fnr].sync(frame_data)
        else:
            raise NotImplementedError(f'Mode {mode} not implemented!')

    # stages that are host logic once the tracker has run (no collective, no shared GPU handle): the tail of the shipped stage list
    HOST_TAIL = ('fit_bounding_boxes_simple', 'propagate_labels', 'evaluate_sequence')

    def split_stages(self):
        """(front, back) of pipeline_active for tools/preprocess_data.py's sequence overlap: `back` = the trailing run of host-only
        stages (HOST_TAIL) that may run on a background thread while the NEXT sequence's GPU stages run; empty when that is not safe
        (several ranks: the stages hold collectives; per-stage files; device.overlap_sequences: false)."""
        active = list(self.cfg.pipeline_active)
        dev = self.cfg.get('device', {}) if hasattr(self.cfg, 'get') else {}
        if self.world_size != 1 or self.sync_every_stage or not dev.get('overlap_sequences', True):
            return active, []
        k = len(active)
        while k > 0 and active[k - 1] in self.HOST_TAIL:
            k -= 1
        return active[:k], active[k:]

    def back_is_host_only(self):
        """After process('front'): does the `back` part stay off the GPU?  The box stage does only when every precondition of its
        host-only route holds -- checked here, not assumed: the tracker has valid tracks, the boxes come from the helper
        processes (box_mode 'reference'), every tracked (frame, row) is covered by the static-box request `classification` sent ahead
        (`_box_prefetch`; empty on a resumed run that skipped classification, or when the two stages' `valid_only` differ), and every
        tracked frame's points are on the host (`_host_X`) and gathered on the device (`_dev[fnr]['X']`, only handed through).  Anything
        else would launch kernels from the tail thread -- on the thread's default device, with the next sequence's stages running --
        so the caller then runs the tail synchronously on the main thread."""
        _, back = self.split_stages()
        if not back:
            return False
        if 'fit_bounding_boxes_simple' in back:
            if self.tracker is None or len(self.tracker.tracks_valid) == 0 or self.pipe.box_mode != 'reference':
                return False
            tracked = {}
            for t in self.tracker.tracks_valid:
                for fnr, row in t.source:
                    tracked.setdefault(fnr, set()).add(int(row))
            for fnr, rows in tracked.items():
                pre = self._box_prefetch.get(fnr)
                if pre is None or not rows <= set(int(r) for r in pre[0]):
                    return False
                if fnr not in self._host_X or 'X' not in self._dev.get(fnr, {}):
                    return False
        return True

    def process(self, part='all'):
        """part: 'all' (the reference's process(), zero_shot_detector.py:58-69), or 'front' / 'back' = the two halves of split_stages()
        (tools/preprocess_data.py runs `back` of sequence s on a thread while `front` of sequence s + 1 runs)."""
        active = list(self.cfg.pipeline_active)
        front, back = self.split_stages() if part != 'all' else (active, [])
        names = {'all': active, 'front': front, 'back': back}[part]
        if part != 'back':
            self.logger.info(f'Processing sequence: {self.name}')
            self._fused = self._fusion_plan()
            if self._fused:
                self.logger.info(f"  per-frame work of {' + '.join(self._fused)} runs inside spatial_clustering's frame pass")
        available = [t['name'] for t in self.cfg.pipeline]
        for task_name in names:
            if task_name in available and hasattr(self, task_name):
                t0 = time.perf_counter()
                getattr(self, task_name)(**self.cfg.pipeline[available.index(task_name)]['args'])
                if part != 'back':
                    torch.cuda.synchronize()             # (the back half is host logic; a device-wide wait there would wait for the next sequence's kernels)
                ms = 1000.0 * (time.perf_counter() - t0) / max(len(self.my_frames), 1)
                self.stage_ms[task_name] = self.stage_ms.get(task_name, 0.0) + ms
                self.logger.info(f'  stage {task_name}: {ms:.2f} ms per frame')
            else:
                self.logger.warning(f'{task_name} NOT FOUND!!!')
        if part == 'front':
            return
        # the sequence-state pickle (zero_shot_detector.py:105-114): the reference rewrites it after every stage; here it is
        # written once per run unless device.sync_every_stage asks for the reference's per-stage files (same final content).
        # With several ranks it is written by rank 0 after the states were gathered, whatever stages ran.
        if self._dirty or self.world_size > 1:
            t0 = time.perf_counter()
            self.sync_lidar_frames(final=True)
            self.stage_ms['write_sequence_state'] = 1000.0 * (time.perf_counter() - t0) / max(len(self.my_frames), 1)
        self.logger.info(f'Finished processing sequence: {self.name}')

    def _fusion_plan(self):
        """Which later stages' per-frame work rides along in spatial_clustering's frame pass (device.fuse_stages, default on).
        The reference runs stage after stage over all frames (zero_shot_detector.py:58-69); on one GPU that leaves the device idle
        while a stage's host part runs (cluster hierarchy, votes) and the host idle while the ViT runs.  filter_detections and
        classification of a frame read only that frame's clusters, so a worker continues with them right after it has clustered
        the frame -- one frame's GEMMs cover another frame's clustering, as in PseudoLabelPipeline.process_sequence.  The stages
        themselves still run in their turn and find the frames done (their own skip rules: `filtered`, `key in cls`), so a
        stage list, a resumed run or a `force: True` behave as before; the results are the same objects either way.
        Only stages between which nothing but track_clusters (reads `valid`, writes tracks) sits are fused, never with `force`
        (the stage would redo the work), and not with device.sync_every_stage (per-stage files on disk)."""
        dev = self.cfg.get('device', {}) if hasattr(self.cfg, 'get') else {}
        active = list(self.cfg.pipeline_active)
        if not dev.get('fuse_stages', True) or self.sync_every_stage or 'spatial_clustering' not in active:
            return {}
        args = {t['name']: (t['args'] or {}) for t in self.cfg.pipeline}
        plan = {}
        for name in active[active.index('spatial_clustering') + 1:]:
            a = args.get(name, {})
            if name == 'filter_detections' and not a.get('force', False):
                plan[name] = a
            elif name == 'track_clusters':
                continue
            elif name == 'classification' and not a.get('force', False) and a.get('image_size', 224) == 224 \
                    and a.get('aggregation', 'voting') == 'voting' and ('filter_detections' in plan or not a.get('valid_only', False)):
                plan[name] = a
                break
            else:
                break
        return plan

    # ---- stages ------------------------------------------------------------------------------------------
    def mask_ground_points(self, min_range, z_offset, **kwargs):
        if all(f.ground_point_indices is not None for f in self.lidar_frame_list):
            return
        self.pipe.z_offset = float(z_offset)
        self.pipe.new_sequence()                         # one stateful Patchwork++ object per sequence, :137-140
        dev = self.cfg.get('device', {}) if hasattr(self.cfg, 'get') else {}
        handoff = dev.get('ground_handoff', 'chain') if self.world_size > 1 else 'replay'

        def run(frames):
            masks = []
            for fnr in frames:                           # sequential and stateful: queued in frame order on one stream
                pts = self._points(fnr) if fnr in mine else self.pipe.upload(self.dataset.get_lidar_points(fnr))
                masks.append((fnr, pts.shape[0], self.pipe.ground(pts)))
            for fnr, n, mask in masks:                   # read back afterwards: no host round trip between two passes
                fs = self.lidar_frame_list[fnr]
                fs.n_points = n
                fs.ground_point_indices = torch.nonzero(mask).squeeze(1).cpu().numpy()

        mine = set(self.my_frames)
        if handoff == 'chain':
            # the adaptive state is handed from rank to rank (vg_ground_export_state / set_state): every rank runs the ground
            # stage over its own block only; the ground sets of the other blocks arrive with the frame states later
            vdist.chain_ground_state(self.pipe.ground_model, lambda: run(self.my_frames), device=self.pipe.device)
            # the other blocks' ground sets (halo frames of the entropy window, the final state pickle) travel as bit masks:
            # 19 KB per 150k-point frame, small pickled objects like the result dicts -- not a data-path collective
            packed = {}
            for f in self.my_frames:
                fs = self.lidar_frame_list[f]
                m = np.zeros(fs.n_points, bool)
                m[fs.ground_point_indices] = True
                packed[f] = (fs.n_points, np.packbits(m))
            for part in vdist.gather_objects(packed):
                for f, (n, bits) in part.items():
                    if f not in mine:
                        fs = self.lidar_frame_list[f]
                        fs.n_points = n
                        fs.ground_point_indices = np.flatnonzero(np.unpackbits(bits, count=n))
        else:
            run(range(self.lenght))                      # every rank replays all frames (no communication)
        self.sync_lidar_frames()

    def _entropy_full(self, fnr):
        """LidarFrame.entropy_scores (lidar_frame.py:111-118) on the device, or None before calculate_entropy_scores."""
        d = self._dev.setdefault(fnr, {})
        if 'ent' not in d:
            kept = self._ent.get(fnr)
            if kept is None:
                fs = self.lidar_frame_list[fnr]
                kept = (fs.entropy_scores, fs.entropy_indices) if fs.entropy_scores is not None else None
            if kept is None:
                return None
            from .entropy import full_scores
            n = self._ref_and_nonground(fnr)[1].shape[0]
            d['ent'] = full_scores(n, kept[0], kept[1], device=self.pipe.device)
        return d['ent']

    def calculate_entropy_scores(self, n_neighbouring_frames, **kwargs):
        """zero_shot_detector.py:153-195.  Every rank scores its own frames plus the neighbours its two-frame
        clustering will read; the sliding 15-frame buffer of the reference becomes "each frame's grid is built once
        and queried by every frame whose window contains it"."""
        from .entropy import EntropyScorer, TwoFrameClusterer
        if all(f.entropy_scores is not None for f in self.lidar_frame_list) and not kwargs.get('force', False):
            return
        if any(f.ground_point_indices is None for f in self.lidar_frame_list):
            raise RuntimeError('calculate_entropy_scores needs mask_ground_points first (points_ref_wo_ground)')
        include_ground = kwargs.get('include_ground_points', False)
        scorer = EntropyScorer(self.pipe.cluster_model, n_neighbouring_frames=n_neighbouring_frames,
                               **{k: v for k, v in kwargs.items() if k in ('skip_frames', 'max_neighbor_point_dist', 'max_neighbor_points')})
        cl = [t for t in self.cfg.pipeline if t['name'] == 'spatial_clustering']
        n_frames = (cl[0]['args'] or {}).get('n_frames', 1) if cl else 1
        queries = set(self.my_frames)
        if n_frames > 1 and self.lenght >= n_frames:
            two = TwoFrameClusterer(self.pipe.cluster_model, n_frames=n_frames)
            for f in self.my_frames:
                queries.update(two.used_frames(f, self.lenght))
        X_list = [None] * self.lenght
        for f in scorer.frames_needed(queries, self.lenght):
            ref, X = self._ref_and_nonground(f)
            X_list[f] = ref if include_ground else X
        if include_ground:
            raise NotImplementedError('include_ground_points: scores would index points_ref, which nothing downstream reads')
        mapper = (lambda items, fn: self.pipe.map_workers(items, lambda w, it: fn(w.cluster_model, it), self.n_workers)) \
            if self.n_workers > 1 else None
        H = scorer.score_sequence(X_list, queries=queries, mapper=mapper)
        mine = set(self.my_frames)
        for f, h in H.items():
            kept = scorer.reduce(h)
            self._ent[f] = kept
            if f in mine:
                self.lidar_frame_list[f].entropy_scores, self.lidar_frame_list[f].entropy_indices = kept
            self._dev.get(f, {}).pop('ent', None)
        for f in range(self.lenght):                      # halo frames: keep X only where clustering will read it
            if f not in queries:
                self._dev.pop(f, None)
        self.sync_lidar_frames()

    def spatial_clustering(self, **kwargs):
        """zero_shot_detector.py:197-256: n_frames == 1 clusters points_ref_wo_ground[..., :3]; n_frames > 1 (the shipped
        default, preprocessing.yaml:68) clusters the entropy-guided 5-D union of n_frames frames and transfers the labels
        to the frame's points by nearest neighbour.  Either way `Detection.static` comes from the entropy scores when
        they exist (lidar_frame.py:238-243)."""
        from .entropy import TwoFrameClusterer
        from .frame_state import static_from_entropy
        n_frames = kwargs.get('n_frames', 1)
        force = kwargs.get('force', False)
        two = None
        if n_frames > 1:
            if self.lenght < n_frames:
                raise RuntimeError(f'spatial_clustering: n_frames={n_frames} but the sequence has {self.lenght} frames')
            dev = self.cfg.get('device', {}) if hasattr(self.cfg, 'get') else {}
            two = TwoFrameClusterer(self.pipe.cluster_model, n_frames=n_frames, seed=int(dev.get('subsample_seed', 0)))
        ecfg = self.cfg.preprocessor.clustering.get('entropy_score_filter', None) \
            if hasattr(self.cfg.preprocessor.clustering, 'get') else None
        todo = [f for f in self.my_frames if self.lidar_frame_list[f].ground_point_indices is not None
                and (self.lidar_frame_list[f].n_detections == 0 or force)]
        X_list, ent_list, parts = [None] * self.lenght, [None] * self.lenght, None
        if two is not None:
            need = sorted({g for f in todo for g in two.used_frames(f, self.lenght)})
            for g in need:
                X_list[g] = self._ref_and_nonground(g)[1]
                ent_list[g] = self._entropy_full(g)
                if ent_list[g] is None:
                    raise RuntimeError('spatial_clustering with n_frames > 1 reads the entropy scores: activate '
                                       'calculate_entropy_scores first (preprocessing.yaml:50)')
            n_used = min(n_frames, self.lenght)
            rows = self.pipe.map_workers(need, lambda w, g: TwoFrameClusterer(w.cluster_model, n_frames=n_frames, seed=two.seed)
                                         .frame_part(g, X_list[g], ent_list[g], n_used), self.n_workers)
            parts = {(g, n_used): r for g, r in zip(need, rows)}                                        # shared, read-only
        seed = two.seed if two is not None else 0

        def body(p, fnr):
            fs = self.lidar_frame_list[fnr]
            X = self._dev[fnr]['X']
            if two is not None:
                labels, probs = TwoFrameClusterer(p.cluster_model, n_frames=n_frames, seed=seed, parts=parts).labels(fnr, X_list, ent_list)
            else:
                labels, probs = p.cluster(X)
            fs.set_clusters(*pack_clusters(labels, probs, p.prob_threshold))     # lidar_frame.py:154-248
            self._box_prefetch.pop(fnr, None)                                    # (requests sent for the frame's previous clusters)
            ent = self._dev[fnr].get('ent')
            if ent is not None and fs.n_detections:
                fs.static = static_from_entropy(ent.cpu().numpy(), fs.index, fs.seg_off,
                                                percentile=float(ecfg['percentile']) if ecfg else 30.0,
                                                min_percentile_pp_score=float(ecfg['min_percentile_pp_score']) if ecfg else 0.5)
            if 'filter_detections' in fused and fs.n_detections:
                self._filter_frame(p, fnr)
                if track_rows is not None:
                    # the tracker's input (Detection.cluster_mass_center of the rows it will see): one small launch per frame, queued
                    # here under other frames' GEMMs instead of 199 launches + read-backs at the start of track_clusters
                    rows = np.flatnonzero(fs.valid) if track_rows == 'valid' else np.arange(fs.n_detections)
                    if len(rows):
                        d_index, d_seg = self._cluster_lists(fnr, rows)
                        self._track_med[fnr] = (rows, p.cluster_medians(X, d_index, d_seg).cpu().numpy())    # (host: the worker syncs anyway)
                    else:
                        self._track_med[fnr] = (rows, None)
            if cls_ctx is not None and fs.n_detections and cls_ctx['key'] not in fs.cls:
                self._classify_frame(p, fnr, cls_ctx)

        fused = getattr(self, '_fused', {})
        cls_ctx = self._classification_context(**fused['classification']) if 'classification' in fused else None
        track_rows = None
        active = list(self.cfg.pipeline_active)
        if 'filter_detections' in fused and 'track_clusters' in active and self.world_size == 1:
            targs = [t['args'] or {} for t in self.cfg.pipeline if t['name'] == 'track_clusters']
            track_rows = 'valid' if (targs and targs[0].get('valid_only', False)) else 'all'
        self._track_med = {}
        # the tracker as a streaming consumer of the frame pass (round 4): it is sequential over the frames and pure host logic on a few
        # hundred cluster medians per frame (~1 ms), so it runs on this thread, frame by frame in order, while the workers process the
        # later frames -- instead of ~1 ms per frame with the GPU idle afterwards.  Same inputs, same order, same tracker; the
        # track_clusters stage then finds its work done.  Only when every frame of the sequence goes through this pass.
        stream_tracker = (track_rows is not None and len(todo) == self.lenght and todo == list(range(self.lenght))
                          and getattr(self, '_stream_tracker_ok', True))
        self._streamed = None
        if stream_tracker:
            self._streamed = self._new_tracker()
            self._streamed_rows = track_rows
            self._med, self._cnt = {}, {}
        self._for_frames(todo, body, prepare=lambda f: (self._ref_and_nonground(f), self._entropy_full(f)),
                         in_order=self._track_frame if stream_tracker else None)
        if todo:
            self.sync_lidar_frames()

    def _filter_frame(self, p, fnr):
        """filter_detections for one frame on pipeline handle `p` (the main one or a worker's)."""
        fs = self.lidar_frame_list[fnr]
        ref, X = self._ref_and_nonground(fnr)
        if p._filters['use_plane']:
            gidx = torch.from_numpy(np.asarray(fs.ground_point_indices)).to(p.device)
            fs.ground_plane_model_ref = p.ground_plane(ref, gidx)             # lidar_frame.py:96-109
        else:
            fs.ground_plane_model_ref = np.array([0.0, 0.0, 1.0, 0.0])
        d_index, d_seg = self._cluster_lists(fnr)
        valid, _ = p.filter(X, d_index, d_seg, fs.ground_plane_model_ref)
        fs.valid = valid.cpu().numpy().astype(bool)
        fs.filtered = True
        self._box_prefetch.pop(fnr, None)                                        # (a request keyed on the previous valid rows)

    def filter_detections(self, **kwargs):
        force = kwargs.get('force', False)
        for fnr in self.my_frames:
            fs = self.lidar_frame_list[fnr]
            if fs.n_detections == 0 or (fs.filtered and not force):
                continue
            self._filter_frame(self.pipe, fnr)
        self.sync_lidar_frames()

    def _classification_context(self, image_size=224, aggregation='voting', **kwargs):
        if image_size != 224 or aggregation != 'voting':
            raise NotImplementedError('image_size 224 and voting aggregation (preprocessing.yaml) only')
        p = self.pipe
        active = list(self.cfg.pipeline_active)
        return {'key': kwargs.get('key', 'clip'), 'valid_only': kwargs.get('valid_only', False),
                'names': np.array(p.mapped_names, dtype=object), 'fine_names': np.array(p.class_list, dtype=object),
                # the static rectangles of the same clusters (box_mode='reference': a helper process per request) are computed while
                # the GPU encodes the crops; fit_bounding_boxes_simple collects them (they do not depend on the classes)
                'prefetch_boxes': (p.box_mode == 'reference' and 'fit_bounding_boxes_simple' in active and 'classification' in active
                                   and active.index('fit_bounding_boxes_simple') > active.index('classification'))}

    def _classify_frame(self, pw, fnr, ctx):
        """classification for one frame on pipeline handle `pw`."""
        p = self.pipe
        V = p.projection.num_views
        fs = self.lidar_frame_list[fnr]
        which = fs.valid.copy() if ctx['valid_only'] else np.ones(fs.n_detections, bool)
        rows = np.flatnonzero(which)
        if len(rows) == 0:
            return
        X = self._dev[fnr]['X']
        d_index, d_seg = self._cluster_lists(fnr, rows)
        if ctx['prefetch_boxes']:
            self._box_prefetch[fnr] = ([int(r) for r in rows], self._fit_rows(fnr, rows, X, wait=False))
        probs, top1, score = pw.classify(X, d_index, d_seg, fs.transform_to_ego)
        self._scores[fnr] = probs
        fine = top1.cpu().numpy().reshape(len(rows), V)
        sc = score.cpu().numpy().reshape(len(rows), V).astype(np.float32)
        mapped = p.fine_to_mapped[fine]
        win, final = vote(mapped, sc, p.mapped_names)
        fs.set_classes(ctx['key'], which, ctx['names'][mapped], ctx['fine_names'][fine], sc, ctx['names'][win], final)

    def classification(self, image_size=224, aggregation='voting', **kwargs):
        ctx = self._classification_context(image_size, aggregation, **kwargs)
        force = kwargs.get('force', False)
        todo = [f for f in self.my_frames if self.lidar_frame_list[f].n_detections > 0
                and (ctx['key'] not in self.lidar_frame_list[f].cls or force)]
        if ctx['prefetch_boxes']:
            for f in todo:
                self._points_host(f)                     # host copy of points_ref_wo_ground, fetched on the caller's thread
        self._for_frames(todo, lambda pw, fnr: self._classify_frame(pw, fnr, ctx), prepare=self._ref_and_nonground)
        self.sync_lidar_frames()

    def fit_bounding_boxes_simple(self, method, **kwargs):
        mname = method['name'] if isinstance(method, dict) else method.name
        if mname != 'minimum_bounding_rectangle':
            raise NotImplementedError(f'{mname}: only minimum_bounding_rectangle (the configured method) has a kernel')
        valid_only, fg_only = kwargs.get('valid_only', False), kwargs.get('fg_only', False)
        ckey = kwargs.get('classification_key', None)
        if self.tracker is not None and len(self.tracker.tracks_valid) > 0:
            # tracks available: static and moving objects are handled differently (zero_shot_detector.py:462-684)
            if not kwargs.get('force', False) and any(fs.boxes is not None and np.isfinite(fs.boxes[:, 0]).any()
                                                      for fs in self.lidar_frame_list):
                return                                   # boxes exist and force is off: upstream skips the stage (:424-432)
            for fs in self.lidar_frame_list:
                fs.boxes = None
            self._fit_boxes_tracked(valid_only)
            self._box_prefetch.clear()
            self._host_X.clear()                         # host copies of the frames' points: only this stage and the prefetch read them
            self.sync_lidar_frames()
            return
        jobs = []
        for fnr in self.my_frames:
            fs = self.lidar_frame_list[fnr]
            if fs.n_detections == 0 or (fs.boxes is not None and not kwargs.get('force', False)):
                continue
            which = fs.valid.copy() if valid_only else np.ones(fs.n_detections, bool)
            if fg_only and ckey is not None and ckey in fs.cls:
                e = fs.cls[ckey]
                which &= e['has'] & np.isin(e['name'].astype(str), self.dataset.class_names)
            rows = np.flatnonzero(which)
            fs.boxes = np.full((fs.n_detections, 7), np.nan)
            if len(rows) == 0:
                continue
            _, X = self._ref_and_nonground(fnr)
            jobs.append((fs, rows, self._boxes_of_rows(fnr, rows, X)))
        for fs, rows, fut in jobs:
            fs.boxes[rows] = fut.result()
        self._box_prefetch.clear()
        self._host_X.clear()
        self.sync_lidar_frames()

    def _boxes_of_rows(self, fnr, rows, X):
        """-> object with .result() -> [len(rows),7]: from the request `classification` already sent for this frame when it covers
        the rows, else a new request."""
        pre = self._box_prefetch.get(fnr)
        if pre is not None:
            pos = {r: i for i, r in enumerate(pre[0])}
            if all(int(r) in pos for r in rows):
                fut, sel = pre[1], [pos[int(r)] for r in rows]

                class _Sel:
                    def result(self_inner):
                        return np.asarray(fut.result())[sel]
                return _Sel()
        return self._fit_rows(fnr, rows, X, wait=False)

    def _fit_rows(self, fnr, rows, X, wait=True):
        """Static-branch boxes of clusters `rows` of frame fnr (pipeline.fit_boxes: reference or fast mode); wait=False returns an
        object with .result() (reference mode: the host part runs in a helper process meanwhile)."""
        from .boxes import _Done
        fs = self.lidar_frame_list[fnr]
        parts = [fs.cluster_index(c) for c in rows]
        index = np.concatenate(parts).astype(np.int32)
        seg = np.r_[0, np.cumsum([len(p) for p in parts])].astype(np.int32)
        if self.pipe.box_mode != 'reference':
            return self.pipe.fit_boxes(X, index, seg) if wait else _Done(self.pipe.fit_boxes(X, index, seg))
        fut = self.pipe.fit_boxes_async(X, index, seg, xy_host=self._points_host(fnr))
        return fut.result() if wait else fut

    def evaluate_sequence(self, modes=('detection_3d',), logger=None, **kwargs):
        key = kwargs.get('classification_key', 'clip')
        local = {}
        if 'detection_3d' in modes:
            for fnr in self.my_frames:
                fs = self.lidar_frame_list[fnr]
                boxes, names, scores, moving = [], [], [], []
                if key in fs.cls and fs.boxes is not None:
                    e = fs.cls[key]
                    for c in range(fs.n_detections):
                        if fs.valid[c] and e['has'][c] and str(e['name'][c]) in self.dataset.class_names and not np.isnan(fs.boxes[c, 0]):
                            boxes.append(fs.boxes[c])
                            names.append(str(e['name'][c]))
                            scores.append(fs.final_score(key, c))       # numpy float32 or python float: np.array() below picks the dtype like upstream
                            moving.append(fs.static_track[c] == 0)          # static_track is not None and not static_track (:843)
                local[fnr] = {'boxes_lidar': self.pipe.boxes_to_ego(np.array(boxes).reshape(-1, 7), fs.transform_to_ego),
                              'name': np.array(names), 'score': np.array(scores),
                              'moving': np.array(moving, dtype=bool)}
        if self.world_size > 1:
            # the one data-path collective: class scores of every crop; results/states are small objects
            self._scores = vdist.gather_scores(self._scores, n_classes=len(self.pipe.class_list), device=self.pipe.device)
            merged = {}
            for part in vdist.gather_objects(local):
                merged.update(part)
            local = merged
        self.detection_3d_result_list = [local[f] for f in sorted(local)]

    # ---- tracking, motion-aware boxes, label propagation (host logic, vilgod_amd/tracking.py) -------------
    def track_clusters(self, **kwargs):
        """zero_shot_detector.py:298-327.  The tracker is sequential over the whole sequence and lives in memory only (upstream
        never writes `tid`); with several ranks every rank first receives all frame states and runs the same deterministic
        tracker, so the later stages can stay sharded."""
        valid_only = kwargs.get('valid_only', False)
        self._exchange_states()
        self._tab = None
        streamed = getattr(self, '_streamed', None)
        if streamed is not None and self._streamed_rows == ('valid' if valid_only else 'all'):
            # the frame pass fed the tracker frame by frame (spatial_clustering, _track_frame): nothing left but to close the open tracks
            self.tracker, self._streamed = streamed, None
            with self._part('track.tracker'):
                self.tracker.finish()
            self.logger.info(f'  tracks: {len(self.tracker.tracks)} ({sum(len(t) >= self.tracker.min_length for t in self.tracker.tracks)} of length >= {self.tracker.min_length})')
            return
        self._streamed = None
        self.tracker = self._new_tracker()
        # Detection.cluster_mass_center (objects.py:121-123) of every detection the tracker sees: one kernel launch per frame
        # (vg_cluster_medians, exact np.median semantics), queued for all frames before the first result is read back
        med, cnt, pending = {}, {}, []
        with self._part('track.medians_queue'):
            for fs in self.lidar_frame_list:
                rows = np.flatnonzero(fs.valid) if valid_only else np.arange(fs.n_detections)
                pre = getattr(self, '_track_med', {}).pop(fs.fnr, None)
                if pre is not None and np.array_equal(pre[0], rows):
                    pending.append((fs, rows, pre[1]))                   # queued by the frame pass (spatial_clustering's body)
                elif len(rows):
                    X = self._ref_and_nonground(fs.fnr)[1]
                    d_index, d_seg = self._cluster_lists(fs.fnr, rows)
                    pending.append((fs, rows, self.pipe.cluster_medians(X, d_index, d_seg)))
                else:
                    pending.append((fs, rows, None))
        self._med = med
        with self._part('track.tracker'):
            for fs, rows, d_med in pending:
                keys = [(fs.fnr, int(r)) for r in rows]
                if keys:
                    m = d_med if isinstance(d_med, np.ndarray) else d_med.cpu().numpy()
                    for j, k in enumerate(keys):
                        med[k], cnt[k] = m[j], int(fs.seg_off[k[1] + 1] - fs.seg_off[k[1]])
                centers = np.array([med[k] for k in keys]) if keys else np.zeros((0, 5), np.float32)
                self.tracker.next(fs.fnr, keys, centers, [cnt[k] for k in keys], lambda k: (med[k], cnt[k]))
            self.tracker.finish()
        self.logger.info(f'  tracks: {len(self.tracker.tracks)} ({sum(len(t) >= self.tracker.min_length for t in self.tracker.tracks)} of length >= {self.tracker.min_length})')

    def _new_tracker(self):
        from .tracking import Tracker
        tcfg = self.cfg.preprocessor.tracking.cluster
        assign = tcfg['assignment'] if isinstance(tcfg, dict) else tcfg.assignment
        if (assign['method'] if isinstance(assign, dict) else assign.method) != 'assign_detections_greedy':
            raise NotImplementedError('tracking.cluster.assignment.method: only assign_detections_greedy (the shipped configuration)')
        g = (lambda k, d=None: tcfg.get(k, d)) if hasattr(tcfg, 'get') else (lambda k, d=None: getattr(tcfg, k, d))
        return Tracker(mode=g('mode', 'cluster_center'), max_distance=(assign['max_distance'] if isinstance(assign, dict) else assign.max_distance),
                       min_length=g('min_length', 5), max_missed=g('max_missed', 3))

    def _track_frame(self, fnr):
        """One frame into the streaming tracker (frames arrive in order, on the dispatcher's thread): Tracker.next of track_clusters."""
        fs = self.lidar_frame_list[fnr]
        rows, m = self._track_med.pop(fnr, (np.zeros(0, np.int64), None))
        keys = [(fnr, int(r)) for r in rows]
        med, cnt = self._med, self._cnt
        for j, k in enumerate(keys):
            med[k], cnt[k] = m[j], int(fs.seg_off[k[1] + 1] - fs.seg_off[k[1]])
        centers = np.array([med[k] for k in keys]) if keys else np.zeros((0, 5), np.float32)
        self._streamed.next(fnr, keys, centers, [cnt[k] for k in keys], lambda k: (med[k], cnt[k]))

    def _cluster_points_host(self, key):
        fnr, row = key
        x3 = self._host_X3.get(fnr)
        if x3 is None:                                   # x, y, z of the frame's points, contiguous: the gathers below read 12-byte rows
            x3 = self._host_X3[fnr] = np.ascontiguousarray(self._points_host(fnr)[:, :3])
        return x3.take(self.lidar_frame_list[fnr].cluster_index(row), axis=0)

    def _fit_boxes_tracked(self, valid_only):
        """The track branch of fit_bounding_boxes_simple (zero_shot_detector.py:463-684): the per-detection rectangle boxes come
        from the GPU kernel (one launch per frame over all tracked clusters), the motion logic is host code."""
        from .tracking import DetectionTable, fit_track_boxes
        tracked = {}
        for t in self.tracker.tracks_valid:
            for fnr, row in t.source:
                tracked.setdefault(fnr, set()).add(row)
        gpu_box = {}
        jobs = []
        with self._part('boxes.request_static'):
            for fnr, rows in tracked.items():                # every frame's request goes out before the first answer is awaited
                rows = sorted(rows)
                _, X = self._ref_and_nonground(fnr)
                jobs.append((fnr, rows, self._boxes_of_rows(fnr, rows, X)))
        with self._part('boxes.await_static'):
            for fnr, rows, fut in jobs:
                for r, b in zip(rows, fut.result()):
                    gpu_box[(fnr, r)] = b
        with self._part('boxes.table'):
            tab = DetectionTable()
            for fs in self.lidar_frame_list:
                for r in range(fs.n_detections):
                    tab.valid[(fs.fnr, r)] = bool(fs.valid[r])
        med = getattr(self, '_med', None) or {}
        with self._part('boxes.fit_track_boxes'):
            # the moving tracks' boxes (per entry: a dozen small numpy calls, ~70 us) are computed by the box helper processes, track
            # by track, while this thread gathers the next track's points; same function, same numpy -> same boxes
            from . import boxes as _boxes
            nproc = self.pipe.box_workers if self.pipe.box_mode == 'reference' else 0
            self._host_X3 = {}
            fit_track_boxes(self.tracker, tab, self._cluster_points_host, lambda k: bool(self.lidar_frame_list[k[0]].static[k[1]]),
                            lambda f: self.lidar_frame_list[f].transform_to_ego, static_box_of=gpu_box.__getitem__,
                            median_of=(med.__getitem__ if med else None),
                            moving_async=(lambda p, d, e, c: _boxes.submit_moving_boxes(p, d, e, c, n_procs=nproc)) if nproc > 0 else None,
                            max_pending=2 * max(nproc, 1))
            self._host_X3 = {}
        self._tab = tab
        with self._part('boxes.write_back'):
            self._write_back_tracked()

    def _write_back_tracked(self, key=None):
        tab = self._tab
        for fs in self.lidar_frame_list:
            if fs.n_detections == 0:
                continue
            if fs.boxes is None:
                fs.boxes = np.full((fs.n_detections, 7), np.nan)
            for r in range(fs.n_detections):
                k = (fs.fnr, r)
                if k in tab.box:
                    fs.boxes[r] = tab.box[k]
                st = tab.static_track.get(k)
                if st is not None:
                    fs.static_track[r] = int(bool(st))
                fs.valid[r] = tab.valid.get(k, bool(fs.valid[r]))
                if key is not None and key in fs.cls and k in tab.name and fs.cls[key]['has'][r]:
                    fs.cls[key]['name'][r] = tab.name[k]
                    fs.set_final_score(key, r, tab.score[k])

    def propagate_labels(self, **kwargs):
        """zero_shot_detector.py:686-824."""
        from .tracking import propagate_labels
        if self.tracker is None or self._tab is None:
            self.logger.warning('propagate_labels needs track_clusters and fit_bounding_boxes_simple in the same run -- skipped')
            return
        key = kwargs.get('classification_key', 'clip')
        self._exchange_states()                         # the class results of the other ranks' frames
        tab = self._tab
        for fs in self.lidar_frame_list:
            if key in fs.cls:
                e = fs.cls[key]
                for r in np.flatnonzero(e['has']):
                    tab.name[(fs.fnr, int(r))], tab.score[(fs.fnr, int(r))] = str(e['name'][r]), fs.final_score(key, int(r))
        for t in self.tracker.tracks_valid:              # upstream would raise on a tracked detection without a class
            for i, k in enumerate(t.source):
                if not t.prediction[i] and k not in tab.name:
                    raise RuntimeError(f'propagate_labels: detection {k} is tracked but was not classified (run classification with '
                                       'the same valid_only setting as track_clusters)')
        # Upstream's propagate_labels changes the detections in memory and does NOT call sync_lidar_frames (:686-824 ends without
        # it; evaluate_sequence does not either): the sequence-state pickle keeps the state the last synchronising stage left
        # (fit_bounding_boxes_simple in the shipped list) while the result pickles carry the propagated labels.  The state is
        # written once at the end of process() here, so it is frozen now, before the labels are propagated.
        if (self._dirty or self.world_size > 1) and not self._frozen:
            self._frozen = True                          # every rank holds every frame's state here (_exchange_states above)
            if self.rank == 0:
                with self._part('propagate.snapshot'):
                    self._snapshot = [f.compact() for f in self.lidar_frame_list]     # (detached copies: see FrameState.compact)
                if self.async_state_write:
                    # what the file will hold is final from here on (no later stage of the reference synchronises): the background
                    # writer starts now, under the label propagation and the evaluation, instead of at the end of process()
                    _submit_state_write(self.sequence_data_dir_path / f'{self.name}{self.cfg.postfix.sequence_data}', self._snapshot)
                    self._written_early = True
        with self._part('propagate.logic'):
            propagate_labels(self.tracker, tab, lambda k: len(self.lidar_frame_list[k[0]].cluster_index(k[1])), self.dataset.class_names,
                             min_length=kwargs.get('min_length', 5))
        with self._part('propagate.write_back'):
            self._write_back_tracked(key)
