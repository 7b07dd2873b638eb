#!/usr/bin/env python3
"""Wall time of the entry point (tools/preprocess_data.py, default 9-stage list) on synthetic sequences WITHOUT a Python profiler
(tools/profile_cli.py runs under cProfile, which slows the host-heavy stages disproportionately).
    [SEQUENCES=1] python tools/time_cli.py [frames=199] [points=150000] [key=value ...]"""
import json, logging, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 199
points = int(sys.argv[2]) if len(sys.argv) > 2 else 150_000
nseq = int(os.environ.get('SEQUENCES', '1'))
procs = int(os.environ.get('PROCS', '1'))
if procs > 1 and 'RANK' not in os.environ:
    # PROCS=P: the entry point as P ranks on ONE GPU (device.processes_per_gpu=P, whole sequences per rank), started the way a user would
    # (torch.distributed.run) as a CHILD of this process, which never touches the GPU.  Wall time = first rank's loop start to last
    # rank's loop end (wall clock), minus nothing: the generator runs inside that window on every rank.
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={procs}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:] + [f'device.processes_per_gpu={procs}', 'device.shard=sequences']
    r = subprocess.run(cmd, env=dict(os.environ, TIME_CLI_JSON='1'), capture_output=True, text=True)
    runs = [json.loads(ln.split('TIME_CLI_JSON ', 1)[1]) for ln in r.stdout.splitlines() if 'TIME_CLI_JSON ' in ln]
    if r.returncode != 0 or len(runs) != procs:
        sys.stderr.write(r.stdout[-3000:] + r.stderr[-3000:])
        sys.exit(r.returncode or 1)
    t0, t1 = min(q['loop_started_at'] for q in runs), max(q['loop_ended_at'] for q in runs)
    seqs = sorted((s for q in runs for s in q['sequences']), key=lambda s: s['name'])
    tot_f = sum(s['frames'] for s in seqs)
    # the generator stands for disk IO and is excluded like in the one-process figure: every rank generates its world at the start of its
    # loop, all ranks at the same time; the SHORTEST rank's generator time is taken out of the window (conservative if they do not overlap)
    gen = min(q.get('generator_seconds', 0.0) for q in runs)
    wall = t1 - t0 - gen
    print(f"{procs} processes on one GPU, {len(seqs)} sequence(s), {tot_f} frames: {1000 * wall / tot_f:.2f} ms per frame = {tot_f / wall:.2f} frames/s "
          f"(first rank's loop start to last rank's loop end = {t1 - t0:.2f} s, minus {gen:.2f} s of synthetic generator at the start of every rank's loop); per rank: " +
          ', '.join(f"{sum(s['frames'] for s in q['sequences'])} frames in {q['loop_seconds']:.2f} s" for q in runs))
    for s in seqs:
        print(f"{s['name']}: frames {s['frames']}  front {1000 * s['front_seconds'] / s['frames']:.2f} + back {1000 * s['back_seconds'] / s['frames']:.2f} ms per frame  " +
              '  '.join(f'{k} {v:.2f}' for k, v in s['stage_ms_per_frame'].items()))
    if os.environ.get('TIME_CLI_JSON'):
        print('TIME_CLI_JSON ' + json.dumps({'loop_seconds': wall, 'window_seconds': t1 - t0, 'processes': procs, 'generator_seconds': gen,
                                            'loop_seconds_per_rank': [q['loop_seconds'] for q in runs],
                                            'state_write_wait_seconds': max(q.get('state_write_wait_seconds', 0.0) for q in runs), 'sequences': seqs}))
    sys.exit(0)
import preprocess_data  # noqa: E402
with tempfile.TemporaryDirectory() as root:
    logging.disable(logging.INFO)
    preprocess_data.main(['preprocessor=waymo', f'dataset.DATA_PATH={root}', f'dataset.SYNTHETIC.frames_per_sequence={frames}',
                          f'dataset.SYNTHETIC.points_per_frame={points}', f'dataset.SYNTHETIC.n_sequences={nseq}', f'end_sequence={nseq - 1}', f"dataset.SYNTHETIC.seed_stride={os.environ.get('SEED_STRIDE', '0')}",
                          f'device.max_points={2 * points}', 'paths.clip_model=/nonexistent'] + sys.argv[3:])
run = preprocess_data.LAST_RUN
tot_f = sum(q['frames'] for q in run['sequences'])
tot_s = run['loop_seconds']
print(f"{len(run['sequences'])} sequence(s), {tot_f} frames: {1000 * tot_s / tot_f:.2f} ms per frame = {tot_f / tot_s:.2f} frames/s "
      f"(wall time of the sequence loop without the synthetic generator; {run.get('state_write_wait_seconds', 0.0):.3f} s of it waiting for the last background state write)")
for seq in run['sequences']:
  print(f"frames {seq['frames']}  front {1000 * seq['front_seconds'] / seq['frames']:.2f} + back {1000 * seq['back_seconds'] / seq['frames']:.2f} ms per frame  " +
      '  '.join(f'{k} {v:.2f}' for k, v in seq['stage_ms_per_frame'].items()))
  if seq.get('detail_ms'):
    print('parts (ms per frame): ' + '  '.join(f"{k} {v / seq['frames']:.3f}" for k, v in seq['detail_ms'].items()))
if os.environ.get('TIME_CLI_JSON'):
    print('TIME_CLI_JSON ' + json.dumps({'loop_seconds': run['loop_seconds'], 'state_write_wait_seconds': run.get('state_write_wait_seconds', 0.0),
                                        'loop_started_at': run['loop_started_at'], 'loop_ended_at': run['loop_ended_at'],
                                        'generator_seconds': run.get('generator_seconds', 0.0),
                                        'sequences': [{k: q[k] for k in ('name', 'frames', 'seconds', 'front_seconds', 'back_seconds', 'stage_ms_per_frame')}
                                                      for q in run['sequences']]}))
