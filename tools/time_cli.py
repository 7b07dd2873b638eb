#!/usr/bin/env python3
"""Wall time of the entry point (tools/preprocess_data.py, default 9-stage list) on synthetic sequences WITHOUT a Python profiler
(tools/profile_cli.py runs under cProfile, which slows the host-heavy stages disproportionately).
    [SEQUENCES=1] python tools/time_cli.py [frames=199] [points=150000] [key=value ...]"""
import logging, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import preprocess_data  # noqa: E402
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 199
points = int(sys.argv[2]) if len(sys.argv) > 2 else 150_000
nseq = int(os.environ.get('SEQUENCES', '1'))
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
    import json
    print('TIME_CLI_JSON ' + json.dumps({'loop_seconds': run['loop_seconds'], 'state_write_wait_seconds': run.get('state_write_wait_seconds', 0.0),
                                        'sequences': [{k: q[k] for k in ('name', 'frames', 'seconds', 'front_seconds', 'back_seconds', 'stage_ms_per_frame')}
                                                      for q in run['sequences']]}))
