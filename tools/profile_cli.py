#!/usr/bin/env python3
"""Host-side profile of the entry point: tools/preprocess_data.py (default 9-stage list) on one synthetic sequence under cProfile.

    python tools/profile_cli.py [frames=60] [points=150000]

Prints the per-stage ms per frame the dispatcher measured and the 45 functions with the largest cumulative host time.
Development aid (where the CLI's host overhead goes); not part of the product path."""
import cProfile
import io
import logging
import os
import pstats
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import preprocess_data  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
points = int(sys.argv[2]) if len(sys.argv) > 2 else 150_000
extra = sys.argv[3:]
with tempfile.TemporaryDirectory() as root:
    ovr = ['preprocessor=waymo', f'dataset.DATA_PATH={root}', f'dataset.SYNTHETIC.frames_per_sequence={frames}',
           f'dataset.SYNTHETIC.points_per_frame={points}', 'dataset.SYNTHETIC.n_sequences=1', 'end_sequence=0',
           f'device.max_points={points + 1024}', 'paths.clip_model=/nonexistent'] + extra
    logging.disable(logging.INFO)
    pr = cProfile.Profile()
    pr.enable()
    preprocess_data.main(ovr)
    pr.disable()
seq = preprocess_data.LAST_RUN['sequences'][0]
print(f"frames {seq['frames']}  {1000 * seq['seconds'] / seq['frames']:.2f} ms per frame  ({seq['frames'] / seq['seconds']:.2f} frames/s)")
for k, v in seq['stage_ms_per_frame'].items():
    print(f'  {k:32s} {v:8.2f} ms per frame')
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue()[:9000])
