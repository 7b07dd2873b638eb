"""Wall time of the drop-in CLI path (stage by stage over a sequence, tools/preprocess_data.py) on one synthetic sequence."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preprocess_data
n = int(os.environ.get('FRAMES', '24'))
root = tempfile.mkdtemp()
ovr = [f'dataset.SYNTHETIC.frames_per_sequence={n}', 'dataset.SYNTHETIC.n_sequences=1', 'end_sequence=0', 'paths.clip_model=/nonexistent',
       f'dataset.DATA_PATH={root}'] + sys.argv[1:]
t0 = time.perf_counter()
preprocess_data.main(['preprocessor=waymo'] + ovr)
t1 = time.perf_counter()
print(f'CLI: {n} frames in {t1 - t0:.2f} s (incl. start-up, synthetic data generation) -> {n / (t1 - t0):.1f} frames/s')
