"""Debug aid: run this repo's CLI on an integration golden's overrides (parity mode) and keep the pickles under gpurun_out/
so that they can be diffed against tests/golden/integration_<which>.pkl.gz on the CPU box.
    python tools/dev/integ_dump.py default [extra overrides ...]"""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

if __name__ == '__main__':
    import pickle
    import test_integration as ti
    which = sys.argv[1]
    g = ti._golden(which)
    root = tempfile.mkdtemp()
    res, idx, state = ti._run_cli(g, root, extra=sys.argv[2:])
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', f'integ_{which}.pkl'), 'wb') as f:
        pickle.dump(dict(results=res, indices=idx, state=state), f)
    print(ti._report(ti._compare(g, res, idx, state))[:3000])
    shutil.rmtree(root, ignore_errors=True)
