"""Helper process that writes sequence-state pickles (vilgod_amd/zero_shot_detector.py, device.async_state_write): reads
length-prefixed pickled requests (path, [FrameState.compact() dicts]) on stdin, builds the reference's frame dicts
(FrameState.serialize: ~18 000 per-detection dicts of numpy objects per 199-frame sequence, ~0.45 s of interpreter time that would
otherwise hold the interpreter lock of the process that launches the GPU kernels), writes the pickle next to its final name and
renames it, answers ('ok', bytes) or ('error', text).  numpy only -- never the GPU runtime.  Ends when stdin closes."""
import os
import pickle
import struct
import sys


def write_state(path, compacts):
    from vilgod_amd.frame_state import FrameState
    data = [FrameState.from_compact(c).serialize for c in compacts]
    tmp = str(path) + '.tmp'
    with open(tmp, 'wb') as fp:
        pickle.dump(data, fp, protocol=pickle.HIGHEST_PROTOCOL)
    n = os.path.getsize(tmp)
    os.replace(tmp, path)                                # readers never see a half-written file
    return n


def main():
    rd, wr = sys.stdin.buffer, sys.stdout.buffer
    while True:
        head = rd.read(8)
        if len(head) < 8:
            return
        (n,) = struct.unpack('<q', head)
        path, compacts = pickle.loads(rd.read(n))
        try:
            ans = ('ok', write_state(path, compacts))
        except Exception as e:          # noqa: BLE001  (reported to the caller, which raises)
            ans = ('error', f'{type(e).__name__}: {e}')
        blob = pickle.dumps(ans, protocol=pickle.HIGHEST_PROTOCOL)
        wr.write(struct.pack('<q', len(blob)))
        wr.write(blob)
        wr.flush()


if __name__ == '__main__':
    main()
