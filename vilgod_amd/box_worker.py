"""Helper process of box_mode='reference' (vilgod_amd/boxes.py): reads length-prefixed pickled requests
(xy_packed [P,2] float32, seg, zmin, zmax) on stdin, answers the [C,7] boxes on stdout.  Imports numpy / scipy only -- never the
GPU runtime.  Started by boxes.BoxWorkerPool as `python -m vilgod_amd.box_worker`; ends when stdin closes."""
import pickle
import struct
import sys


def main():
    from vilgod_amd.boxes import reference_boxes_packed
    rd, wr = sys.stdin.buffer, sys.stdout.buffer
    while True:
        head = rd.read(8)
        if len(head) < 8:
            return
        (n,) = struct.unpack('<q', head)
        req = pickle.loads(rd.read(n))
        try:
            ans = ('ok', reference_boxes_packed(*req))
        except Exception as e:          # noqa: BLE001  (reported to the caller, which raises)
            ans = ('error', f'{type(e).__name__}: {e}')
        blob = pickle.dumps(ans, protocol=pickle.HIGHEST_PROTOCOL)
        wr.write(struct.pack('<q', len(blob)))
        wr.write(blob)
        wr.flush()


if __name__ == '__main__':
    main()
