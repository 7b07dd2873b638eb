"""Helper process of box_mode='reference' (vilgod_amd/boxes.py): reads length-prefixed pickled requests on stdin, answers on stdout.
  (xy_packed [P,2] float32, seg, zmin, zmax)                                   -> [C,7] static boxes (boxes.reference_boxes_packed)
  ('moving_boxes', xyz_packed [P,3] float32, seg, directions, to_ego, centers3) -> [n,7] boxes of one moving track (tracking.moving_boxes)
Imports numpy / scipy only -- never the GPU runtime.  Started by boxes.BoxWorkerPool as `python -m vilgod_amd.box_worker`; ends when
stdin closes."""
import pickle
import struct
import sys


def main():
    from vilgod_amd.boxes import reference_boxes_packed
    from vilgod_amd.tracking import moving_boxes_packed
    named = {'moving_boxes': moving_boxes_packed}
    rd, wr = sys.stdin.buffer, sys.stdout.buffer
    while True:
        head = rd.read(8)
        if len(head) < 8:
            return
        (n,) = struct.unpack('<q', head)
        req = pickle.loads(rd.read(n))
        try:
            ans = ('ok', named[req[0]](*req[1:]) if isinstance(req[0], str) else reference_boxes_packed(*req))
        except Exception as e:          # noqa: BLE001  (reported to the caller, which raises)
            ans = ('error', f'{type(e).__name__}: {e}')
        blob = pickle.dumps(ans, protocol=pickle.HIGHEST_PROTOCOL)
        wr.write(struct.pack('<q', len(blob)))
        wr.write(blob)
        wr.flush()


if __name__ == '__main__':
    main()
