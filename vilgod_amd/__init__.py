"""MI355X-native pseudo-label generation path for ViLGOD (see README.md / DESIGN.md)."""
import os as _os

# Frames in flight run on one HIP stream each (six worker streams + the ground stream per PseudoLabelPipeline).  The HIP runtime
# multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a queue serialise: measured on
# one MI355X, 2 queues cost 17 % of the frames/s, and a second pipeline object in the same process (whose streams land on queues
# already in use) ran 10 % slower than the first until the limit was raised.  The variable is read when the runtime initialises,
# i.e. at the first GPU call of the process: importing this package early enough sets a default the user can still override.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
