"""MI355X-native drop-in for the Montreal Forced Aligner's alignment hot path."""
import os as _os

# Several batches in flight (one HIP stream each) only overlap if every stream gets a hardware queue of its own; the HIP
# runtime's default is 4 and it reads this variable once, when it initialises.  Set before the first HIP call; an explicit
# setting in the environment wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
