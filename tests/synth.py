"""The synthetic workload generator lives at the repo root (synth_workload.py) so that bench.py does not depend on the
test package; tests keep importing it under this name."""
from synth_workload import *  # noqa: F401,F403
from synth_workload import _topology, _log_probs, _accumulate, _gauss  # noqa: F401
