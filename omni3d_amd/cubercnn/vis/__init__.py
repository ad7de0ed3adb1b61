"""`cubercnn.vis` (reference cubercnn/vis/*: matplotlib / OpenCV drawing) is cosmetic and outside the MI355X hot path
(SURVEY.md 2.1 #17).  The names the training script touches exist so that `from cubercnn import vis` and
`import cubercnn.vis.logperf` resolve; calling them says what is missing."""
from . import logperf  # noqa: F401


def visualize_from_instances(*args, **kwargs):
    raise NotImplementedError("visualisation is outside the MI355X hot path (set VIS_PERIOD 0)")
