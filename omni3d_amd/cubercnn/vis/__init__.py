"""`cubercnn.vis` (reference cubercnn/vis/*: matplotlib / OpenCV drawing of 3D boxes) is cosmetic and outside the MI355X hot path
(SURVEY.md 2.1 #17).  The names the training script touches exist; `visualize_from_instances` renders nothing and says so in
the log string `tools/train_net.py:do_test` prints (:99-106), `logperf` prints the evaluation tables."""
from . import logperf  # noqa: F401


def visualize_from_instances(detections, dataset, dataset_name, min_size_test, output_folder, category_names_official, iteration=""):
    """reference vis/vis.py: draws predictions next to the ground truth for a sample of images and returns a log line with the
    mean 3D error of matched boxes.  Here: no rendering (no matplotlib / OpenCV on the MI355X image)."""
    return "Visualisation skipped for {} ({} predictions, iteration {}): cubercnn.vis renders nothing in this package".format(
        dataset_name, len(detections), iteration)
