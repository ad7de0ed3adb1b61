"""`cubercnn.vis.logperf`: the tables `Omni3DEvaluationHelper` prints (reference cubercnn/vis/logperf.py, used at
tools/train_net.py:52 and omni3d_evaluation.py:371,517-519) -- tabulate-based, host only."""
import logging

logger = logging.getLogger(__name__)


def _table(rows, headers):
    from tabulate import tabulate
    return tabulate(rows, headers=headers, tablefmt="pipe", floatfmt=".2f", numalign="left")


def print_ap_category_histogram(dataset, results):
    """results: {category: {'AP2D', 'AP3D'}}"""
    rows = [(k, v.get("AP2D", float("nan")), v.get("AP3D", float("nan"))) for k, v in results.items()]
    logger.info("Performance for each of {} categories on {}:\n{}".format(len(results), dataset, _table(rows, ["category", "AP2D", "AP3D"])))


def print_ap_analysis_histogram(results):
    """results: {dataset: {'iters', 'AP2D', 'AP3D', 'AP3D@15', 'AP3D@25', 'AP3D@50', 'AP3D-N', 'AP3D-M', 'AP3D-F'}}"""
    keys = ["iters", "AP2D", "AP3D", "AP3D@15", "AP3D@25", "AP3D@50", "AP3D-N", "AP3D-M", "AP3D-F"]
    rows = [[name] + [v.get(k, float("nan")) for k in keys] for name, v in results.items()]
    logger.info("Per-dataset performance analysis on test set:\n{}".format(_table(rows, ["Dataset"] + keys)))


def print_ap_omni_histogram(results):
    """results: {dataset: {'iters', 'AP2D', 'AP3D'}}"""
    rows = [[name, v.get("iters"), v.get("AP2D", float("nan")), v.get("AP3D", float("nan"))] for name, v in results.items()]
    logger.info("Omni3D performance on test set:\n{}".format(_table(rows, ["Dataset", "iters", "AP2D", "AP3D"])))
