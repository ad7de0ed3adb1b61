"""`cubercnn.vis.logperf`: table printing of evaluation results (tools/train_net.py:52) -- tabulate-based, host only."""


def print_ap_category_histogram(dataset, results):
    from tabulate import tabulate
    rows = [(k, "{:.2f}".format(v.get("AP2D", float("nan"))), "{:.2f}".format(v.get("AP3D", float("nan")))) for k, v in results.items()]
    print("Performance for each of {} categories on {}:\n{}".format(len(results), dataset, tabulate(rows, headers=["category", "AP2D", "AP3D"])))


def print_ap_analysis_histogram(results):
    from tabulate import tabulate
    print(tabulate([(k, v) for k, v in results.items()], headers=["metric", "value"]))
