"""detectron2.utils.logger plumbing"""
import logging
import sys


def setup_logger(output=None, distributed_rank=0, *, color=True, name="detectron2", abbrev_name=None):
    logger = logging.getLogger(name)
    logger.setLevel(logging.DEBUG)
    logger.propagate = False
    if distributed_rank == 0 and not logger.handlers:
        h = logging.StreamHandler(stream=sys.stdout)
        h.setLevel(logging.DEBUG)
        h.setFormatter(logging.Formatter("[%(asctime)s %(name)s]: %(message)s", datefmt="%m/%d %H:%M:%S"))
        logger.addHandler(h)
    return logger


def _log_api_usage(identifier):
    pass
