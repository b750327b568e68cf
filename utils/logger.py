"""Console (+ optional file) logger with the reference's format (utils/logger.py:10-55)."""
import logging
import sys
from pathlib import Path

_FORMAT = '%(asctime)s - %(name)s - %(levelname)s - %(message)s'
_DATEFMT = '%Y-%m-%d %H:%M:%S'


def setup_logger(name='video_diffusion', log_file=None, level=logging.INFO):
    log = logging.getLogger(name)
    log.setLevel(level)
    log.handlers = []
    targets = [logging.StreamHandler(sys.stdout)]
    if log_file:
        Path(log_file).parent.mkdir(parents=True, exist_ok=True)
        targets.append(logging.FileHandler(Path(log_file)))
    for handler in targets:
        handler.setLevel(level)
        handler.setFormatter(logging.Formatter(_FORMAT, datefmt=_DATEFMT))
        log.addHandler(handler)
    return log
