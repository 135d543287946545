"""pgl.utils.logger (pgl/utils/logger.py): the package logger the examples import as `from pgl.utils.logger import log`."""
import logging

log = logging.getLogger("pgl")
if not log.handlers:
    _console = logging.StreamHandler()
    _console.setFormatter(logging.Formatter(fmt="[%(levelname)s] %(asctime)s [%(filename)12s:%(lineno)5d]:\t%(message)s"))
    log.addHandler(_console)
log.setLevel(logging.DEBUG)
log.propagate = False
