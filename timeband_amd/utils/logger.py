"""stdlib logger with the three methods the Timeband classes call (info / warn / debug with a
verbosity `level`), standing in for the reference's structlog Logger (source/utils/logger.py:14-81)."""
import logging


class Logger:
    DETAIL, NORMAL, IMPORT = 0, 1, 2

    def __init__(self, logpath: str = None, verbosity: int = 1, name: str = "timeband_amd"):
        self.verbosity = verbosity
        self.logger = logging.getLogger(name)
        if not self.logger.handlers:
            h = logging.StreamHandler()
            h.setFormatter(logging.Formatter("%(asctime)s [%(levelname)s] %(message)s", "%Y-%m-%d %H:%M:%S"))
            self.logger.addHandler(h)
            if logpath:
                import os
                os.makedirs(logpath, exist_ok=True)
                self.logger.addHandler(logging.FileHandler(os.path.join(logpath, "timeband.log")))
            self.logger.setLevel(logging.INFO)

    def _emit(self, fn, message, level):
        if level >= self.verbosity:
            fn(message)

    def debug(self, message, level: int = 1):
        self._emit(self.logger.debug, message, level)

    def info(self, message, level: int = 1):
        self._emit(self.logger.info, message, level)

    def warn(self, message, level: int = 1):
        self._emit(self.logger.warning, message, level)


class NullLogger:
    def info(self, *a, **k):
        pass

    warn = debug = info
