import logging
class ColoredFormatter(logging.Formatter):
    def __init__(self, fmt=None, *a, **k):
        super().__init__()
