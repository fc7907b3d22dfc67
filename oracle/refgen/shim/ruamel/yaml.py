import yaml as _y
class YAML:
    def __init__(self, *a, **k): pass
    def load(self, s): return _y.safe_load(s)
    def dump(self, d, s=None): return _y.safe_dump(d, s)
