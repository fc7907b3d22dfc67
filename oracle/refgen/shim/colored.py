def attr(*a, **k): return ''
def fg(*a, **k): return ''
def stylize(s, *a, **k): return s
