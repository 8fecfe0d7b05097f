"""Process-wide cache of Engine objects keyed by batch geometry (the reference-surface functions are stateless)."""
from .engine import Engine

_cache = {}


def get_engine(**cfg):
    key = tuple(sorted(cfg.items()))
    eng = _cache.get(key)
    if eng is None or eng.ctx is None:
        if len(_cache) > 16:
            _cache.clear()
        eng = _cache[key] = Engine(**cfg)
    return eng
