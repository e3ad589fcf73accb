"""Build-container-only placeholder for `gym` (see ../README.md)."""
from . import spaces  # noqa: F401


class Env:
    def __init__(self, *a, **k):
        pass
