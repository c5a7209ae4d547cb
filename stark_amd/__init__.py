"""stark_amd — MI355X-native engine for STARK's per-Newton-step hot path (libmistark.so + host mirrors)."""
from . import capi  # noqa: F401
from .engine import Engine, EngineError  # noqa: F401
from . import sim  # noqa: F401,E402
