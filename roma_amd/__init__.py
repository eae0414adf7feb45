"""roma_amd — MI355X-native dense-matching inference path with the reference's Python surface
(`roma_outdoor`, `roma_indoor`, `tiny_roma_v1_outdoor`, `.match/.sample/...`; romatch/__init__.py:2)."""
from . import ops  # noqa: F401
from .model_zoo import roma_indoor, roma_outdoor, tiny_roma_v1_outdoor  # noqa: F401
