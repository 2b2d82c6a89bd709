"""Pre-processing on the detection path (reference skyeye/core/data): only what the inference callers use."""
from .augmentation import letterbox, letterbox_geometry  # noqa: F401
