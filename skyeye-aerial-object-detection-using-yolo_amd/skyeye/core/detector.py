"""Import path advertised by the reference README (README.md:41): ``from skyeye.core.detector import SkyEyeDetector``."""
from .models.detector import (DetectionHead, EnhancedSkyEyeDetector, FeatureNeck, SkyEyeDetector, construct_model,  # noqa: F401
                              load_model, parse_model)
