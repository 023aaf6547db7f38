"""vtamiq_amd -- MI355X-native engine for VTAMIQ's ViT patch-pair forward (drop-in model callable)."""
from .spec import ModelSpec, make_spec, VIT_VARIANT_B16, VIT_VARIANT_L16  # noqa: F401
from .model import VTAMIQ  # noqa: F401
from .predict import get_data_tuple, split_per_image, predict, model_forward, PreferenceModule  # noqa: F401
from . import patches, weights, dist, validate  # noqa: F401,E402
