from src.model.config import MultiModalBartConfig
from src.model.model import (LazyLogits, MultiModalBartForConditionalGeneration, MultiModalBartModel)

__all__ = ["MultiModalBartConfig", "MultiModalBartForConditionalGeneration", "MultiModalBartModel", "LazyLogits"]
