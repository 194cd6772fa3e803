from src.model.config import MultiModalBartConfig
from src.model.model import (LazyLogits, MultiModalBartForConditionalGeneration, MultiModalBartForPreTraining,
                             MultiModalBartModel)

__all__ = ["MultiModalBartConfig", "MultiModalBartForConditionalGeneration", "MultiModalBartForPreTraining",
           "MultiModalBartModel", "LazyLogits"]
