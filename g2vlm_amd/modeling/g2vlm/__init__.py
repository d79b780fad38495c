"""Mirror of reference modeling/g2vlm/__init__.py:5-18 — same export names."""
from .g2vlm import G2VLMConfig, G2VLM
from .qwen2vl import Qwen2VLConfig, Qwen2VLModel, Qwen2VLForCausalLM, NaiveCache, Qwen2VLVisionConfig, Qwen2VisionTransformerPretrainedModel
from .dinov2_model import Dinov2WithRegistersConfig, Dinov2WithRegistersModel

__all__ = ["G2VLMConfig", "G2VLM", "Qwen2VLConfig", "Qwen2VLModel", "Qwen2VLForCausalLM", "NaiveCache",
           "Dinov2WithRegistersConfig", "Dinov2WithRegistersModel", "Qwen2VLVisionConfig",
           "Qwen2VisionTransformerPretrainedModel"]
