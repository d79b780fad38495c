"""Config holder with the reference's names (reference modeling/g2vlm/dinov2_model.py:277-356,
modeling/dinov2_with_registers/configuration_dinov2_with_registers.py:107-158)."""
from .qwen2vl import _JsonConfig


class Dinov2WithRegistersConfig(_JsonConfig):
    defaults = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, mlp_ratio=4, hidden_act="gelu",
                    layer_norm_eps=1e-6, image_size=224, patch_size=16, num_channels=3, qkv_bias=True,
                    layerscale_value=1.0, use_swiglu_ffn=False, num_register_tokens=4)


class Dinov2WithRegistersModel:
    def __init__(self, config):
        self.config = config
