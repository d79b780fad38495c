"""Config + cache classes with the reference's names (reference modeling/g2vlm/qwen2vl.py:50-251,
modeling/qwen2vl/configuration_qwen2_vl.py).  The sub-model classes only carry their config: all
compute lives in g2vlm_amd.engine, driven by modeling.g2vlm.G2VLM."""
import json

from ...engine import KVCache


class _JsonConfig:
    defaults = {}

    def __init__(self, **kw):
        for k, v in {**self.defaults, **kw}.items():
            setattr(self, k, v)

    @classmethod
    def from_json_file(cls, path):
        with open(path) as f:
            return cls(**json.load(f))

    def to_dict(self):
        return dict(self.__dict__)


class Qwen2VLConfig(_JsonConfig):
    """Fields of reference qwen2vl.py:175-234 that the inference path reads."""
    defaults = dict(vocab_size=152064, hidden_size=8192, intermediate_size=29568, num_hidden_layers=80,
                    num_attention_heads=64, num_key_value_heads=8, hidden_act="silu", rms_norm_eps=1e-5,
                    rope_theta=1000000.0, rope_scaling=None, tie_word_embeddings=False, qk_norm=True,
                    layer_module="Qwen2VLDecoderLayer", is_causal=True)

    def __init__(self, **kw):
        super().__init__(**kw)
        if self.qk_norm is False:
            raise AssertionError("qk_norm should be TRUE")          # reference qwen2vl.py:229-230


class Qwen2VLVisionConfig(_JsonConfig):
    defaults = dict(depth=32, embed_dim=1280, hidden_size=3584, hidden_act="quick_gelu", mlp_ratio=4, num_heads=16,
                    in_channels=3, patch_size=14, spatial_merge_size=2, temporal_patch_size=2)


class NaiveCache(KVCache):
    """Same constructor as reference qwen2vl.py:237-251: NaiveCache(num_layers); storage is the
    pre-allocated contiguous cache of g2vlm_amd.engine.KVCache."""

    def __init__(self, num_layers, n_kv_heads=2, device="cuda", capacity=0):
        super().__init__(num_layers, n_kv_heads, device, capacity)


class Qwen2VLModel:
    def __init__(self, config):
        self.config = config


class Qwen2VLForCausalLM:
    def __init__(self, config):
        self.config = config
        self.model = Qwen2VLModel(config)


class Qwen2VisionTransformerPretrainedModel:
    def __init__(self, config):
        self.config = config
