"""Top-level `modeling` - the package name the reference's loader and notebooks import (reference g2vlm_utils.py:5-20,
modeling/__init__.py, modeling/g2vlm/__init__.py:5-18, `from modeling.g2vlm.qwen2vl import NaiveCache` g2vlm.py:22).

With this repository's root on `sys.path` in place of the reference's, these imports resolve to the MI355X engine:

    modeling.g2vlm                        G2VLMConfig, G2VLM, Qwen2VLConfig, Qwen2VLModel, Qwen2VLForCausalLM,
                                          Dinov2WithRegistersConfig, Dinov2WithRegistersModel
    modeling.g2vlm.{g2vlm, qwen2vl, dinov2_model}          (NaiveCache lives in qwen2vl, as in the reference)
    modeling.dinov3[.dinov3_model]        DINOv3ViTConfig, DINOv3ViTModel
    modeling.qwen2                        Qwen2Tokenizer (the reference vendors HF's class; this is HF's, local files only)
    modeling.qwen2vl.modeling_qwen2_vl    Qwen2VisionTransformerPretrainedModel
    modeling.qwen2vl.configuration_qwen2_vl   Qwen2VLVisionConfig

No code is duplicated: every name is the object defined under g2vlm_amd/ (module aliases in sys.modules, so
`modeling.g2vlm.G2VLM is g2vlm_amd.modeling.g2vlm.G2VLM`).  Names of the reference's training stack (losses, FSDP wrappers,
Qwen2VLImageProcessor fetched from the hub by name) are outside the inference hot path and are not provided.
"""
import sys
import types

import g2vlm_amd.modeling.dinov3 as _dinov3
import g2vlm_amd.modeling.dinov3.dinov3_model as _dinov3_model
import g2vlm_amd.modeling.g2vlm as _g2vlm
import g2vlm_amd.modeling.g2vlm.dinov2_model as _dinov2_model
import g2vlm_amd.modeling.g2vlm.g2vlm as _g2vlm_g2vlm
import g2vlm_amd.modeling.g2vlm.qwen2vl as _qwen2vl


def _module(name, doc, **names):
    m = types.ModuleType(name, doc)
    m.__dict__.update(names)
    m.__all__ = sorted(names)
    return m


def _qwen2_getattr(name):
    if name == "Qwen2Tokenizer":                              # imported on first use: transformers is heavy
        from transformers import Qwen2Tokenizer
        return Qwen2Tokenizer
    raise AttributeError(f"module 'modeling.qwen2' has no attribute {name!r} (only Qwen2Tokenizer is on the inference path)")


_qwen2 = _module("modeling.qwen2", "reference modeling/qwen2/__init__.py: the tokenizer the loader constructs (g2vlm_utils.py:57)")
_qwen2.__getattr__ = _qwen2_getattr
_qwen2.__all__ = ["Qwen2Tokenizer"]
_qwen2vl_pkg = _module("modeling.qwen2vl", "reference modeling/qwen2vl: ViT config + model names")
_qwen2vl_pkg.__path__ = []
_qwen2vl_pkg.modeling_qwen2_vl = _module("modeling.qwen2vl.modeling_qwen2_vl", "reference modeling_qwen2_vl.py:987-1072",
                                         Qwen2VisionTransformerPretrainedModel=_qwen2vl.Qwen2VisionTransformerPretrainedModel)
_qwen2vl_pkg.configuration_qwen2_vl = _module("modeling.qwen2vl.configuration_qwen2_vl", "reference configuration_qwen2_vl.py",
                                              Qwen2VLVisionConfig=_qwen2vl.Qwen2VLVisionConfig)

g2vlm, dinov3, qwen2, qwen2vl = _g2vlm, _dinov3, _qwen2, _qwen2vl_pkg
for _name, _mod in {"modeling.g2vlm": _g2vlm, "modeling.g2vlm.g2vlm": _g2vlm_g2vlm, "modeling.g2vlm.qwen2vl": _qwen2vl,
                    "modeling.g2vlm.dinov2_model": _dinov2_model, "modeling.dinov3": _dinov3,
                    "modeling.dinov3.dinov3_model": _dinov3_model, "modeling.qwen2": _qwen2, "modeling.qwen2vl": _qwen2vl_pkg,
                    "modeling.qwen2vl.modeling_qwen2_vl": _qwen2vl_pkg.modeling_qwen2_vl,
                    "modeling.qwen2vl.configuration_qwen2_vl": _qwen2vl_pkg.configuration_qwen2_vl}.items():
    sys.modules.setdefault(_name, _mod)
