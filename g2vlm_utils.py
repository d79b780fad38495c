"""Top-level `g2vlm_utils` - the module name the reference's scripts import (reference inference_recon.py:15,
inference_chat.py:8, the notebook): `from g2vlm_utils import load_model_and_tokenizer, ...` works with this repository's
root on `sys.path` in place of the reference's.  The implementation lives in g2vlm_amd/g2vlm_utils.py; this file only
re-exports it under the reference's name (reference g2vlm_utils.py:31-149)."""
from g2vlm_amd.g2vlm_utils import (LazySafetensors, add_special_tokens, build_model, build_transform,  # noqa: F401
                                   configs_from_dims, load_model_and_tokenizer, pil_img2rgb, process_conversation,
                                   save_ply_visualization)

__all__ = ["load_model_and_tokenizer", "build_transform", "process_conversation", "save_ply_visualization"]
