"""Loader / glue with the reference's names and return conventions (reference g2vlm_utils.py:31-149).

    model, tokenizer, new_token_ids, vit_image_transform, dino_transform = load_model_and_tokenizer(model_path)

Differences from the reference, all inside the boundary contract of SURVEY.md §8b: accepts a path
string OR the argparse Namespace the reference scripts pass by mistake (H6); takes `device=`;
never reaches the hub (configs, tokenizer files and `model.safetensors` are read from the local
directory only); writes PLY without open3d.
"""
import os

import numpy as np
import torch

from . import host
from .host import QwenVL2ImageTransform, DinoImageNormalizeTransform
from .modeling.g2vlm import (G2VLM, G2VLMConfig, Qwen2VLConfig, Qwen2VLForCausalLM, Dinov2WithRegistersConfig,
                             Dinov2WithRegistersModel, Qwen2VLVisionConfig, Qwen2VisionTransformerPretrainedModel)


def add_special_tokens(tokenizer):
    """reference data/data_utils.py:278-313"""
    specials = []
    for v in tokenizer.special_tokens_map.values():
        specials += [v] if isinstance(v, str) else list(v)
    new = [t for t in ("<|im_start|>", "<|im_end|>", "<|vision_start|>", "<|vision_end|>") if t not in specials]
    n_new = tokenizer.add_tokens(new)
    ids = dict(bos_token_id=tokenizer.convert_tokens_to_ids("<|im_start|>"), eos_token_id=tokenizer.convert_tokens_to_ids("<|im_end|>"),
               start_of_image=tokenizer.convert_tokens_to_ids("<|vision_start|>"),
               end_of_image=tokenizer.convert_tokens_to_ids("<|vision_end|>"))
    return tokenizer, ids, n_new


def pil_img2rgb(image):
    """reference data/data_utils.py:254-263"""
    from PIL import Image
    if image.mode == "RGBA" or image.info.get("transparency", None) is not None:
        image = image.convert("RGBA")
        white = Image.new(mode="RGB", size=image.size, color=(255, 255, 255))
        white.paste(image, mask=image.split()[3])
        return white
    return image.convert("RGB")


def build_model(llm_config, vit_config, dino_config, state_dict, device="cuda"):
    llm_config.qk_norm = True
    llm_config.tie_word_embeddings = False
    llm_config.layer_module = "Qwen2VLMoTDecoderLayer"
    if vit_config is not None:
        vit_config.patch_size = 14
    from .modeling.dinov3 import DINOv3ViTConfig, DINOv3ViTModel
    v3 = isinstance(dino_config, DINOv3ViTConfig)          # the use_dinov3 variant (reference g2vlm.py:86, 134): config objects only,
    config = G2VLMConfig(visual_und=vit_config is not None, visual_recon=True, llm_config=llm_config, vit_config=vit_config,
                         dino_config=dino_config, vit_max_num_patch_per_side=36, use_dinov3=v3)   # the reference's loader never sets it
    model = G2VLM(Qwen2VLForCausalLM(llm_config), Qwen2VisionTransformerPretrainedModel(vit_config) if vit_config else None,
                  DINOv3ViTModel(dino_config) if v3 else Dinov2WithRegistersModel(dino_config), config)
    model.load_state_dict(state_dict, strict=False)
    return model.to(device).eval()


def configs_from_dims(dims):
    L, D, V = dims["llm"], dims["dino"], dims["vit"]
    llm = Qwen2VLConfig(vocab_size=L["vocab"], hidden_size=L["hidden"], intermediate_size=L["ffn"], num_hidden_layers=L["layers"],
                        num_attention_heads=L["heads"], num_key_value_heads=L["kv_heads"], rms_norm_eps=L["eps"],
                        rope_theta=L["theta"], rope_scaling={"type": "mrope", "mrope_section": [16, 24, 24]})
    vit = Qwen2VLVisionConfig(depth=V["depth"], embed_dim=V["embed"], hidden_size=V["out"], mlp_ratio=V["mlp_ratio"],
                              num_heads=V["heads"]) if V["depth"] > 0 else None
    if D.get("v3"):
        from .modeling.dinov3 import DINOv3ViTConfig
        dino = DINOv3ViTConfig(**D["v3"])
    else:
        dino = Dinov2WithRegistersConfig(hidden_size=D["hidden"], num_hidden_layers=D["layers"], num_attention_heads=D["heads"],
                                         patch_size=14, image_size=518)
    return llm, vit, dino


class LazySafetensors:
    """Mapping view of a .safetensors file that reads one tensor per access (fp32), so the 4.5 B-parameter checkpoint is
    never resident on the host as a whole: g2vlm_amd.weights.Weights pulls each tensor, rounds / lays it out and uploads it
    (the reference loads the full dict with safetensors.torch.load_file, g2vlm_utils.py:63-66)."""

    def __init__(self, path):
        from safetensors import safe_open
        self.f = safe_open(path, framework="pt", device="cpu")
        self._keys = set(self.f.keys())

    def __contains__(self, k):
        return k in self._keys

    def __getitem__(self, k):
        if k not in self._keys:
            raise KeyError(f"{k!r} is not in the checkpoint (state-dict key contract: g2vlm_amd/synthetic.py::param_shapes)")
        return self.f.get_tensor(k).float()

    def keys(self):
        return self._keys


def load_model_and_tokenizer(model_path, device=None):
    if not isinstance(model_path, (str, os.PathLike)):                      # argparse Namespace (reference bug H6)
        model_path = getattr(model_path, "model_path", None) or getattr(model_path, "model-path", None)
        if model_path is None:
            raise TypeError("load_model_and_tokenizer takes a checkpoint directory or a Namespace with .model_path")
    if not os.path.isdir(model_path):
        raise FileNotFoundError(f"{model_path!r} is not a local directory; this loader never fetches from the hub. "
                                "Download InternRobotics/G2VLM-2B-MoT first and pass its path.")
    device = device or "cuda"
    llm_config = Qwen2VLConfig.from_json_file(os.path.join(model_path, "text_config.json"))
    vit_config = Qwen2VLVisionConfig.from_json_file(os.path.join(model_path, "vit_config.json"))
    dino_config = Dinov2WithRegistersConfig.from_json_file(os.path.join(model_path, "dino_config.json"))
    model = build_model(llm_config, vit_config, dino_config, LazySafetensors(os.path.join(model_path, "model.safetensors")), device)
    from transformers import AutoTokenizer
    tokenizer = AutoTokenizer.from_pretrained(model_path, local_files_only=True)
    tokenizer, new_token_ids, _ = add_special_tokens(tokenizer)
    vit_image_transform = QwenVL2ImageTransform(768, 768, 14)
    dino_transform = DinoImageNormalizeTransform(target_size=518)
    return model, tokenizer, new_token_ids, vit_image_transform, dino_transform


def build_transform(pixel=224):
    return QwenVL2ImageTransform(pixel, pixel, 14)


def process_conversation(images, conversation):
    return [pil_img2rgb(image) for image in images], conversation


def save_ply_visualization(pred_dict, save_path, init_conf_threshold=20.0, filter_nan=True, verbose=True):
    """reference g2vlm_utils.py:84-149: PLY of points[0] coloured by images[0], NaN/inf filtered, plus
    results/input_images.png (8-per-row grid).  The reference's identity-sized bilinear resample of
    the points is skipped (H8: it is a no-op when sizes match)."""
    images = pred_dict["images"][0].detach().float().cpu()
    pts = pred_dict["points"][0].detach().float().cpu()
    n, _, h, w = images.shape
    if pts.shape[1:3] != (h, w):
        pts = torch.nn.functional.interpolate(pts.permute(0, 3, 1, 2), (h, w), mode="bilinear", align_corners=False,
                                              antialias=True).permute(0, 2, 3, 1)
    os.makedirs(os.path.dirname(save_path) or ".", exist_ok=True)
    try:
        from PIL import Image
        nrow, pad = 8, 2
        cols, rows = min(n, nrow), (n + nrow - 1) // nrow
        grid = torch.zeros((3, rows * (h + pad) + pad, cols * (w + pad) + pad))
        for i in range(n):
            r, c = divmod(i, nrow)
            grid[:, pad + r * (h + pad): pad + r * (h + pad) + h, pad + c * (w + pad): pad + c * (w + pad) + w] = images[i]
        os.makedirs("results", exist_ok=True)
        Image.fromarray((grid.clamp(0, 1) * 255 + 0.5).to(torch.uint8).permute(1, 2, 0).numpy()).save("results/input_images.png")
    except Exception as e:                                                    # noqa: BLE001  (visual by-product only)
        if verbose:
            print("input_images.png not written:", e)
    p = pts.numpy().reshape(-1, 3)
    c = images.permute(0, 2, 3, 1).numpy().reshape(-1, 3)
    if filter_nan:
        ok = ~(np.isnan(p).any(1) | np.isinf(p).any(1))
        p, c = p[ok], c[ok]
    host.write_ply_binary(save_path, p, c)
