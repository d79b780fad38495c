"""Import the upstream reference (read-only at /root/reference) on CPU through shims.

TEST INFRASTRUCTURE ONLY.  Used by ``oracle/gen_golden.py`` in the build container to
generate the golden vectors committed under ``tests/golden/``.  Nothing here ships, and
nothing here runs on the GPU box (``/root/reference`` does not exist there).

Recipe follows SURVEY.md §8(c):
  * transformers is imported first, then fakes are registered for the wheels that are
    absent in this image (flash_attn, torchvision, cv2, easydict);
  * ``modeling`` / ``modeling.g2vlm`` are pre-registered as empty packages so the eager
    ``modeling/__init__.py`` (pulls torchvision/open3d paths) never runs;
  * transformers 4.49 -> 5.x API drift is patched (three removed helpers);
  * ``torch.autocast('cuda', ...)`` is redirected to ``'cpu'`` so the bf16 regions and the
    ``enabled=False`` fp32 islands behave on CPU as they do on the GPU reference.

The fake ``flash_attn_varlen_func`` is the convention the oracle pins for third-party
flash-attn 2.7.4 (source not under /root/reference => "parity unpinned" for its own
numerics): fp32 softmax(QK^T/sqrt(d))V per ``cu_seqlens`` window, GQA by head repeat,
bottom-right aligned causal mask, rows outside every window = 0 (App. D-H1).
"""
import importlib.machinery
import math
import sys
import types

REF_ROOT = "/root/reference"


def _fake_flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                                 dropout_p=0.0, softmax_scale=None, causal=False, **kw):
    import torch
    # the call sites sit inside autocast(bf16) regions; the real kernel keeps S and the softmax in fp32,
    # so autocast must not touch the fp32 matmuls below
    with _ORIG_AUTOCAST[0]("cpu", enabled=False):
        return _fake_flash_impl(q, k, v, cu_seqlens_q, cu_seqlens_k, softmax_scale, causal)


_ORIG_AUTOCAST = []


def _fake_flash_impl(q, k, v, cu_seqlens_q, cu_seqlens_k, softmax_scale, causal):
    import torch
    Hq, Hk, D = q.shape[1], k.shape[1], q.shape[2]
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(D)
    out = torch.zeros_like(q)
    cq = [int(x) for x in cu_seqlens_q.tolist()]
    ck = [int(x) for x in cu_seqlens_k.tolist()]
    rep = Hq // Hk
    for i in range(len(cq) - 1):
        qs, qe, ks, ke = cq[i], cq[i + 1], ck[i], ck[i + 1]
        if qe <= qs:
            continue
        qi = q[qs:qe].float().transpose(0, 1)                                   # [Hq, lq, D]
        ki = k[ks:ke].float().transpose(0, 1).repeat_interleave(rep, dim=0)     # [Hq, lk, D]
        vi = v[ks:ke].float().transpose(0, 1).repeat_interleave(rep, dim=0)
        s = torch.matmul(qi, ki.transpose(1, 2)) * scale
        if causal:
            lq, lk = qe - qs, ke - ks
            row = torch.arange(lq).view(-1, 1)
            col = torch.arange(lk).view(1, -1)
            s = s.masked_fill(col > row + (lk - lq), float("-inf"))
        p = torch.softmax(s, dim=-1)
        o = torch.matmul(p, vi)                                                  # [Hq, lq, D]
        out[qs:qe] = o.transpose(0, 1).to(q.dtype)
    return out


def install():
    """Register shims; idempotent.  Returns the dict of reference modules that were imported."""
    if getattr(install, "_done", None) is not None:
        return install._done
    sys.dont_write_bytecode = True
    import torch
    import transformers  # noqa: F401  (must precede the stubs: its availability probes crash on spec-less stubs)
    import transformers.pytorch_utils as _tpu
    import transformers.utils.backbone_utils as _tbu
    import transformers.modeling_utils  # noqa: F401
    import transformers.image_utils  # noqa: F401
    import transformers.image_transforms  # noqa: F401
    import transformers.image_processing_utils  # noqa: F401
    from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS

    for p in (REF_ROOT + "/modeling", REF_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)

    def _mod(name, **attrs):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        for k_, v_ in attrs.items():
            setattr(m, k_, v_)
        sys.modules[name] = m
        return m

    # empty packages so `modeling/__init__.py` (eager, pulls everything) never executes
    pk = _mod("modeling"); pk.__path__ = [REF_ROOT + "/modeling"]
    pg = _mod("modeling.g2vlm"); pg.__path__ = [REF_ROOT + "/modeling/g2vlm"]

    _mod("flash_attn", flash_attn_varlen_func=_fake_flash_attn_varlen_func)

    # torchvision stub: only Normalize / ToTensor / InterpolationMode are touched on the hot path
    class _Normalize:
        def __init__(self, mean, std, inplace=False):
            self.mean, self.std = mean, std

        def __call__(self, x):
            m = torch.tensor(self.mean, dtype=x.dtype).view(-1, 1, 1)
            s = torch.tensor(self.std, dtype=x.dtype).view(-1, 1, 1)
            return (x - m) / s

    class _ToTensor:
        def __call__(self, pic):
            import numpy as np
            a = np.asarray(pic, dtype=np.uint8)
            if a.ndim == 2:
                a = a[:, :, None]
            return torch.from_numpy(a.copy()).permute(2, 0, 1).float().div(255)

    class _Interp:
        BICUBIC = "bicubic"; BILINEAR = "bilinear"; NEAREST = "nearest"; LANCZOS = "lanczos"

    tvt = _mod("torchvision.transforms", Normalize=_Normalize, ToTensor=_ToTensor, InterpolationMode=_Interp)
    tvf = _mod("torchvision.transforms.functional", InterpolationMode=_Interp)
    tvu = _mod("torchvision.utils")
    tv = _mod("torchvision", transforms=tvt, utils=tvu)
    tv.__path__ = []
    tvt.functional = tvf
    _mod("cv2")

    class EasyDict(dict):
        def __getattr__(self, k_):
            try:
                return self[k_]
            except KeyError as e:
                raise AttributeError(k_) from e
        __setattr__ = dict.__setitem__
    _mod("easydict", EasyDict=EasyDict)

    # transformers 4.49 -> 5.x drift
    def _gone(*a, **k_):
        raise NotImplementedError("pruning helper removed in transformers 5; not on the hot path")
    if not hasattr(_tpu, "find_pruneable_heads_and_indices"):
        _tpu.find_pruneable_heads_and_indices = _gone
    if not hasattr(_tpu, "prune_linear_layer"):
        _tpu.prune_linear_layer = _gone
    if not hasattr(_tbu, "get_aligned_output_features_output_indices"):
        def _aligned(out_features=None, out_indices=None, stage_names=None):
            return [stage_names[-1]], [len(stage_names) - 1]
        _tbu.get_aligned_output_features_output_indices = _aligned
    if "default" not in ROPE_INIT_FUNCTIONS:
        def _default_rope(config, device=None, seq_len=None, **kw):
            base = config.rope_theta if hasattr(config, "rope_theta") else config.rope_parameters["rope_theta"]
            dim = config.hidden_size // config.num_attention_heads
            inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.int64).float() / dim))
            return inv_freq, 1.0
        ROPE_INIT_FUNCTIONS["default"] = _default_rope
    import transformers.image_utils as _tiu
    if not hasattr(_tiu, "VideoInput"):
        _tiu.VideoInput = _tiu.ImageInput
    if not hasattr(_tiu, "make_batched_videos"):
        _tiu.make_batched_videos = lambda v: v                           # video path unused

    # autocast('cuda') -> autocast('cpu')
    _orig_autocast = torch.amp.autocast_mode.autocast
    _ORIG_AUTOCAST.append(_orig_autocast)

    class _CpuAutocast(_orig_autocast):
        def __init__(self, device_type="cpu", *a, **k_):
            if device_type == "cuda":
                device_type = "cpu"
            super().__init__(device_type, *a, **k_)

    torch.amp.autocast = _CpuAutocast
    torch.autocast = _CpuAutocast
    torch.amp.autocast_mode.autocast = _CpuAutocast

    class _CudaAmpAutocast(_CpuAutocast):
        def __init__(self, enabled=True, dtype=torch.float16, cache_enabled=True):
            super().__init__("cpu", enabled=enabled, dtype=dtype, cache_enabled=cache_enabled)
    torch.cuda.amp.autocast = _CudaAmpAutocast

    import modeling.g2vlm.g2vlm as ref_g2vlm
    import modeling.g2vlm.qwen2vl as ref_qwen2vl
    import modeling.g2vlm.dinov2_model as ref_dino
    import modeling.qwen2vl.modeling_qwen2_vl as ref_qwen2vl_hf
    ref_qwen2vl_hf.flash_attn_varlen_func = _fake_flash_attn_varlen_func   # guarded import there resolves to None
    from modeling.qwen2vl.configuration_qwen2_vl import Qwen2VLVisionConfig
    from modeling.dinov2_with_registers.configuration_dinov2_with_registers import Dinov2WithRegistersConfig

    install._done = dict(
        g2vlm=ref_g2vlm, qwen2vl=ref_qwen2vl, dino=ref_dino, qwen2vl_hf=ref_qwen2vl_hf,
        Qwen2VLVisionConfig=Qwen2VLVisionConfig, Dinov2WithRegistersConfig=Dinov2WithRegistersConfig,
    )
    return install._done


def build_reference_model(dims, seed=0):
    """Construct the reference G2VLM on CPU from config objects (no hub access), seeded.

    ``dims`` is the dict used everywhere in this repo (see oracle/dims.py).
    """
    import torch
    R = install()
    g, qv = R["g2vlm"], R["qwen2vl"]
    llm = dims["llm"]
    llm_config = qv.Qwen2VLConfig(
        vocab_size=llm["vocab"], hidden_size=llm["hidden"], intermediate_size=llm["ffn"],
        num_hidden_layers=llm["layers"], num_attention_heads=llm["heads"], num_key_value_heads=llm["kv_heads"],
        rms_norm_eps=llm["eps"], rope_theta=llm["theta"],
        rope_scaling={"type": "mrope", "mrope_section": [16, 24, 24]},
        layer_module="Qwen2VLMoTDecoderLayer", qk_norm=True, tie_word_embeddings=False, pad_token_id=None,
    )
    if not hasattr(llm_config, "rope_theta"):
        llm_config.rope_theta = llm["theta"]
    if getattr(llm_config, "rope_scaling", None) is None:
        llm_config.rope_scaling = {"type": "mrope", "mrope_section": [16, 24, 24]}
    vit = dims["vit"]
    vit_config = R["Qwen2VLVisionConfig"](
        depth=vit["depth"], embed_dim=vit["embed"], hidden_size=vit["out"], hidden_act="quick_gelu",
        mlp_ratio=vit["mlp_ratio"], num_heads=vit["heads"], in_channels=3, patch_size=14,
        spatial_merge_size=2, temporal_patch_size=2,
    )
    vit_config._attn_implementation = "sdpa"
    dino = dims["dino"]
    v3 = dino.get("v3")
    if v3:                                                                # the use_dinov3 variant (g2vlm.py:86, 134)
        import importlib
        m3 = importlib.import_module("modeling.dinov3.dinov3_model")
        dino_config = m3.DINOv3ViTConfig(**v3, image_size=224)
    else:
        dino_config = R["Dinov2WithRegistersConfig"](
            hidden_size=dino["hidden"], num_hidden_layers=dino["layers"], num_attention_heads=dino["heads"],
            mlp_ratio=4, image_size=518, patch_size=14, num_register_tokens=4, layerscale_value=1.0,
        )
    cfg = g.G2VLMConfig(visual_und=True, visual_recon=True, llm_config=llm_config, vit_config=vit_config,
                        dino_config=dino_config, vit_max_num_patch_per_side=36, use_dinov3=bool(v3))
    torch.manual_seed(seed)
    lm = qv.Qwen2VLForCausalLM(llm_config)
    vit_model = R["qwen2vl_hf"].Qwen2VisionTransformerPretrainedModel(vit_config)
    dino_model = m3.DINOv3ViTModel(dino_config) if v3 else R["dino"].Dinov2WithRegistersModel(dino_config)
    model = g.G2VLM(lm, vit_model, dino_model, cfg).eval()
    return model
