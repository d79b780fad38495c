"""Model dimensions used by the oracle, the fixture generator and the tests.

TEST INFRASTRUCTURE (oracle/).  REAL follows SURVEY.md App. A (inferred G2VLM-2B-MoT dims:
Qwen2-VL-2B LLM with a duplicated geometric expert, DINOv2-L/14+4 registers, Qwen2-VL ViT,
Pi3 decoders).  Decoder/head widths are hard-coded by the reference relative to the LLM
hidden size (modeling/g2vlm/g2vlm.py:162-203): dec dim = hidden, 16 heads, out 1024/512,
heads 1024->588, camera 512.  The LLM head_dim must be 128 (mrope_section [16,24,24]*2 is
hard-coded, modeling/qwen2vl/modeling_qwen2_vl.py:561-566).
"""
import copy

REAL = {
    "llm": dict(hidden=1536, layers=28, heads=12, kv_heads=2, ffn=8960, vocab=151936, eps=1e-6, theta=1e6),
    "dino": dict(hidden=1024, layers=24, heads=16),
    "vit": dict(embed=1280, depth=32, heads=16, mlp_ratio=4, out=1536),
    "dec": dict(depth=5, heads=16),
}

# smallest legal instance: head_dim 128 in the LLM, 16 decoder heads
TINY = {
    "llm": dict(hidden=256, layers=2, heads=2, kv_heads=1, ffn=512, vocab=512, eps=1e-6, theta=1e6),
    "dino": dict(hidden=128, layers=2, heads=2),
    "vit": dict(embed=160, depth=2, heads=2, mlp_ratio=4, out=256),
    "dec": dict(depth=5, heads=16),
}


def reduced(llm_layers=2, dino_layers=2, vit_depth=2, vocab=None):
    """Real widths, reduced depth (the fixture shape SURVEY §8c prescribes)."""
    d = copy.deepcopy(REAL)
    d["llm"]["layers"] = llm_layers
    d["dino"]["layers"] = dino_layers
    d["vit"]["depth"] = vit_depth
    if vocab is not None:
        d["llm"]["vocab"] = vocab
    return d
