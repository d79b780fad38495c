import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order of the GPU suite (the driver runs `pytest -x`, and in round 2 one failing simulation
# test that happened to sort first hid 293 parity tests).  Evidence first: kernels against the oracle, then the end-to-end
# goldens produced by the reference, then full depth / the BASELINE configs, and only then the harness-heavy tests
# (rank simulations on threads, child processes), so that a failure there can never mask the parity results.
_MODULE_ORDER = ["test_oracle_golden", "test_host_cpu", "test_boundary", "test_kernels_gpu", "test_heads_fp32_gpu", "test_e2e_gpu",
                 "test_dinov3_gpu", "test_full_depth_gpu", "test_configs_gpu", "test_comm_abi_gpu", "test_sharded_gpu"]
_LATE_KEYWORDS = ("sharded", "thread", "two_process", "two_rank", "c4_32_views")


def pytest_collection_modifyitems(session, config, items):
    def key(it):
        mod = os.path.splitext(os.path.basename(str(it.fspath)))[0]
        rank = _MODULE_ORDER.index(mod) if mod in _MODULE_ORDER else len(_MODULE_ORDER) - 2
        late = (mod == "test_sharded_gpu" or any(k in it.name for k in _LATE_KEYWORDS)) and it.get_closest_marker("gpu") is not None
        return (1 if late else 0, rank)
    items.sort(key=key)                                       # stable: file order is kept inside a module


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def write_tiny_checkpoint(path, conf=False):
    """A checkpoint DIRECTORY in the layout load_model_and_tokenizer reads (reference g2vlm_utils.py:31-70): the three config
    JSONs, model.safetensors (TINY dims, seeded synthetic weights = the state-dict key contract) and the files of a
    Qwen2Tokenizer (vocab.json / merges.txt of a tiny byte-level BPE trained here with the `tokenizers` package - no vocab
    is available offline).  Returns (dims, state dict)."""
    import json
    from safetensors.torch import save_file
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    from oracle import dims as D, synth
    dims = D.TINY
    os.makedirs(path, exist_ok=True)
    L, V, Dn = dims["llm"], dims["vit"], dims["dino"]
    json.dump(dict(vocab_size=L["vocab"], hidden_size=L["hidden"], intermediate_size=L["ffn"], num_hidden_layers=L["layers"],
                   num_attention_heads=L["heads"], num_key_value_heads=L["kv_heads"], rms_norm_eps=L["eps"], rope_theta=L["theta"],
                   rope_scaling={"type": "mrope", "mrope_section": [16, 24, 24]}, hidden_act="silu", model_type="qwen2_vl"),
              open(os.path.join(path, "text_config.json"), "w"))
    json.dump(dict(depth=V["depth"], embed_dim=V["embed"], hidden_size=V["out"], mlp_ratio=V["mlp_ratio"], num_heads=V["heads"],
                   hidden_act="quick_gelu", in_channels=3, patch_size=14, spatial_merge_size=2, temporal_patch_size=2),
              open(os.path.join(path, "vit_config.json"), "w"))
    json.dump(dict(hidden_size=Dn["hidden"], num_hidden_layers=Dn["layers"], num_attention_heads=Dn["heads"], patch_size=14,
                   image_size=518, num_register_tokens=4, mlp_ratio=4, hidden_act="gelu", layer_norm_eps=1e-6, use_swiglu_ffn=False),
              open(os.path.join(path, "dino_config.json"), "w"))
    sd = synth.synth_state_dict(dims, seed=21, shapes=synth.param_shapes(dims, conf=True) if conf else None)
    save_file(sd, os.path.join(path, "model.safetensors"))
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    corpus = ["Reconstruct the 3D scene.", "You are a helpful assistant.", "system user assistant your text",
              "How far is the chair from the door? Please answer the question using a single word or phrase.",
              "If the table (red point) is positioned at 2.6 meters, estimate the depth of the clothes (blue point)."] * 4
    tok.train_from_iterator(corpus, trainers.BpeTrainer(vocab_size=400, special_tokens=["<|endoftext|>"],
                                                        initial_alphabet=pre_tokenizers.ByteLevel.alphabet()))
    tok.model.save(path)
    json.dump({"tokenizer_class": "Qwen2Tokenizer", "model_max_length": 32768, "eos_token": "<|endoftext|>",
               "pad_token": "<|endoftext|>", "unk_token": None, "bos_token": None}, open(os.path.join(path, "tokenizer_config.json"), "w"))
    return dims, sd


@pytest.fixture(scope="session")
def tiny_checkpoint(tmp_path_factory):
    path = str(tmp_path_factory.mktemp("g2vlm_ckpt"))
    dims, sd = write_tiny_checkpoint(path)
    return path, dims, sd
