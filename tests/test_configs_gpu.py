"""GPU: BASELINE configs C4 and C5 at their real sizes (`configs_untested`).

C4 = one 32-view 518x518 scene, 4 views per GPU on 8 GPUs (reference: no multi-GPU inference exists, SURVEY §8e is the
design).  One MI355X holds the whole scene, so the 8-rank view-sharded prefill (g2vlm_amd/sharded.py; ranks as threads of
one process, the collectives as rendezvous copies) is compared with the unsharded engine on the same 32 views at real
widths - reduced depth for the comparison, full depth for the property run.
C5 = per GPU 2 scenes x (8-view reconstruction + chat over the same 8 views: 8 ViT images, a question, 256 greedy tokens,
the two scenes decoded together); here with 32 decode steps, the batched ids checked against each scene's own batch-1 ids.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dims as D, synth  # noqa: E402  (inputs / fake tokenizer only)

N4, HW = 32, 518
P = (HW // 14) ** 2


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def point_q(a, b):
    a, b = a.double().reshape(-1, 3), b.double().reshape(-1, 3)
    e = (a - b).norm(dim=1) / (b.norm(dim=1) + 1e-30)
    return float(e.quantile(0.5)), float(e.quantile(0.9))


@pytest.fixture(scope="module")
def full_model():
    from g2vlm_amd.g2vlm_utils import build_model, configs_from_dims
    from g2vlm_amd.synthetic import REAL_DIMS, SyntheticStateDict
    dev = torch.device("cuda", 0)
    return build_model(*configs_from_dims(REAL_DIMS), SyntheticStateDict(REAL_DIMS, dev, seed=0), dev), REAL_DIMS


def test_c4_32_views_sharded_over_8_ranks_matches_unsharded():
    """Real widths, 2 DINO + 2 MoT layers (decoders keep their 5 blocks), 32 views of 518x518: Lq 43 872, Lk 43 880; 8
    simulated ranks of 4 views (5 484 query rows each) exchange their K/V blocks per layer, the DINO boundary rows (hazard
    H1: 5 N = 160 rows move between neighbours) and view 0's context.  The gathered KV cache and every output must equal
    the unsharded engine's up to fp32 summation order (different attention schedules, bf16 P)."""
    from g2vlm_amd.g2vlm_utils import build_model, configs_from_dims
    from g2vlm_amd.modeling.g2vlm import NaiveCache
    from g2vlm_amd.sharded import run_thread_sim
    dims = D.reduced(vocab=2048)
    model = build_model(*configs_from_dims(dims), synth.synth_state_dict(dims, seed=31), "cuda")
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    g = torch.Generator(); g.manual_seed(31)
    imgs = torch.rand((N4, 3, HW, HW), generator=g).cuda()
    ref = model.recon(tok, tok.new_token_ids, None, imgs)
    past = NaiveCache(dims["llm"]["layers"], dims["llm"]["kv_heads"], "cuda")
    gi, nl, nr = model.prepare_prompts_addbos([0], [0], ["Reconstruct the 3D scene."], tok, tok.new_token_ids)
    past = model.forward_cache_update_text(past, **gi)
    gi, nl, nr = model.prepare_dino_images_pi3(nl, nr, imgs, None, tok.new_token_ids)
    past, _ = model.forward_cache_update_dino(past, **gi)
    assert past.length == nl[0]
    res = run_thread_sim(model, 8, tok, tok.new_token_ids, imgs, gather=False)
    last = dims["llm"]["layers"] - 1
    for r in range(8):
        lo, hi = 4 * r, 4 * r + 4
        assert res[r]["view_range"] == (lo, hi)
        pk = res[r]["past_key_values"]
        assert pk.length == past.length
        # a rank's 4 DINO windows are cut differently over the workgroups than the scene's 32 (items split, partials merged in
        # another order, P rounded against another running maximum): bf16-level noise from the first layer on
        assert rel(pk.key_cache[0], past.key_cache[0]) < 8e-3 and rel(pk.value_cache[last], past.value_cache[last]) < 1e-2, r
        for k in ("points", "local_points", "global_points"):
            q50, q90 = point_q(res[r][k], ref[k][:, lo:hi])
            assert q50 < 2e-2 and q90 < 5e-2, (r, k, q50, q90)
        assert rel(res[r]["camera_poses"], ref["camera_poses"][:, lo:hi]) < 5e-3, r
        assert torch.equal(res[r]["images"], ref["images"][:, lo:hi])


def test_c4_full_depth_properties(full_model):
    """The 32-view scene at full width and depth on one GPU (24 DINO + 28 MoT + 15 decoder blocks): shapes / dtypes of the
    reference's dict, everything finite, world points = pose . [local, 1] (g2vlm.py:1226), rotations orthonormal,
    cache length T0 + N (P + 2)."""
    model, dims = full_model
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    g = torch.Generator(); g.manual_seed(32)
    imgs = torch.rand((N4, 3, HW, HW), generator=g).cuda()
    pred = model.recon(tok, tok.new_token_ids, None, imgs)
    for k, shp in (("points", (1, N4, HW, HW, 3)), ("local_points", (1, N4, HW, HW, 3)), ("global_points", (1, N4, HW, HW, 3)),
                   ("camera_poses", (1, N4, 4, 4)), ("images", (1, N4, 3, HW, HW))):
        assert pred[k].shape == shp and pred[k].dtype == torch.float32 and torch.isfinite(pred[k]).all(), k
    local, poses = pred["local_points"].double(), pred["camera_poses"].double()
    R, t = poses[0, :, :3, :3], poses[0, :, :3, 3]
    eye = torch.eye(3, dtype=torch.float64, device=R.device)
    assert float((R @ R.transpose(1, 2) - eye).abs().max()) < 1e-5 and float((torch.linalg.det(R) - 1).abs().max()) < 1e-5
    world = torch.einsum("nij,nhwj->nhwi", R, local[0]) + t[:, None, None, :]
    err = (world - pred["points"][0].double()).norm(dim=-1) / (world.norm(dim=-1) + 1e-12)
    assert float(err.max()) < 1e-5


def test_c5_two_scenes_recon_then_batched_chat(full_model):
    """BASELINE config 5's per-GPU shape at full width and depth: 2 scenes; each = 8-view pointmaps, then the chat prefill over
    the same 8 views (DINO geo prefill, 8 ViT images of 2916 patches through the device front end, a question), then the
    two scenes' greedy decodes as ONE batch (32 steps).  Each scene's batched ids equal its own batch-1 ids (flips only at
    near-ties of the batch-1 logits, towards the runner-up); cache lengths follow the packed lengths of every stage."""
    import numpy as np
    from PIL import Image
    from g2vlm_amd import host
    model, dims = full_model
    dev = model.device
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    nt = tok.new_token_ids
    rng = np.random.default_rng(5)
    tf = host.QwenVL2ImageTransform(768, 768, 14, device=dev, k_pad=model.weights["vit.patch.w"].shape[1])
    scenes = []
    for sidx in range(2):
        g = torch.Generator(); g.manual_seed(50 + sidx)
        views = torch.rand((8, 3, HW, HW), generator=g).cuda()
        vit = [tf([Image.fromarray(rng.integers(0, 256, size=(768, 768, 3), dtype=np.uint8))]) for _ in range(8)]
        scenes.append((views, vit, ["How far is the chair from the door?", "Describe the layout of the room and count its windows."][sidx]))

    def prefill(j):
        views, vit, prompt = scenes[j]
        it = iter(vit)
        return model._chat_prefill(tok, nt, lambda _im: next(it), None, views, prompt)

    for views, _, _ in scenes:                              # the reconstruction half of the config
        pred = model.recon(tok, nt, None, views)
        assert pred["points"].shape == (1, 8, HW, HW, 3) and torch.isfinite(pred["points"]).all()
    steps = 32
    singles = []
    for j in range(2):
        past, gi = prefill(j)
        n_sys = len(tok.encode("<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n<|im_start|>user\n"))
        n_q = len(tok.encode(scenes[j][2] + "<|im_end|>\n<|im_start|>assistant"))
        assert past.length == n_sys + 8 * (P + 2) + 8 * (729 + 2) + n_q
        st = model.engine.decode_begin(past, int(gi["packed_start_tokens"][0]), int(gi["packed_query_position_ids"][0, 0]), steps, use_graph=True)
        ids, lgs = [int(st["tok"][0])], []
        for _ in range(steps):
            ids.append(int(model.engine.decode_step(st)[0]))
            lgs.append(st["logits"].float().cpu().clone())
        singles.append((ids, lgs))
    pairs = [prefill(j) for j in range(2)]
    outs = model.generate_text_batch([p for p, _ in pairs], [gi for _, gi in pairs], steps + 1, end_token_id=None)
    for j in range(2):
        got, want = outs[j][:, 0].tolist(), singles[j][0]
        assert len(got) == steps + 1 and got[0] == want[0]
        fd = next((i for i in range(len(got)) if got[i] != want[i]), None)
        if fd is not None:                                 # a flip is legitimate only among tokens tied within 2 bf16 ulps
            lg = singles[j][1][fd - 1]
            top = float(lg.max())
            ulp = 2.0 ** (math.floor(math.log2(abs(top))) - 7)            # bf16 spacing at the maximum
            assert top - float(lg[got[fd]]) <= 2 * ulp, (j, fd, got[:fd + 1], want[:fd + 1])
