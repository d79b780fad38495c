"""GPU: the g2vlm_amd engine, driven through the reference's stage-method API, against the
reference-generated golden vectors (tests/golden) and the CPU oracle.

Tolerances.  The golden vectors come from the reference run under bf16 autocast; two bf16
implementations that differ only in fp32 accumulation order disagree at the bf16 rounding level
(~2^-9 per op), and the heads amplify that through exp(z).  So each quantity is checked two ways:
  (1) rel-L2 against the golden vector, bounded by a stated constant;
  (2) "as accurate as the reference": error against the oracle run in full precision
      (precise=True: fp32 weights/activations, fp64 attention) must not exceed
      `SLACK` x the bf16 reference's own error against that same precise result.
      For the point maps the error is the per-point relative error |p - p_precise| / |p_precise| compared at its
      median and 90th percentile, not one global L2 norm: local points are (x z, y z, z) with z = exp(z_raw), the
      random-weight fixtures have |z| up to 1e4, and the global norm is then set by a handful of pixels where one
      bf16 ulp of z_raw is a 3-6 % change - a coin flip per implementation (measured: the same engine with the
      128x128 GEMM kernel or with the split-K skinny kernel, which differ only in fp32 summation order, gives
      global ratios 1.01 and 1.42 on recon_tiny_3v_56x56).  The quantiles are stable under such re-orderings.
Greedy decode is compared token-for-token.
"""
import json
import os

import pytest
import torch
from safetensors.torch import load_file

pytestmark = pytest.mark.gpu

from oracle import synth  # noqa: E402  (checker only)
from oracle.g2vlm_oracle import NaiveCache as ONaiveCache, OracleG2VLM, vit_patchify  # noqa: E402

SLACK = 1.25          # large tensors; measured ratio on MI355X is 0.99-1.11 (profiles/parity_r01.md)
SLACK_SMALL = 4.0     # camera poses: 16 numbers per view, the ratio of two tiny error norms is noisy


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def point_err_quantiles(p, p_precise):
    """median and 90th percentile of |p - p_precise| / |p_precise| over the points of a [1,N,H,W,3] map"""
    p, q = p.double().cpu().reshape(-1, 3), p_precise.double().cpu().reshape(-1, 3)
    e = (p - q).norm(dim=1) / (q.norm(dim=1) + 1e-30)
    return float(e.quantile(0.5)), float(e.quantile(0.9))


def load(golden_dir, name):
    with open(os.path.join(golden_dir, name + ".json")) as f:
        meta = json.load(f)
    return meta, load_file(os.path.join(golden_dir, name + ".safetensors"))


def build(dims, seed, conf=False):
    from g2vlm_amd.g2vlm_utils import build_model, configs_from_dims
    sd = synth.synth_state_dict(dims, seed=seed, shapes=synth.param_shapes(dims, conf=True) if conf else None)
    return build_model(*configs_from_dims(dims), sd, "cuda"), sd


def run_recon(model, tok, imgs):
    from g2vlm_amd.modeling.g2vlm import NaiveCache
    nt = tok.new_token_ids
    out = {}
    past = NaiveCache(model.dims["llm"]["layers"], model.dims["llm"]["kv_heads"], "cuda")
    gi, nl, nr = model.prepare_prompts_addbos([0], [0], ["Reconstruct the 3D scene."], tok, nt)
    past = model.forward_cache_update_text(past, **gi)
    out["text_kv0_k"], out["text_kv0_v"] = past.key_cache[0].clone(), past.value_cache[0].clone()
    gi, nl, nr = model.prepare_dino_images_pi3(nl, nr, imgs, None, nt)
    model.engine.taps = {}
    past, last = model.forward_cache_update_dino(past, **gi)
    out["last_hidden"] = last
    out["dino_tokens"] = model.engine.taps.pop("dino_tokens")
    model.engine.taps = None
    nlay = model.dims["llm"]["layers"]
    out["geo_kv_last_k"], out["geo_kv_last_v"] = past.key_cache[nlay - 1], past.value_cache[nlay - 1]
    pred = model.reconstruct(past_key_values=past, selected_hidden_states=last, **gi)
    out.update({k: v for k, v in pred.items() if torch.is_tensor(v)})
    return gi, out


def precise_recon(sd, dims, tok, imgs):
    orc = OracleG2VLM(sd, dims, precise=True)
    nt = tok.new_token_ids
    cache = ONaiveCache(orc.num_layers)
    gi, nl, nr = orc.prepare_prompts([0], [0], ["Reconstruct the 3D scene."], tok, nt, bos=True)
    orc.forward_cache_update_text(cache, **gi)
    gi, nl, nr = orc.prepare_dino_images(nl, nr, imgs, nt)
    cache, last = orc.forward_cache_update_dino(cache, gi)
    out = {"last_hidden": last}
    out.update({k: v for k, v in orc.reconstruct(last, gi).items() if torch.is_tensor(v)})
    return out


# rel-L2 vs the reference golden: at most 2 x the largest value measured on MI355X over the six fixtures (profiles/parity_r01.md,
# profiles/parity_r02.md); the text-prefill KV is bit-exact up to single one-ulp
# roundings (see below)
BOUND = {"text_kv0_k": 0.0, "text_kv0_v": 0.0, "dino_tokens": 8e-3, "last_hidden": 9e-3, "geo_kv_last_k": 1e-2, "geo_kv_last_v": 1e-2,
         "global_points": 1.6e-2, "camera_poses": 2.6e-2, "local_points": 3.4e-2, "points": 3.4e-2, "conf": 1.5e-2}


@pytest.mark.parametrize("name", ["recon_tiny_2v_70x98", "recon_tiny_3v_56x56", "recon_tiny518_2v", "recon_real2_2v_56x84",
                                  "recon_tiny_conf_2v_56x70", "recon_real2_dl3dv_2v",
                                  # the use_dinov3 variant (DINOv3 encoder, patch-16 heads / grids; g2vlm.py:134, 169-172, 1172-1174):
                                  # the reference's modules driven stage by stage, see the fixtures' note
                                  "recon_dinov3_tiny_2v_64x96", "recon_dinov3_real2_3v_80x64"])
def test_recon_against_reference_golden(golden_dir, name):
    meta, g = load(golden_dir, name)
    dims = meta["dims"]
    model, sd = build(dims, meta["seed"], conf=bool(meta.get("conf")))
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    if meta.get("real_images"):
        # BASELINE config C2's inputs: two frames of the reference's examples/dl3dv as its own loader produced them (k/255)
        imgs = g["inp.images_u8"].float() / 255
    else:
        imgs = synth.synth_images(meta["n"], meta["h"], meta["w"], meta["seed"])
    gi, out = run_recon(model, tok, imgs)
    st = meta.get("strided")
    # host bookkeeping is integer work: bit-exact
    if meta.get("use_dinov3"):
        # the reference's prepare_dino_images_pi3 hard-codes a //14 grid (g2vlm.py:899) and has no output for this variant: the
        # bookkeeping that fed the reference's forward in the fixture is the oracle's prepare at patch 16, compared here
        assert model.use_dinov3 and model.dino_patch_size == 16 and model.weights.dinov3 is not None
        orc = OracleG2VLM(sd, dims)
        ogi, _, _ = orc.prepare_dino_images([gi["packed_key_value_indexes"].numel()], [int(gi["packed_position_ids"][0, 0])], imgs,
                                            tok.new_token_ids)
        for k in ("packed_position_ids", "packed_indexes", "packed_text_indexes", "packed_dino_token_indexes", "dino_token_seqlens",
                  "packed_seqlens", "packed_text_ids", "packed_key_value_indexes", "key_values_lens"):
            assert torch.equal(gi[k].long().cpu(), ogi[k].long()), k
    else:
        for k in ("packed_position_ids", "packed_indexes", "packed_text_indexes", "packed_dino_token_indexes"):
            assert torch.equal(gi[k].to(torch.int32), g["prep." + k]), k
    prec = precise_recon(sd, dims, tok, imgs) if not st else None
    report = {}
    assert (out.get("conf") is not None) == bool(meta.get("conf"))
    for k, bound in BOUND.items():
        if k == "conf" and not meta.get("conf"):
            continue
        mine = out[k].float().cpu()
        if st and mine.dim() == 5 and mine.shape[2] > 64:
            mine = mine[:, :, ::st, ::st]
        elif st and k in ("last_hidden", "geo_kv_last_k", "geo_kv_last_v"):
            mine = mine[::5]
        elif st and k == "dino_tokens":
            mine = mine[:, ::5]
        ref = g["ref." + k].float()
        assert torch.isfinite(mine).all(), k
        r = rel(mine, ref)
        report[k] = r
        if bound == 0.0:
            # the 8-token text prefill: bit-exact on the round-1 fixtures; a different fp32 summation order in a GEMM may
            # move a qkv output that sits on a bf16 rounding boundary (K = 1536: a few of the 2048 values per layer), and
            # k-norm + RoPE pass that one-ulp step on to the rotation partner, so the criterion is "at most 1 element in 100
            # differs, rel-L2 <= 5e-4" (measured: 0 differing on seven fixtures; 0.27 %, 1.3e-4 on recon_dinov3_real2_3v_80x64)
            ne = mine != ref
            report[k + ".exact_frac"] = 1.0 - float(ne.float().mean())
            assert float(ne.float().mean()) <= 1e-2 and r <= 5e-4, f"{k}: rel-L2 {r:.3e}, {int(ne.sum())} elements differ"
            continue
        assert r < bound, f"{k}: rel-L2 {r:.3e} vs reference golden exceeds {bound}"
        if prec is not None and k in prec:
            e_mine, e_ref = rel(mine, prec[k]), rel(ref, prec[k])
            report[k + ".vs_precise"] = (e_mine, e_ref)
            # world points inherit the pose noise (points = pose . local), so they get half the small-tensor slack
            sl = {"camera_poses": SLACK_SMALL, "points": SLACK_SMALL / 2}.get(k, SLACK)
            if mine.dim() == 5 and mine.shape[-1] == 3:
                qm, qr = point_err_quantiles(mine, prec[k]), point_err_quantiles(ref, prec[k])
                report[k + ".point_err_q50_q90"] = (qm, qr)
                for a_, b_, nm in zip(qm, qr, ("median", "p90")):
                    assert a_ <= sl * b_ + 1e-6, f"{k}: {nm} per-point error vs full-precision {a_:.3e} > {sl} x reference's {b_:.3e}"
            else:
                assert e_mine <= sl * e_ref + 1e-6, f"{k}: error vs full-precision {e_mine:.3e} > {sl} x reference's {e_ref:.3e}"
    print(name, json.dumps(report))
    assert out["points"].shape == (1, meta["n"], meta["h"], meta["w"], 3) and out["camera_poses"].shape == (1, meta["n"], 4, 4)
    assert out["images"].shape == (1, meta["n"], 3, meta["h"], meta["w"])


def test_recon_dinov3_from_pil_images(golden_dir):
    """`recon` of the use_dinov3 variant from PIL images: the loader is load_and_resize16 (518 -> 288x512 for a 540x960
    source), grids are //16, and the call equals the stage path fed the loader's tensor."""
    import numpy as np
    from PIL import Image
    from g2vlm_amd import host
    meta, _ = load(golden_dir, "recon_dinov3_tiny_2v_64x96")
    dims = meta["dims"]
    model, sd = build(dims, meta["seed"])
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    rng = np.random.RandomState(3)
    pil = [Image.fromarray(rng.randint(0, 256, size=(540, 960, 3), dtype=np.uint8), "RGB") for _ in range(2)]
    pred = model.recon(tok, tok.new_token_ids, None, pil)
    assert pred["points"].shape == (1, 2, 288, 512, 3) and pred["images"].shape == (1, 2, 3, 288, 512)
    assert torch.isfinite(pred["points"]).all() and torch.isfinite(pred["camera_poses"]).all()
    frames = host.load_and_resize16(pil, 518)
    assert torch.equal(pred["images"][0].cpu(), frames)
    again = model.recon(tok, tok.new_token_ids, None, frames)
    for k in ("points", "local_points", "global_points", "camera_poses"):
        assert torch.equal(pred[k], again[k]), k
    # the view-sharded entry on one rank is the same computation (the variant is sharded by window since round 3,
    # tests/test_sharded_gpu.py::test_dinov3_view_sharded_matches_unsharded)
    from g2vlm_amd.sharded import recon_view_sharded, LocalComm
    one = recon_view_sharded(model, LocalComm(), tok, tok.new_token_ids, frames)
    for k in ("points", "local_points", "global_points", "camera_poses"):
        assert torch.equal(pred[k], one[k]), k
    # chat_with_recon over the same variant: the geometry prefill goes through the DINOv3 encoder, the cache length follows the
    # //16 grid, the decode runs (graph replay == eager)
    gen = torch.Generator(); gen.manual_seed(77)
    pv, thw = vit_patchify(torch.randn((1, 3, 8 * 14, 8 * 14), generator=gen))
    small = frames[:1, :, :64, :96].contiguous()
    outs = []
    for use_graph in (True, False):
        model.use_decode_graph = use_graph
        past, gi = model._chat_prefill(tok, tok.new_token_ids, lambda _im: (pv, torch.tensor([list(thw)])), None, small, "where is the door")
        n_sys = len(tok.encode("<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n<|im_start|>user\n"))
        n_q = len(tok.encode("where is the door<|im_end|>\n<|im_start|>assistant"))
        assert past.length == n_sys + (4 * 6 + 2) + (16 + 2) + n_q
        outs.append(model.generate_text(past_key_values=past, max_length=10, end_token_id=None, **gi)[:, 0].tolist())
    model.use_decode_graph = True
    assert outs[0] == outs[1] and len(outs[0]) == 10


@pytest.mark.parametrize("name", ["chat_tiny", "chat_real2"])
def test_multi_image_vit_prefill_in_one_pass_matches_stage_by_stage(golden_dir, name):
    """forward_cache_update_vit_multi: the chat prefill's images (reference: one forward_cache_update_vit per image,
    g2vlm.py:1362-1370) as ONE ViT pass + ONE und prefill with an attention window per image.  Against the stage-by-stage
    path on the same model: the bookkeeping is identical, every image's cache rows agree to bf16 GEMM-order noise at the first
    and the last layer, the rows of the stages before are untouched, and the greedy ids agree (a flip only at a near-tie).
    3 views of one grid, then one of another grid (closes the run: 3 images batched + 1 alone)."""
    meta, g = load(golden_dir, name)
    dims = meta["dims"]
    model, sd = build(dims, meta["seed"])
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    nt = tok.new_token_ids
    imgs = synth.synth_images(4, meta["h"], meta["w"], meta["seed"])
    grids = [meta["vit_grid"]] * 3 + [(meta["vit_grid"][0] + 2, meta["vit_grid"][1])]
    vit_inputs = []
    for i, (gh, gw) in enumerate(grids):
        gen = torch.Generator(); gen.manual_seed(4321 + i)
        vit_inputs.append(vit_patchify(torch.randn((1, 3, gh * 14, gw * 14), generator=gen)))

    def prefill(batched):
        it = iter(vit_inputs)

        def image_transform(_imgs):
            pv, thw = next(it)
            return pv, torch.tensor([list(thw)])
        model.batch_vit_prefill = batched
        try:
            return model._chat_prefill(tok, nt, image_transform, None, imgs, meta["prompt"])
        finally:
            model.batch_vit_prefill = True

    calls = []
    orig = model.forward_cache_update_vit_multi
    model.forward_cache_update_vit_multi = lambda past, gis, **kw: calls.append(len(gis)) or orig(past, gis, **kw)
    past_b, gi_b = prefill(True)
    model.forward_cache_update_vit_multi = orig
    assert calls == [3]
    past_s, gi_s = prefill(False)
    assert past_b.length == past_s.length
    for k in gi_s:
        assert torch.equal(gi_b[k], gi_s[k]), k
    nl = dims["llm"]["layers"]
    S = [(gh // 2) * (gw // 2) + 2 for gh, gw in grids]
    n_q = len(tok.encode(meta["prompt"] + "<|im_end|>\n<|im_start|>assistant"))
    v0 = past_s.length - n_q - sum(S)                                    # first row of the first image
    worst = {}
    for lay in (0, nl - 1):
        kb, ks, vb, vs = past_b.key_cache[lay], past_s.key_cache[lay], past_b.value_cache[lay], past_s.value_cache[lay]
        assert torch.equal(kb[:v0], ks[:v0]) and torch.equal(vb[:v0], vs[:v0]), lay       # system prompt + geometry views
        lo = v0
        for j, s_ in enumerate(S + [n_q]):                                                 # the images, then the question after them
            worst[(lay, j)] = max(rel(kb[lo:lo + s_], ks[lo:lo + s_]), rel(vb[lo:lo + s_], vs[lo:lo + s_]))
            lo += s_
    print(name, "batched vs stage-by-stage K/V rel-L2:", {k: round(v, 5) for k, v in worst.items()})
    # two bf16 pipelines that differ in GEMM tiling and attention schedule: rounding-level noise that grows with depth (the
    # ViT keeps its residual stream in bf16); a wrong window or row would show as O(1).  Measured <= 9e-3 (layer 1, tiny dims)
    assert max(worst.values()) < 2e-2, worst
    steps = 12
    ids_b, lg_b = _decode_single(model, past_b, gi_b, steps)
    ids_s, lg_s = _decode_single(model, past_s, gi_s, steps)
    fd = next((i for i in range(len(ids_s)) if ids_b[i] != ids_s[i]), None)
    assert fd is None or _near_tie(lg_s[fd - 1], ids_b[fd]), (fd, ids_b, ids_s)
    assert rel(lg_b[0], lg_s[0]) < 3e-2


@pytest.mark.parametrize("name,min_div", [("chat_tiny", 8), ("chat_real2", 6)])
def test_chat_greedy_token_exact(golden_dir, name, min_div):
    """chat_real2: real widths (LLM 1536 / ViT 1280 / DINO 1024, 2 layers each): the decode kernels at their real K / N."""
    meta, g = load(golden_dir, name)
    dims = meta["dims"]
    model, sd = build(dims, meta["seed"])
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    imgs = synth.synth_images(meta["n"], meta["h"], meta["w"], meta["seed"])
    vit_inputs = []
    for i in range(meta["n"]):
        gen = torch.Generator(); gen.manual_seed(1234 + i)
        vit_inputs.append(vit_patchify(torch.randn((1, 3, meta["vit_grid"][0] * 14, meta["vit_grid"][1] * 14), generator=gen)))
    it = iter(vit_inputs)

    def image_transform(_imgs):
        pv, thw = next(it)
        return pv, torch.tensor([list(thw)])

    got = []
    dec = tok.decode
    tok.decode = lambda ids: got.extend(int(v) for v in ids) or ""
    model.chat_with_recon(tok, tok.new_token_ids, image_transform, None, images=imgs, prompt=meta["prompt"],
                          max_length=meta["max_length"])
    tok.decode = dec
    ref = g["ref.ids"].tolist()
    first_div = next((i for i, (a, b) in enumerate(zip(got, ref)) if a != b), None)
    print(name, "first divergence:", first_div, "of", len(ref))
    if first_div is not None:
        # hazard H2: argmax over bf16 logits.  A divergence is legitimate only at a near-tie of the REFERENCE's own
        # logits (top-2 margin <= 2 bf16 ulps) and only towards the reference's runner-up; logits row i produced ids[i].
        lg = g["ref.logits"].float()[first_div]
        top = lg.topk(2)
        margin = float(top.values[0] - top.values[1])
        ulp = 2.0 ** -8 * float(top.values[0].abs())
        assert first_div >= min_div, f"diverged too early ({first_div})"
        assert margin <= 2 * ulp and got[first_div] == int(top.indices[1]), (
            f"greedy ids diverge at step {first_div} with reference top-2 margin {margin:.4f} (bf16 ulp {ulp:.4f}): "
            f"{got[:first_div + 2]} vs {ref[:first_div + 2]}")
    else:
        assert got == ref


@pytest.mark.parametrize("name", ["chat_tiny", "chat_real2"])
def test_vit_tokens(golden_dir, name):
    meta, g = load(golden_dir, name)
    dims = meta["dims"]
    model, sd = build(dims, meta["seed"])
    from g2vlm_amd import host
    gen = torch.Generator(); gen.manual_seed(1234)
    pv, thw = vit_patchify(torch.randn((1, 3, meta["vit_grid"][0] * 14, meta["vit_grid"][1] * 14), generator=gen))
    kp = model.weights["vit.patch.w"].shape[1]
    pvd = torch.nn.functional.pad(pv, (0, kp - pv.shape[1])).cuda()
    D = dims["vit"]["embed"] // dims["vit"]["heads"]
    cos, sin = host.vit_rot_pos(*thw, D)
    emb = model.engine.vit_forward(pvd, thw, cos.cuda(), sin.cuda())
    r = rel(emb, g["ref.vit_tokens"])
    assert r < 2e-2, r


def test_no_cpu_fallback_when_library_missing(monkeypatch):
    from g2vlm_amd import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", "/nonexistent/libg2vlm_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hip.lib()


def test_full_size_c3_properties():
    """BASELINE config C3 at full width and depth (8 views 518x518, 24 DINO + 28 MoT + 15 decoder blocks, random-init
    weights as bench.py builds them).  No oracle finishes this size in seconds, so the checks are size-independent
    properties of the path (reference g2vlm.py:1200-1238):
      * every output finite, shapes / dtypes of the reference's dict;
      * world points = pose . [local, 1] recomputed on the host from the returned poses and local points (g2vlm.py:1226);
      * local z = exp(z_raw) > 0 and x/z, y/z finite (g2vlm.py:1219-1221); rotations orthonormal with det +1
        (camera_head.py:84-90: SVD projection);
      * the text prefix's KV rows are untouched by the geo prefill that appends after them (NaiveCache semantics,
        qwen2vl.py:626-634), and the cache length is T0 + N (P + 2);
      * a second run on the same inputs is bit-identical (deterministic split-K / stream-K merge orders)."""
    from g2vlm_amd.g2vlm_utils import build_model, configs_from_dims
    from g2vlm_amd.modeling.g2vlm import NaiveCache
    from g2vlm_amd.synthetic import REAL_DIMS, SyntheticStateDict
    dims, dev = REAL_DIMS, torch.device("cuda", 0)
    model = build_model(*configs_from_dims(dims), SyntheticStateDict(dims, dev, seed=0), dev)
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    nt = tok.new_token_ids
    g = torch.Generator(); g.manual_seed(7)
    N, HW = 8, 518
    imgs = torch.rand((N, 3, HW, HW), generator=g)
    P = (HW // 14) ** 2

    def run():
        past = NaiveCache(dims["llm"]["layers"], dims["llm"]["kv_heads"], dev)
        gi, nl, nr = model.prepare_prompts_addbos([0], [0], ["Reconstruct the 3D scene."], tok, nt)
        t0 = int(gi["packed_text_ids"].numel())
        past = model.forward_cache_update_text(past, **gi)
        kv_text = (past.key_cache[0][:t0].clone(), past.value_cache[27][:t0].clone())
        gi, nl, nr = model.prepare_dino_images_pi3(nl, nr, imgs, None, nt)
        past, last = model.forward_cache_update_dino(past, **gi)
        pred = model.reconstruct(past_key_values=past, selected_hidden_states=last, **gi)
        return t0, kv_text, past, last, pred

    t0, kv_text, past, last, pred = run()
    assert past.length == t0 + N * (P + 2)
    assert torch.equal(past.key_cache[0][:t0], kv_text[0]) and torch.equal(past.value_cache[27][:t0], kv_text[1])
    assert last.shape == (N * (P + 2), dims["llm"]["hidden"]) and torch.isfinite(last).all()
    for k, shp in (("points", (1, N, HW, HW, 3)), ("local_points", (1, N, HW, HW, 3)), ("global_points", (1, N, HW, HW, 3)),
                   ("camera_poses", (1, N, 4, 4)), ("images", (1, N, 3, HW, HW))):
        assert pred[k].shape == shp and pred[k].dtype == torch.float32 and torch.isfinite(pred[k]).all(), k
    assert pred["conf"] is None
    local, poses = pred["local_points"].double(), pred["camera_poses"].double()
    assert (local[..., 2] > 0).all()
    R, t = poses[0, :, :3, :3], poses[0, :, :3, 3]
    eye = torch.eye(3, dtype=torch.float64, device=dev)
    assert float((R @ R.transpose(1, 2) - eye).abs().max()) < 1e-5
    assert float((torch.linalg.det(R) - 1).abs().max()) < 1e-5
    assert float((poses[0, :, 3] - torch.tensor([0, 0, 0, 1.0], dtype=torch.float64, device=dev)).abs().max()) == 0
    world = torch.einsum("nij,nhwj->nhwi", R, local[0]) + t[:, None, None, :]
    err = (world - pred["points"][0].double()).norm(dim=-1) / (world.norm(dim=-1) + 1e-12)
    assert float(err.max()) < 1e-5, float(err.max())
    # determinism
    _, _, past2, last2, pred2 = run()
    assert torch.equal(last, last2)
    for k in ("points", "local_points", "global_points", "camera_poses"):
        assert torch.equal(pred[k], pred2[k]), k
    # two scenes in flight on two streams of one process (bench.py's secondary figure): kernels of the two interleave on the
    # device, nothing is shared between them (per-stream attention / GEMM scratch), results stay bit-identical
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    for s_ in streams:
        s_.wait_stream(torch.cuda.current_stream())
    outs = []
    for i in range(4):
        with torch.cuda.stream(streams[i % 2]):
            outs.append(run())
    torch.cuda.synchronize()
    for (_, _, _, last_i, pred_i) in outs:
        assert torch.equal(last, last_i)
        for k in ("points", "local_points", "global_points", "camera_poses"):
            assert torch.equal(pred[k], pred_i[k]), k


def test_full_size_chat_properties():
    """BASELINE configs C1 / C5 at full width and depth: one 518x518 view through DINO + MoT geo prefill, the same view
    as a 756x756 Qwen2-VL ViT input (2916 patches -> 729 merged tokens, 32 ViT blocks at width 1280), the question,
    then a greedy decode replayed from the hipGraph.  Size-independent properties (reference g2vlm.py:1305-1410):
      * the KV cache grows by exactly the packed lengths of every stage (NaiveCache append semantics);
      * ids are valid vocabulary indices, the first is the assistant start token, the decode is deterministic;
      * graph-replayed decode == the same steps launched eagerly (same kernels, same device-side state)."""
    from g2vlm_amd.g2vlm_utils import build_model, configs_from_dims
    from g2vlm_amd.modeling.g2vlm import NaiveCache
    from g2vlm_amd.synthetic import REAL_DIMS, SyntheticStateDict
    dims, dev = REAL_DIMS, torch.device("cuda", 0)
    model = build_model(*configs_from_dims(dims), SyntheticStateDict(dims, dev, seed=0), dev)
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    nt = tok.new_token_ids
    g = torch.Generator(); g.manual_seed(11)
    imgs = torch.rand((1, 3, 518, 518), generator=g)
    pv, thw = vit_patchify(torch.randn((1, 3, 756, 756), generator=g))          # [2916, 1176], (1, 54, 54)

    def image_transform(_imgs):
        return pv, torch.tensor([list(thw)])

    def run(use_graph, max_length=12):
        model.use_decode_graph = use_graph
        past = NaiveCache(dims["llm"]["layers"], dims["llm"]["kv_heads"], dev)
        lens = []
        gi, nl, nr = model.prepare_prompts_pure_text([0], [0], ["<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n<|im_start|>user\n"], tok, nt)
        past = model.forward_cache_update_text(past, **gi); lens.append(past.length)
        gi, nl, nr = model.prepare_dino_images_pi3(nl, nr, imgs, None, nt)
        past, _ = model.forward_cache_update_dino(past, **gi); lens.append(past.length)
        gi, nl, nr = model.prepare_vit_images(nl, nr, [None], image_transform, nt)
        past = model.forward_cache_update_vit(past, **gi); lens.append(past.length)
        gi, nl, nr = model.prepare_prompts_pure_text(nl, nr, ["How far is the chair from the door?<|im_end|>\n<|im_start|>assistant"], tok, nt)
        past = model.forward_cache_update_text(past, **gi); lens.append(past.length)
        gs = model.prepare_start_tokens(nl, nr, tok, nt)
        ids = model.generate_text(past_key_values=past, max_length=max_length, do_sample=False, end_token_id=None, **gs)
        return lens, ids, int(gs["packed_start_tokens"][0])

    lens, ids, start = run(True)
    P = 37 * 37
    assert lens[1] - lens[0] == P + 2                     # geo prefill: patches + <|vision_start|>/<|vision_end|>
    assert lens[2] - lens[1] == 729 + 2                   # ViT tokens after the 2x2 merge + markers
    assert lens[3] > lens[2]
    assert ids.shape == (12, 1) and int(ids[0, 0]) == start
    assert int(ids.min()) >= 0 and int(ids.max()) < dims["llm"]["vocab"]
    lens2, ids2, _ = run(True)
    assert lens2 == lens and torch.equal(ids, ids2)
    _, ids3, _ = run(False)
    assert torch.equal(ids, ids3), "graph replay and eager decode disagree"
    model.use_decode_graph = True


def _near_tie(logits_row, other_tok, ulps=2):
    """A greedy flip is legitimate only at a near-tie of the logits (hazard H2): the other token's logit must be within
    `ulps` bf16 ulps of the maximum (random-weight logits over a 152 k vocabulary hold exact ties among several tokens)."""
    lg = logits_row.float()
    top = float(lg.max())
    return top - float(lg[other_tok]) <= ulps * 2.0 ** -8 * abs(top)


def _decode_single(model, past, gi, steps):
    """batch-1 engine decode of `steps` tokens: (ids, per-step logits)"""
    eng = model.engine
    st = eng.decode_begin(past, int(gi["packed_start_tokens"][0]), int(gi["packed_query_position_ids"][0, 0]), steps, use_graph=False)
    ids, lg = [int(st["tok"][0])], []
    for _ in range(steps):
        ids.append(int(eng.decode_step(st)[0]))
        lg.append(st["logits"].float().cpu().clone())
    return ids, lg


def _check_batch_vs_single(model, scenes, steps, use_graph):
    """scenes: list of callables returning a freshly prefilled (past, start_inputs).  The batched decode must give every
    scene the ids (and, up to bf16 GEMM-order noise, the logits) of its own batch-1 decode."""
    eng = model.engine
    singles = []
    for mk in scenes:
        past, gi = mk()
        singles.append(_decode_single(model, past, gi, steps))
    pairs = [mk() for mk in scenes]
    lens = [p.length for p, _ in pairs]
    st = eng.decode_begin_batch([p for p, _ in pairs], [int(g["packed_start_tokens"][0]) for _, g in pairs],
                                [int(g["packed_query_position_ids"][0, 0]) for _, g in pairs], steps, use_graph=use_graph)
    B = len(scenes)
    ids = [[t] for t in st["tok"].tolist()]
    alive = [True] * B
    for s in range(steps):
        tok = eng.decode_step_batch(st).tolist()
        lg = st["logits"].float().cpu()
        for j in range(B):
            if not alive[j]:
                continue
            ref_ids, ref_lg = singles[j]
            # (full depth, random weights: fp32-order noise of the M = B MFMA GEMMs vs the M = 1 GEMVs, amplified by 28 layers
            #  and carried through the cache from step to step; measured up to 2.1e-2 at step 8)
            assert rel(lg[j], ref_lg[s]) < 3e-2, (j, s, rel(lg[j], ref_lg[s]))
            ids[j].append(tok[j])
            if tok[j] != ref_ids[s + 1]:
                assert _near_tie(ref_lg[s], tok[j]), f"scene {j} step {s}: {tok[j]} vs {ref_ids[s + 1]} is not a near-tie flip"
                alive[j] = False                       # the continuations legitimately differ from here on
    for j in range(B):
        assert ids[j][0] == singles[j][0][0]
    # cache rows written by the batched steps sit behind each scene's prefill, which is untouched
    for j, (p, _) in enumerate(pairs):
        assert torch.equal(st["k"][0][j, :lens[j]], p.k[0][:lens[j]]) and torch.equal(st["v"][-1][j, :lens[j]], p.v[-1][:lens[j]])
    assert st["len"].tolist() == [n + 1 + steps for n in lens]
    return ids, [s[0] for s in singles], alive


@pytest.mark.parametrize("name", ["chat_tiny", "chat_real2"])
def test_batched_decode_matches_single_and_golden(golden_dir, name):
    """SURVEY 8f-3 (batch > 1 decode; the reference asserts batch 1, g2vlm.py:1006, 1137).  Three scenes share the
    weights: the golden scene, the same views under a longer question (different cache length and RoPE position), and the
    golden scene again.  Each scene's ids equal its own batch-1 decode; the golden scenes' ids equal the reference's
    (ref.ids) under the same near-tie rule as test_chat_greedy_token_exact."""
    meta, g = load(golden_dir, name)
    dims = meta["dims"]
    model, sd = build(dims, meta["seed"])
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    imgs = synth.synth_images(meta["n"], meta["h"], meta["w"], meta["seed"])

    def scene(prompt):
        def mk():
            vit_inputs = []
            for i in range(meta["n"]):
                gen = torch.Generator(); gen.manual_seed(1234 + i)
                vit_inputs.append(vit_patchify(torch.randn((1, 3, meta["vit_grid"][0] * 14, meta["vit_grid"][1] * 14), generator=gen)))
            it = iter(vit_inputs)

            def image_transform(_imgs):
                pv, thw = next(it)
                return pv, torch.tensor([list(thw)])
            return model._chat_prefill(tok, tok.new_token_ids, image_transform, None, imgs, prompt)
        return mk

    steps = meta["max_length"] - 1
    scenes = [scene(meta["prompt"]), scene(meta["prompt"] + " and how large is the room in square metres"), scene(meta["prompt"])]
    for use_graph in (False, True):
        ids, single_ids, alive = _check_batch_vs_single(model, scenes, steps, use_graph)
        assert ids[0] == ids[2]
        ref = g["ref.ids"].tolist()
        got = ids[0][1:][:len(ref)]                      # ref.ids starts after the assistant start token
        first_div = next((i for i, (a, b) in enumerate(zip(got, ref)) if a != b), None)
        if first_div is not None:
            assert _near_tie(g["ref.logits"][first_div], got[first_div]), (first_div, got, ref)
    # the public entry point: same scenes through chat_with_recon_batch, EOS handling included
    texts = []
    dec = tok.decode
    tok.decode = lambda ids_: texts.append([int(v) for v in ids_]) or ""

    def scene_args(prompt):
        vit_inputs = []
        for i in range(meta["n"]):
            gen = torch.Generator(); gen.manual_seed(1234 + i)
            vit_inputs.append(vit_patchify(torch.randn((1, 3, meta["vit_grid"][0] * 14, meta["vit_grid"][1] * 14), generator=gen)))
        return vit_inputs

    queue = scene_args(meta["prompt"]) + scene_args("x")
    it = iter(queue)

    def image_transform(_imgs):
        pv, thw = next(it)
        return pv, torch.tensor([list(thw)])
    model.chat_with_recon_batch(tok, tok.new_token_ids, image_transform, None, [(imgs, meta["prompt"]), (imgs, "x")], meta["max_length"])
    it = iter(scene_args(meta["prompt"]))
    model.chat_with_recon(tok, tok.new_token_ids, image_transform, None, images=imgs, prompt=meta["prompt"], max_length=meta["max_length"])
    tok.decode = dec
    assert len(texts) == 3 and len(texts[0]) >= 1
    n = min(len(texts[0]), len(texts[2]))
    fd = next((i for i in range(n) if texts[0][i] != texts[2][i]), None)
    assert fd is None or fd >= 4, (texts[0], texts[2])           # batch and batch-1 public paths agree (near-tie flips aside)


def test_batched_decode_beyond_eight_scenes_keeps_the_gemm_body(golden_dir):
    """B <= 8 decodes on the persistent-grid GEMVs (decode_batch.hip); more slots fall back to the skinny-GEMM body.  Nine
    scenes (two different questions alternating) through that body: every scene still gets its own batch-1 ids."""
    meta, g = load(golden_dir, "chat_tiny")
    dims = meta["dims"]
    model, sd = build(dims, meta["seed"])
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    imgs = synth.synth_images(meta["n"], meta["h"], meta["w"], meta["seed"])

    def scene(prompt):
        def mk():
            gen = torch.Generator(); gen.manual_seed(1234)
            pv, thw = vit_patchify(torch.randn((1, 3, meta["vit_grid"][0] * 14, meta["vit_grid"][1] * 14), generator=gen))
            return model._chat_prefill(tok, tok.new_token_ids, lambda _im: (pv, torch.tensor([list(thw)])), None, imgs, prompt)
        return mk

    scenes = [scene(meta["prompt"] if j % 2 == 0 else meta["prompt"] + " and why") for j in range(9)]
    ids, single_ids, alive = _check_batch_vs_single(model, scenes, 10, use_graph=True)
    assert ids[0] == ids[2] == ids[8] and ids[1] == ids[3]


def test_full_size_batched_decode_properties():
    """Batched decode at full width and depth (28 und layers, vocab 151 936): two different scenes (one 518x518 view each,
    different questions, hence different cache lengths and positions) decoded together give each scene its batch-1 ids,
    graph replay == eager, and the state advances by one row per scene and step."""
    from g2vlm_amd.g2vlm_utils import build_model, configs_from_dims
    from g2vlm_amd.synthetic import REAL_DIMS, SyntheticStateDict
    dims, dev = REAL_DIMS, torch.device("cuda", 0)
    model = build_model(*configs_from_dims(dims), SyntheticStateDict(dims, dev, seed=0), dev)
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    g = torch.Generator(); g.manual_seed(21)
    views = [torch.rand((1, 3, 518, 518), generator=g) for _ in range(2)]
    pvs = [vit_patchify(torch.randn((1, 3, 392, 392), generator=g)) for _ in range(2)]
    prompts = ["How far is the chair from the door?", "Describe the layout of this room and count the windows you can see."]

    def scene(j):
        def mk():
            def image_transform(_imgs):
                return pvs[j][0], torch.tensor([list(pvs[j][1])])
            return model._chat_prefill(tok, tok.new_token_ids, image_transform, None, views[j], prompts[j])
        return mk

    scenes = [scene(0), scene(1)]
    ids_e, single, alive_e = _check_batch_vs_single(model, scenes, 10, use_graph=False)
    ids_g, _, alive_g = _check_batch_vs_single(model, scenes, 10, use_graph=True)
    assert ids_e == ids_g, "graph replay and eager batched decode disagree"
    for j in range(2):
        assert all(0 <= t < dims["llm"]["vocab"] for t in ids_e[j])


def test_vit_prefill_with_device_preprocessing_is_bit_identical(golden_dir):
    """an image that goes through host.QwenVL2ImageTransform(device=...) (uint8 upload, one kernel) fills
    the KV cache with exactly the bytes the host-preprocessed fp32 patch matrix does."""
    import numpy as np
    from PIL import Image
    from g2vlm_amd import host
    from g2vlm_amd.modeling.g2vlm import NaiveCache
    meta, g = load(golden_dir, "chat_tiny")
    dims = meta["dims"]
    model, sd = build(dims, meta["seed"])
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    img = Image.fromarray(np.random.default_rng(3).integers(0, 256, size=(120, 160, 3), dtype=np.uint8))
    kp = model.weights["vit.patch.w"].shape[1]
    caches = []
    for tr in (host.QwenVL2ImageTransform(112, 140), host.QwenVL2ImageTransform(112, 140, device="cuda", k_pad=kp)):
        past = NaiveCache(dims["llm"]["layers"], dims["llm"]["kv_heads"], torch.device("cuda", 0))
        gi, nl, nr = model.prepare_prompts_pure_text([0], [0], ["hello there"], tok, tok.new_token_ids)
        past = model.forward_cache_update_text(past, **gi)
        gi, nl, nr = model.prepare_vit_images(nl, nr, [img], tr, tok.new_token_ids)
        past = model.forward_cache_update_vit(past, **gi)
        caches.append(past)
    a, b = caches
    assert a.length == b.length and a.length > 0
    for i in range(dims["llm"]["layers"]):
        assert torch.equal(a.k[i][:a.length], b.k[i][:b.length]) and torch.equal(a.v[i][:a.length], b.v[i][:b.length])


@pytest.mark.parametrize("name", ["chat_tiny"])
def test_continuous_batching_matches_single(golden_dir, name):
    """generate_text_stream (continuous batching): 5 scenes with different questions through 2 slots, with an
    EOS id chosen so that scenes finish at different steps and slots are refilled mid-stream; every scene's ids equal its
    own batch-1 generate_text (flips only at near-ties of the batch-1 logits)."""
    meta, g = load(golden_dir, name)
    dims = meta["dims"]
    model, sd = build(dims, meta["seed"])
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    imgs = synth.synth_images(meta["n"], meta["h"], meta["w"], meta["seed"])
    prompts = [meta["prompt"], "where is the door", meta["prompt"] + " and what colour is the sofa next to it", "count the chairs", "is the window open or closed now"]

    def scene(prompt):
        def mk():
            vit_inputs = []
            for i in range(meta["n"]):
                gen = torch.Generator(); gen.manual_seed(1234 + i)
                vit_inputs.append(vit_patchify(torch.randn((1, 3, meta["vit_grid"][0] * 14, meta["vit_grid"][1] * 14), generator=gen)))
            it = iter(vit_inputs)

            def image_transform(_imgs):
                pv, thw = next(it)
                return pv, torch.tensor([list(thw)])
            return model._chat_prefill(tok, tok.new_token_ids, image_transform, None, imgs, prompt)
        return mk

    max_length = 14
    # batch-1 runs without EOS first; then pick as EOS an id that the first scene emits mid-way
    singles = []
    for p in prompts:
        past, gi = scene(p)()
        singles.append(_decode_single(model, past, gi, max_length))
    eos = singles[0][0][5]
    want = []
    for ids, _ in singles:
        cut = next((i for i in range(1, len(ids)) if ids[i] == eos), len(ids))
        want.append(ids[:min(cut, max_length)])
    kv_max = max(scene(p)()[0].length for p in prompts[:1]) + 64
    for use_graph in (False, True):
        model.use_decode_graph = use_graph
        got = model.generate_text_stream([scene(p) for p in prompts], max_batch=2, max_length=max_length, max_kv_len=kv_max,
                                         end_token_id=eos, chunk=4)
        assert len(got) == len(prompts)
        for j, (gt, w) in enumerate(zip(got, want)):
            gl = gt[:, 0].tolist()
            fd = next((i for i in range(min(len(gl), len(w))) if gl[i] != w[i]), None)
            if fd is None:
                assert len(gl) == len(w), (j, gl, w)
            else:
                assert _near_tie(singles[j][1][fd - 1], gl[fd]), (j, fd, gl, w)
    model.use_decode_graph = True


def test_generate_text_graph_mode_appends_to_the_callers_cache(golden_dir):
    """Graph mode decodes in an engine-owned KV block and reuses the captured step across calls (Engine.decode_begin /
    decode_end): the caller's cache must still end up with the appended rows (NaiveCache semantics, qwen2vl.py:626-634),
    bit-identical to an eager decode in the caller's cache, on a first call (capture) and on a second (reuse)."""
    meta, g = load(golden_dir, "chat_tiny")
    dims = meta["dims"]
    model, sd = build(dims, meta["seed"])
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    imgs = synth.synth_images(meta["n"], meta["h"], meta["w"], meta["seed"])

    def prefill():
        vit_inputs = []
        for i in range(meta["n"]):
            gen = torch.Generator(); gen.manual_seed(1234 + i)
            vit_inputs.append(vit_patchify(torch.randn((1, 3, meta["vit_grid"][0] * 14, meta["vit_grid"][1] * 14), generator=gen)))
        it = iter(vit_inputs)

        def image_transform(_imgs):
            pv, thw = next(it)
            return pv, torch.tensor([list(thw)])
        return model._chat_prefill(tok, tok.new_token_ids, image_transform, None, imgs, meta["prompt"])

    runs = []
    for use_graph in (False, True, True):
        model.use_decode_graph = use_graph
        past, gi = prefill()
        n0 = past.length
        ids = model.generate_text(past_key_values=past, max_length=9, end_token_id=None, **gi)
        assert past.length == n0 + 9 and ids.shape == (9, 1)
        runs.append((ids, [past.k[i][:past.length].clone() for i in range(dims["llm"]["layers"])], past.v[-1][:past.length].clone()))
    model.use_decode_graph = True
    for ids, ks, v in runs[1:]:
        assert torch.equal(ids, runs[0][0])
        assert all(torch.equal(a, b) for a, b in zip(ks, runs[0][1])) and torch.equal(v, runs[0][2])


def test_chat_greedy_token_exact_with_margin(golden_dir):
    """Token-exact greedy decode with NO near-tie escape (north_star: "greedy text decode token-id exact").  The fixture
    chat_real2_margin was produced by the reference's chat_with_recon (g2vlm.py:1305-1410, loop :1086-1137) at real widths
    with lm_head rows rescaled (oracle/synth.py::peaked_lm_head) so that the reference's own top-1 / top-2 logit gap is
    >= 4 bf16 ulp at every one of its 71 steps: an engine whose logits are within bf16 rounding of the reference's must
    return exactly its ids - through the batch-1 step (hipGraph replay and eager), the batched step and the continuous
    batching path."""
    meta, g = load(golden_dir, "chat_real2_margin")
    assert meta["min_margin_ulp"] >= 4.0 and meta["distinct_ids"] >= 8
    dims = meta["dims"]
    from g2vlm_amd.g2vlm_utils import build_model, configs_from_dims
    sd = synth.peaked_lm_head(synth.synth_state_dict(dims, seed=meta["seed"]), meta["head_sigma"], meta["head_seed"])
    model = build_model(*configs_from_dims(dims), sd, "cuda")
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    imgs = synth.synth_images(meta["n"], meta["h"], meta["w"], meta["seed"])
    ref = g["ref.ids"].tolist()
    assert len(ref) >= 63

    def vit_inputs():
        out = []
        for i in range(meta["n"]):
            gen = torch.Generator(); gen.manual_seed(1234 + i)
            out.append(vit_patchify(torch.randn((1, 3, meta["vit_grid"][0] * 14, meta["vit_grid"][1] * 14), generator=gen)))
        return out

    def transform_over(queue):
        it = iter(queue)

        def image_transform(_imgs):
            pv, thw = next(it)
            return pv, torch.tensor([list(thw)])
        return image_transform

    # batch-1, public entry point: graph replay, then eager
    for use_graph in (True, False):
        model.use_decode_graph = use_graph
        got = []
        dec = tok.decode
        tok.decode = lambda ids: got.extend(int(v) for v in ids) or ""
        model.chat_with_recon(tok, tok.new_token_ids, transform_over(vit_inputs()), None, images=imgs, prompt=meta["prompt"],
                              max_length=meta["max_length"])
        tok.decode = dec
        assert got == ref, (use_graph, next(i for i, (a, b) in enumerate(zip(got, ref)) if a != b))
        # logits of the last step agree with the reference's to bf16 rounding noise (the margin is what makes ids robust)
    model.use_decode_graph = True

    def scene(prompt):
        return lambda: model._chat_prefill(tok, tok.new_token_ids, transform_over(vit_inputs()), None, imgs, prompt)

    eos = tok.new_token_ids["eos_token_id"]
    # batched step: the golden scene twice around a different one
    pairs = [scene(p)() for p in (meta["prompt"], meta["prompt"] + " and how wide is the door", meta["prompt"])]
    outs = model.generate_text_batch([p for p, _ in pairs], [gi for _, gi in pairs], meta["max_length"], end_token_id=eos)
    for j in (0, 2):
        assert outs[j][1:, 0].tolist() == ref, j
    # continuous batching: three golden scenes through two slots (the third enters a slot mid-stream)
    kv_max = pairs[0][0].length + 8
    outs = model.generate_text_stream([scene(meta["prompt"])] * 3, max_batch=2, max_length=meta["max_length"], max_kv_len=kv_max,
                                      end_token_id=eos, chunk=8)
    for j in range(3):
        assert outs[j][1:, 0].tolist() == ref, j


def test_generate_text_do_sample(golden_dir):
    """generate_text(do_sample=True, temperature) (reference g2vlm.py:1119-1122): valid ids, the call is reproducible for a
    re-seeded model, different seeds draw different sequences, graph replay == eager (the sampler state lives on the
    device), and a non-positive temperature is refused BEFORE the prefill."""
    meta, g = load(golden_dir, "chat_tiny")
    dims = meta["dims"]
    model, sd = build(dims, meta["seed"])
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    imgs = synth.synth_images(meta["n"], meta["h"], meta["w"], meta["seed"])

    def prefill():
        vit_inputs = []
        for i in range(meta["n"]):
            gen = torch.Generator(); gen.manual_seed(1234 + i)
            vit_inputs.append(vit_patchify(torch.randn((1, 3, meta["vit_grid"][0] * 14, meta["vit_grid"][1] * 14), generator=gen)))
        it = iter(vit_inputs)

        def image_transform(_imgs):
            pv, thw = next(it)
            return pv, torch.tensor([list(thw)])
        return model._chat_prefill(tok, tok.new_token_ids, image_transform, None, imgs, meta["prompt"])

    def run(seed, temperature, use_graph, do_sample=True):
        model.use_decode_graph, model.sample_seed = use_graph, seed
        past, gi = prefill()
        return model.generate_text(past_key_values=past, max_length=16, do_sample=do_sample, temperature=temperature, end_token_id=None,
                                   **gi)[:, 0].tolist()

    a = run(5, 1.0, True)
    assert len(a) == 16 and all(0 <= t < dims["llm"]["vocab"] for t in a)
    assert run(5, 1.0, True) == a, "same seed, same draws"
    assert run(5, 1.0, False) == a, "graph replay and eager sampling disagree"
    assert run(6, 1.0, True) != a, "a different seed should draw a different sequence"
    # (temperature -> 0 recovering argmax is checked at the kernel level on logits with a clear margin: this model's
    #  random-weight logits hold exact top-2 ties, which any noise breaks either way)
    model.use_decode_graph = True
    with pytest.raises(ValueError):
        model.chat_with_recon(tok, tok.new_token_ids, None, None, images=None, prompt="x", max_length=4, do_sample=True, temperature=0.0)
