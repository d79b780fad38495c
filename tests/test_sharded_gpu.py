"""GPU: the view-sharded reconstruction (BASELINE C4 design, g2vlm_amd/sharded.py) reproduces the unsharded engine.

W ranks are simulated as W threads on the one available GPU (ThreadSimComm): same kernels, same per-rank row
subsets, collectives replaced by rendezvous copies.  Differences can only come from fp32 summation order - the
attention's stream-K split points (different Lq per rank) and, when a rank holds <= 64 rows, the split-K skinny GEMM
instead of the tiled one - i.e. bf16-ulp noise.  The point maps pass that noise through exp(z_raw) (|z| up to 1e4 on
random weights), so they are compared per point (median / 90th percentile of the relative error); a sharding bug
(wrong K/V rows, a missed exchange) shows up as O(1) errors in every point."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dims as D, synth  # noqa: E402  (checker/inputs only)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def point_q(a, b):
    a, b = a.double().cpu().reshape(-1, 3), b.double().cpu().reshape(-1, 3)
    e = (a - b).norm(dim=1) / (b.norm(dim=1) + 1e-30)
    return float(e.quantile(0.5)), float(e.quantile(0.9))


def check_maps(got, ref, tag):
    if got.dim() == 5 and got.shape[-1] == 3:
        q50, q90 = point_q(got, ref)
        # two bf16 realisations of the same network differ by 1-2e-2 per point (tests/test_e2e_gpu.py: reference vs
        # full precision); measured sharded-vs-unsharded on MI355X: median 6-8e-3, p90 2e-2
        assert q50 < 2e-2 and q90 < 5e-2 and rel(got, ref) < 5e-2, (tag, q50, q90, rel(got, ref))
    else:
        e = rel(got, ref)
        assert e < 5e-3, (tag, e)


@pytest.mark.parametrize("world", [2, 4])
def test_view_sharded_matches_unsharded(world):
    from g2vlm_amd.g2vlm_utils import build_model, configs_from_dims
    from g2vlm_amd.sharded import run_thread_sim
    dims = D.TINY
    sd = synth.synth_state_dict(dims, seed=21)
    model = build_model(*configs_from_dims(dims), sd, "cuda")
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    imgs = synth.synth_images(4, 70, 98, 21)
    ref = model.recon(tok, tok.new_token_ids, None, imgs)
    res = run_thread_sim(model, world, tok, tok.new_token_ids, imgs, gather=True)
    for r in range(world):
        assert res[r]["view_range"] == (0, 4)
        for k in ("points", "local_points", "global_points", "camera_poses", "images"):
            assert res[r][k].shape == ref[k].shape, k
            check_maps(res[r][k], ref[k], (world, r, k))
    # the gathered KV cache of every rank equals the unsharded one (same rows, same kernels)
    from g2vlm_amd.modeling.g2vlm import NaiveCache
    past = NaiveCache(dims["llm"]["layers"], dims["llm"]["kv_heads"], "cuda")
    gi, nl, nr = model.prepare_prompts_addbos([0], [0], ["Reconstruct the 3D scene."], tok, tok.new_token_ids)
    past = model.forward_cache_update_text(past, **gi)
    gi, nl, nr = model.prepare_dino_images_pi3(nl, nr, imgs, None, tok.new_token_ids)
    past, _ = model.forward_cache_update_dino(past, **gi)
    for r in range(world):
        pk = res[r]["past_key_values"]
        assert pk.length == past.length
        # layer 0 K depends only on DINO + layer-0 projections: identical up to isolated 1-ulp flips (a rank with <= 64
        # rows runs the split-K skinny GEMM, the unsharded engine the tiled one)
        assert rel(pk.key_cache[0], past.key_cache[0]) < 2e-3
        last = dims["llm"]["layers"] - 1
        assert rel(pk.value_cache[last], past.value_cache[last]) < 5e-3


def test_views_without_gather_are_local_slices():
    from g2vlm_amd.g2vlm_utils import build_model, configs_from_dims
    from g2vlm_amd.sharded import run_thread_sim
    dims = D.TINY
    model = build_model(*configs_from_dims(dims), synth.synth_state_dict(dims, seed=22), "cuda")
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    imgs = synth.synth_images(4, 56, 112, 22)
    ref = model.recon(tok, tok.new_token_ids, None, imgs)
    res = run_thread_sim(model, 2, tok, tok.new_token_ids, imgs, gather=False)
    for r, (lo, hi) in enumerate(((0, 2), (2, 4))):
        assert res[r]["view_range"] == (lo, hi)
        check_maps(res[r]["points"], ref["points"][:, lo:hi], (r, "points"))
