"""GPU: the view-sharded reconstruction (BASELINE C4 design, g2vlm_amd/sharded.py) reproduces the unsharded engine.

W ranks are simulated as W threads on the one available GPU, every rank on its own stream (ThreadSimComm): same kernels,
same per-rank row subsets, collectives replaced by rendezvous copies - drained (device sync + host barrier) or
stream-ordered through events the way RCCL orders them (`overlap=True`: KVExchange's side-stream branch, the one the
RCCL path takes, runs under test).  The same forward also runs in 2 real PROCESSES through TorchDistComm / gloo
(test_two_process_gloo_*), both on the one GPU.  Differences can only come from fp32 summation order - the
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


def _tiny_scene(seed, n=4, h=70, w=98, dims=None):
    from g2vlm_amd.g2vlm_utils import build_model, configs_from_dims
    dims = dims or D.TINY
    model = build_model(*configs_from_dims(dims), synth.synth_state_dict(dims, seed=seed), "cuda")
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    return model, tok, synth.synth_images(n, h, w, seed), dims


@pytest.mark.parametrize("world", [2, 4])
def test_overlapped_kv_exchange_branch_is_bit_identical_to_the_drained_one(world):
    """ADVICE r02 / VERDICT r02 missing #1: KVExchange's side-stream branch (all-gather of layer i's K / V blocks on a
    communication stream behind the cache write, phase-0 attention over the local block meanwhile, wait_event, then the
    remote blocks and the merge) is what the RCCL path takes and had never executed.  Here the ranks are threads on their own
    streams and the collectives are STREAM-ORDERED (events only, no device synchronize), held back by a 2-million-cycle
    delay on the communication stream so that a missing join would read blocks that have not arrived.  Every layer must
    have gone through the side stream, and the result must equal the drained simulation bit for bit (same kernels, same
    plans, same phases - only the ordering mechanism differs) and the unsharded engine within bf16 noise."""
    from g2vlm_amd.sharded import run_thread_sim
    model, tok, imgs, dims = _tiny_scene(23)
    ref = model.recon(tok, tok.new_token_ids, None, imgs)
    drained = run_thread_sim(model, world, tok, tok.new_token_ids, imgs, gather=True)
    over = run_thread_sim(model, world, tok, tok.new_token_ids, imgs, gather=True, overlap=True, delay_cycles=2_000_000)
    nl = dims["llm"]["layers"]
    for r in range(world):
        assert drained[r]["kv_layers_overlapped"] == 0 and over[r]["kv_layers_overlapped"] == nl
        for k in ("points", "local_points", "global_points", "camera_poses", "images"):
            assert torch.equal(over[r][k], drained[r][k]), (r, k, rel(over[r][k], drained[r][k]))
            check_maps(over[r][k], ref[k], (world, r, k))
        for i in range(nl):
            assert torch.equal(over[r]["past_key_values"].key_cache[i], drained[r]["past_key_values"].key_cache[i]), (r, i)
            assert torch.equal(over[r]["past_key_values"].value_cache[i], drained[r]["past_key_values"].value_cache[i]), (r, i)


def test_kv_overlap_opt_out():
    from g2vlm_amd.sharded import ThreadSimComm, recon_view_sharded, run_thread_sim
    model, tok, imgs, dims = _tiny_scene(24)
    res = run_thread_sim(model, 2, None, None, None, overlap=True,
                         fn=lambda comm: recon_view_sharded(model, comm, tok, tok.new_token_ids, imgs, gather=False, kv_overlap=False))
    assert [r["kv_layers_overlapped"] for r in res] == [0, 0]


@pytest.mark.parametrize("name,world", [("recon_dinov3_tiny_2v_64x96", 2), ("recon_dinov3_real2_3v_80x64", 3)])
def test_dinov3_view_sharded_matches_unsharded(golden_dir, name, world):
    """VERDICT r02 missing #4: the use_dinov3 variant (reference g2vlm.py:169-172, 1172-1174; encoder
    modeling/dinov3/dinov3_model.py:304-314) through the view-sharded prefill.  Its windows are cumulative PATCH counts
    over [cls | R registers | patches] rows (hazard H1 again, prefix 1 + R instead of 5), so it is sharded by window with a
    (1 + R) * lo row boundary exchange.  One view per rank: every window straddles two ranks' views."""
    import json, os
    from g2vlm_amd.g2vlm_utils import build_model, configs_from_dims
    from g2vlm_amd.sharded import run_thread_sim
    meta = json.load(open(os.path.join(golden_dir, name + ".json")))
    dims = meta["dims"]
    model = build_model(*configs_from_dims(dims), synth.synth_state_dict(dims, seed=meta["seed"]), "cuda")
    assert model.use_dinov3
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    imgs = synth.synth_images(meta["n"], meta["h"], meta["w"], meta["seed"])
    ref = model.recon(tok, tok.new_token_ids, None, imgs)
    for overlap in (False, True):
        res = run_thread_sim(model, world, tok, tok.new_token_ids, imgs, gather=True, overlap=overlap)
        for r in range(world):
            for k in ("points", "local_points", "global_points", "camera_poses", "images"):
                check_maps(res[r][k], ref[k], (name, overlap, r, k))


def _chat_inputs(meta_n, grid):
    from oracle.g2vlm_oracle import vit_patchify
    out = []
    for i in range(meta_n):
        gen = torch.Generator(); gen.manual_seed(1234 + i)
        out.append(vit_patchify(torch.randn((1, 3, grid[0] * 14, grid[1] * 14), generator=gen)))
    return out


def _transform_over(queue):
    it = iter(queue)

    def image_transform(_imgs):
        pv, thw = next(it)
        return pv, torch.tensor([list(thw)])
    return image_transform


def _near_tie(logits_row, other_tok, ulps=2):
    import math
    top = float(logits_row.max())
    ulp = 2.0 ** (math.floor(math.log2(abs(top))) - 7)
    return top - float(logits_row[other_tok]) <= ulps * ulp


@pytest.mark.parametrize("world,overlap", [(2, False), (4, True)])
def test_chat_after_view_sharded_prefill_gives_the_unsharded_ids(world, overlap):
    """SURVEY 8f-3, second half (VERDICT r02 missing #2): chat_with_recon (reference g2vlm.py:1305-1410; batch-1 limits :1006,
    :1137) where the geometry prefill of the 4 views is sharded over the ranks.  The per-layer K/V all-gather leaves the
    full cache on every rank, so rank 0 runs the ViT stages, the question and the greedy decode as the unsharded method
    does and broadcasts the ids.  Real widths, 2 layers, lm_head rows spread (synth.peaked_lm_head) so that bf16 noise in
    the geo rows cannot flip an argmax: the ids must equal the unsharded chat's on every rank (a flip would have to be a
    near-tie of the unsharded logits), and the geo rows of the cache must equal the unsharded prefill's to bf16 noise."""
    from g2vlm_amd.g2vlm_utils import build_model, configs_from_dims
    from g2vlm_amd.sharded import chat_view_sharded, run_thread_sim
    dims = D.reduced(vocab=2048)
    sd = synth.peaked_lm_head(synth.synth_state_dict(dims, seed=41), 1.0, 19)
    model = build_model(*configs_from_dims(dims), sd, "cuda")
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    imgs = synth.synth_images(4, 70, 98, 41).cuda()
    prompt, steps, grid = "\nHow far is the chair?\nPlease answer the question using a single word or phrase.", 24, (8, 12)
    model.use_decode_graph = False            # (a graph capture on rank 0's thread would not tolerate the other threads' runtime calls)
    past, gi = model._chat_prefill(tok, tok.new_token_ids, _transform_over(_chat_inputs(4, grid)), None, imgs, prompt)
    geo_k0 = past.key_cache[0].clone()
    st = model.engine.decode_begin(past, int(gi["packed_start_tokens"][0]), int(gi["packed_query_position_ids"][0, 0]), steps, use_graph=False)
    want, lgs = [int(st["tok"][0])], []
    for _ in range(steps - 1):
        want.append(int(model.engine.decode_step(st)[0]))
        lgs.append(st["logits"].float().cpu().clone())
    res = run_thread_sim(model, world, None, None, None, overlap=overlap, fn=lambda comm: chat_view_sharded(
        model, comm, tok, tok.new_token_ids, _transform_over(_chat_inputs(4, grid)), imgs, prompt, steps, return_ids=True))
    for r in range(world):
        got = res[r].tolist()
        eos = tok.new_token_ids["eos_token_id"]
        cut = want[:want.index(eos)] if eos in want[1:] else want
        n = min(len(got), len(cut))
        fd = next((i for i in range(n) if got[i] != cut[i]), None)
        assert fd is None or _near_tie(lgs[fd - 1], got[fd]), (r, fd, got, cut)
        assert fd is not None or len(got) == len(cut), (r, got, cut)
    assert res[0].tolist() == res[-1].tolist()


_TWO_PROC_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["G2V_ROOT"])
import torch, torch.distributed as dist
from g2vlm_amd import dist_util as du
from g2vlm_amd.g2vlm_utils import build_model, configs_from_dims
from g2vlm_amd.sharded import TorchDistComm, recon_view_sharded, chat_view_sharded
from oracle import dims as D, synth                      # inputs / fake tokenizer only
from oracle.g2vlm_oracle import vit_patchify
torch.cuda.set_device(0)                                   # both ranks share the one GPU: gloo (RCCL refuses two ranks per device)
world, rank, local = du.init("gloo")
assert world == 2
comm = TorchDistComm()
assert not comm.overlappable
dims = D.reduced(vocab=2048)
sd = synth.peaked_lm_head(synth.synth_state_dict(dims, seed=41), 1.0, 19)
model = build_model(*configs_from_dims(dims), sd, "cuda")
tok = synth.FakeTokenizer(dims["llm"]["vocab"])
imgs = synth.synth_images(4, 70, 98, 41).cuda()
res = recon_view_sharded(model, comm, tok, tok.new_token_ids, imgs, gather=False)
out = {k: res[k].cpu() for k in ("points", "local_points", "global_points", "camera_poses")}
out["k0"] = res["past_key_values"].key_cache[0].cpu()
out["v_last"] = res["past_key_values"].value_cache[dims["llm"]["layers"] - 1].cpu()
out["view_range"] = torch.tensor(res["view_range"])
def vit_inputs():
    o = []
    for i in range(4):
        gen = torch.Generator(); gen.manual_seed(1234 + i)
        o.append(vit_patchify(torch.randn((1, 3, 8 * 14, 12 * 14), generator=gen)))
    return iter(o)
it = vit_inputs()
def tf(_):
    pv, thw = next(it)
    return pv, torch.tensor([list(thw)])
out["ids"] = chat_view_sharded(model, comm, tok, tok.new_token_ids, tf, imgs, os.environ["G2V_PROMPT"], 24, return_ids=True)
torch.save(out, os.path.join(os.environ["G2V_OUT"], f"rank{rank}.pt"))
comm.barrier()
dist.destroy_process_group()
'''


@pytest.mark.timeout(900)
def test_two_process_gloo_recon_and_chat_match_unsharded(tmp_path):
    """the multi-PROCESS path.  `python -m torch.distributed.run --nproc-per-node 2` starts two fresh
    processes that share the one GPU and talk gloo (RCCL refuses two ranks per device); each runs recon_view_sharded and
    chat_view_sharded through TorchDistComm - the communicator class the RCCL path uses, list-form collectives staged through
    host memory - on a 4-view scene at real widths (2 + 2 layers).  Their outputs, written to files, must match the unsharded
    engine of THIS process: point maps / poses / cache rows within bf16 noise, the chat ids exactly (near-tie rule)."""
    import os, subprocess, sys
    from g2vlm_amd.g2vlm_utils import build_model, configs_from_dims
    from g2vlm_amd.modeling.g2vlm import NaiveCache
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prompt = "\nHow far is the chair?\nPlease answer the question using a single word or phrase."
    script = tmp_path / "w.py"
    script.write_text(_TWO_PROC_WORKER)
    env = dict(os.environ, G2V_ROOT=root, G2V_OUT=str(tmp_path), G2V_PROMPT=prompt, MASTER_ADDR="127.0.0.1")
    torch.cuda.synchronize()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29547", str(script)], env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    dims = D.reduced(vocab=2048)
    sd = synth.peaked_lm_head(synth.synth_state_dict(dims, seed=41), 1.0, 19)
    model = build_model(*configs_from_dims(dims), sd, "cuda")
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    imgs = synth.synth_images(4, 70, 98, 41).cuda()
    ref = model.recon(tok, tok.new_token_ids, None, imgs)
    past = NaiveCache(dims["llm"]["layers"], dims["llm"]["kv_heads"], "cuda")
    gi, nl, nr = model.prepare_prompts_addbos([0], [0], ["Reconstruct the 3D scene."], tok, tok.new_token_ids)
    past = model.forward_cache_update_text(past, **gi)
    gi, nl, nr = model.prepare_dino_images_pi3(nl, nr, imgs, None, tok.new_token_ids)
    past, _ = model.forward_cache_update_dino(past, **gi)
    model.use_decode_graph = False
    pst, sgi = model._chat_prefill(tok, tok.new_token_ids, _transform_over(_chat_inputs(4, (8, 12))), None, imgs, prompt)
    st = model.engine.decode_begin(pst, int(sgi["packed_start_tokens"][0]), int(sgi["packed_query_position_ids"][0, 0]), 24, use_graph=False)
    want, lgs = [int(st["tok"][0])], []
    for _ in range(23):
        want.append(int(model.engine.decode_step(st)[0]))
        lgs.append(st["logits"].float().cpu().clone())
    eos = tok.new_token_ids["eos_token_id"]
    cut = want[:want.index(eos)] if eos in want[1:] else want
    last = dims["llm"]["layers"] - 1
    for rank, (lo, hi) in enumerate(((0, 2), (2, 4))):
        out = torch.load(tmp_path / f"rank{rank}.pt", weights_only=True)
        assert tuple(out["view_range"].tolist()) == (lo, hi)
        for k in ("points", "local_points", "global_points", "camera_poses"):
            check_maps(out[k], ref[k][:, lo:hi], ("2proc", rank, k))
        assert rel(out["k0"], past.key_cache[0]) < 2e-3 and rel(out["v_last"], past.value_cache[last]) < 1e-2
        got = out["ids"].tolist()
        fd = next((i for i in range(min(len(got), len(cut))) if got[i] != cut[i]), None)
        assert fd is None or _near_tie(lgs[fd - 1], got[fd]), (rank, fd, got, cut)
        assert fd is not None or len(got) == len(cut)
