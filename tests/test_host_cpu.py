"""CPU: host-side logic of the product (bookkeeping, image loading, config/key contract) against
the reference-generated fixtures, the C-ABI export list, and the 2-rank gloo rehearsal of the
multi-GPU plumbing.  No compute kernel is called here (there is no GPU in this container)."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch
from safetensors.torch import load_file

from g2vlm_amd import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NT = dict(bos_token_id=1, eos_token_id=2, start_of_image=3, end_of_image=4)


class Tok:
    def encode(self, text, add_special_tokens=False):
        return [5 + (b % 500) for b in text.encode()]


def test_prepare_image_tokens_matches_reference(golden_dir):
    g = load_file(os.path.join(golden_dir, "prepare_indexes.safetensors"))
    meta = json.load(open(os.path.join(golden_dir, "prepare_indexes.json")))
    for n in (1, 2, 8):
        for (h, w) in ((518, 518), (294, 518), (392, 518)):
            key = f"n{n}_{h}x{w}"
            m = meta[key]
            gi, newlen, new_rope = host.prepare_image_tokens(m["T0"], m["T0"], [(1, h // 14, w // 14)] * n, NT)
            assert torch.equal(gi["packed_position_ids"].to(torch.int32), g[key + ".packed_position_ids"])
            assert torch.equal(gi["packed_text_indexes"].to(torch.int32), g[key + ".packed_text_indexes"])
            assert newlen == m["newlens"] and new_rope == m["new_rope"]
            assert int(gi["packed_token_indexes"].sum()) == m["sum_dino_idx"] and gi["packed_token_indexes"].numel() == m["n_dino"]
            assert int(gi["packed_indexes"].sum()) == m["sum_indexes"]
            assert [int(x) for x in gi["packed_seqlens"]] == m["packed_seqlens"]
            assert [int(x) for x in gi["token_seqlens"]] == m["dino_token_seqlens"]
    # ViT bookkeeping on the merged 27x27 grid
    gi, newlen, new_rope = host.prepare_image_tokens(17, 23, [(1, 54, 54)], NT, merge=2)
    assert torch.equal(gi["packed_position_ids"].to(torch.int32), g["vit.packed_position_ids"])
    m = meta["vit"]
    assert (newlen, new_rope, gi["packed_token_indexes"].numel(), int(gi["packed_indexes"][0])) == (m["newlens"], m["new_rope"], m["n_tok"], m["first_idx"])


def test_prepare_text_variants():
    gi, nl, nr = host.prepare_text([3], [7], ["ab"], Tok(), NT, bos=True)
    assert gi["packed_text_ids"].tolist()[0] == 1 and gi["packed_text_position_ids"].shape == (3, 3)
    assert gi["packed_text_indexes"].tolist() == [3, 4, 5] and gi["packed_key_value_indexes"].tolist() == [0, 1, 2]
    assert nl == [6] and nr == [10]
    gi, _, _ = host.prepare_text([0], [0], ["ab"], Tok(), NT, bos=True, eos=True)
    assert gi["packed_text_ids"].tolist()[0] == 1 and gi["packed_text_ids"].tolist()[-1] == 2


def _synthetic_pair():
    from PIL import Image
    rng = np.random.RandomState(7)
    srcs = []
    for _ in range(2):
        base = rng.rand(27, 48, 3)
        img = np.kron(base, np.ones((20, 20, 1))) * 255
        img = np.clip(img + rng.randn(*img.shape) * 8, 0, 255).astype(np.uint8)
        srcs.append(Image.fromarray(img, "RGB"))
    return srcs


def test_load_and_resize14_matches_reference(golden_dir):
    g = load_file(os.path.join(golden_dir, "loader.safetensors"))
    out = host.load_and_resize14(_synthetic_pair(), 518)
    assert out.shape == (2, 3, 294, 518)
    assert torch.equal((out * 255).round().to(torch.uint8), g["loader.out_u8"])


@pytest.mark.parametrize("h,w,oh,ow", [(720, 1280, 294, 518), (540, 960, 294, 518), (100, 130, 294, 518), (300, 518, 294, 518),
                                        (294, 400, 294, 518), (64, 64, 70, 98), (37, 53, 14, 14)])
def test_lanczos_restatement_matches_pillow(h, w, oh, ow):
    """host.lanczos_tables / lanczos_resize_u8_reference (the integer restatement the device kernel executes) against
    Pillow's own Image.resize(..., LANCZOS) - the call the reference's loader makes (data/transforms_vggt.py:437): bit-exact
    for down-scaling, up-scaling, one-axis resizes, on noise and on a smooth image."""
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(h * 7 + w)
    src = rng.integers(0, 256, size=(2, h, w, 3), dtype=np.uint8)
    src[1] = np.clip(np.kron(rng.random((h // 20 + 1, w // 20 + 1, 3)), np.ones((20, 20, 1)))[:h, :w] * 255, 0, 255).astype(np.uint8)
    mine = host.lanczos_resize_u8_reference(torch.from_numpy(src), oh, ow)
    ref = np.stack([np.asarray(Image.fromarray(s).resize((ow, oh), Image.Resampling.LANCZOS)) for s in src])
    assert np.array_equal(mine.numpy(), ref)
    b, k = host.lanczos_tables(w, ow)
    assert b.dtype == torch.int32 and k.dtype == torch.int32 and b.shape == (ow, 2) and int((b[:, 0] + b[:, 1]).max()) <= w
    assert bool(((k.sum(1) - (1 << host.PIL_PRECISION_BITS)).abs() <= k.shape[1]).all())      # taps sum to one in 22-bit fixed point


def test_load_and_resize16_matches_reference(golden_dir):
    """the use_dinov3 variant's loader (reference data/transforms_vggt.py:464-471): LANCZOS to 294x518, then the antialiased
    bilinear resize to 288x512 - bit-exact against the reference's own output"""
    g = load_file(os.path.join(golden_dir, "loader.safetensors"))
    out = host.load_and_resize16(_synthetic_pair(), 518)
    assert out.shape == (2, 3, 288, 512)
    assert torch.equal(out, g["loader16.out"])


def test_qwenvl2_image_transform_matches_reference(golden_dir):
    g = load_file(os.path.join(golden_dir, "loader.safetensors"))
    meta = json.load(open(os.path.join(golden_dir, "loader.json")))["vitproc"]
    pv, thw = host.QwenVL2ImageTransform(768, 768, 14)([_synthetic_pair()[0]])
    assert list(pv.shape) == meta["shape"] and thw.tolist() == [meta["grid"]]
    assert (pv[::37] - g["vitproc.pixel_values_sub"]).abs().max() < 2e-6
    assert abs(float(pv.double().sum()) - meta["checksum"]) < 1e-2 * max(1.0, abs(meta["checksum"])) * 1e-2 + 1.0


def test_smart_resize_cases():
    assert host.smart_resize(768, 768) == (756, 756)
    assert host.smart_resize(968, 1296) == (840, 1148)      # > max_pixels: beta path
    with pytest.raises(ValueError):
        host.smart_resize(10, 800)


def test_state_dict_key_contract_matches_oracle_and_configs():
    from g2vlm_amd.synthetic import REAL_DIMS, param_shapes
    from g2vlm_amd.g2vlm_utils import configs_from_dims
    from g2vlm_amd.modeling.g2vlm.g2vlm import dims_from_configs
    from oracle import dims as D, synth
    for mine, theirs in ((REAL_DIMS, D.REAL), (D.TINY, D.TINY)):
        a, b = param_shapes(mine), synth.param_shapes(theirs)
        assert a == {k: tuple(v) for k, v in b.items()}
    llm, vit, dino = configs_from_dims(REAL_DIMS)
    d = dims_from_configs(llm, vit, dino)
    assert d["llm"] == REAL_DIMS["llm"] and d["dino"] == REAL_DIMS["dino"] and d["vit"] == REAL_DIMS["vit"]


def test_interleave_gate_up_layout():
    from g2vlm_amd.weights import interleave_gate_up, pad_k
    g, u = torch.arange(64.).view(32, 2), -torch.arange(64.).view(32, 2)
    w = interleave_gate_up(g, u)
    assert torch.equal(w[:16], g[:16]) and torch.equal(w[16:32], u[:16]) and torch.equal(w[32:48], g[16:])
    assert pad_k(torch.ones(3, 588)).shape == (3, 640) and float(pad_k(torch.ones(3, 588))[:, 588:].abs().sum()) == 0


def test_library_exports_every_declared_symbol():
    from g2vlm_amd import build, hip
    path = build.build()
    lib = ctypes.CDLL(path)
    header = open(os.path.join(ROOT, "include", "g2vlm_hip.h")).read()
    declared = set(re.findall(r"\b(g2v_[a-z0-9_]+)\s*\(", header)) - {"g2v_gemm_desc", "g2v_gemm_group", "g2v_attn_tile"}
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/g2vlm_hip.h but not exported"
    assert set(hip.EXPORTS) <= declared
    lib.g2v_arch.restype = ctypes.c_char_p
    assert lib.g2v_arch() == b"gfx950" and lib.g2v_version() == 1


def test_product_never_imports_oracle():
    """The product - the package, the top-level drop-in names (g2vlm_utils, modeling, data) and the entry scripts - never
    imports the oracle; bench.py may, in its cpu_baseline leg only (checked by its own structure: the import sits inside
    cpu_baseline())."""
    srcs = [os.path.join(ROOT, f) for f in ("g2vlm_utils.py", "inference_recon.py", "inference_chat.py")]
    for top in ("g2vlm_amd", "modeling", "data"):
        for dp, _, files in os.walk(os.path.join(ROOT, top)):
            srcs += [os.path.join(dp, f) for f in files if f.endswith(".py")]
    assert len(srcs) > 15
    for path in srcs:
        src = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{path} imports the oracle"


def test_ply_writer(tmp_path):
    p = tmp_path / "a.ply"
    host.write_ply_binary(str(p), np.arange(12, dtype=np.float32).reshape(4, 3), np.full((4, 3), 0.5))
    raw = p.read_bytes()
    hdr, body = raw.split(b"end_header\n")
    assert b"element vertex 4" in hdr and len(body) == 4 * 15


def test_shard_views():
    from g2vlm_amd.dist_util import shard_views
    assert [shard_views(32, 8, r) for r in range(8)] == [(4 * r, 4 * r + 4) for r in range(8)]
    parts = [shard_views(10, 4, r) for r in range(4)]
    assert parts[0][0] == 0 and parts[-1][1] == 10 and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))


_GLOO_WORKER = r'''
import os, sys, time
sys.path.insert(0, os.environ["G2V_ROOT"])
import torch.distributed as dist
from g2vlm_amd import dist_util as du
world, rank, local = du.init("gloo")
assert world == 2
lo, hi = du.shard_views(8, world, rank)
du.barrier(sync_cuda=False)
secs = 0.5 + rank            # rank 1 is the slow one
mx = du.max_over_ranks(secs)
val = du.aggregate_throughput(hi - lo, secs)
assert abs(mx - 1.5) < 1e-9 and abs(val - 2 * 4 / 1.5) < 1e-9, (mx, val)
du.barrier(sync_cuda=False)
dist.destroy_process_group()
open(os.path.join(os.environ["G2V_OUT"], f"ok{rank}"), "w").write("ok")
'''


_GLOO_COMM_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["G2V_ROOT"])
import torch, torch.distributed as dist
from g2vlm_amd import dist_util as du
from g2vlm_amd.sharded import TorchDistComm
world, rank, local = du.init("gloo")
comm = TorchDistComm()
# K/V-block exchange as in recon_view_sharded: every rank owns one contiguous block of the cache slice
blk = 5
full = torch.zeros((world * blk, 2, 4))
full[rank * blk:(rank + 1) * blk] = rank + 1
comm.all_gather_blocks(full, blk)
for r in range(world):
    assert bool((full[r * blk:(r + 1) * blk] == r + 1).all()), full
ctx = torch.full((3, 4), 7.0) if rank == 0 else torch.zeros((3, 4))
comm.broadcast(ctx, 0)
assert bool((ctx == 7).all())
comm.barrier()
dist.destroy_process_group()
open(os.path.join(os.environ["G2V_OUT"], f"ok{rank}"), "w").write("ok")
'''


@pytest.mark.timeout(180)
def test_two_rank_gloo_kv_block_exchange(tmp_path):
    """TorchDistComm (the RCCL path's wrapper) rehearsed on CPU with gloo: K/V block all-gather + context broadcast."""
    script = tmp_path / "c.py"
    script.write_text(_GLOO_COMM_WORKER)
    env = dict(os.environ, G2V_ROOT=ROOT, G2V_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29542", str(script)], env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


@pytest.mark.timeout(180)
def test_two_rank_gloo_replica_timing(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, G2V_ROOT=ROOT, G2V_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", str(script)], env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def _reference_greedy(seqs, max_length, eos):
    """The reference loop (g2vlm.py:1088-1135) per scene: append the current id, step, stop when the NEW id is EOS."""
    outs = []
    for seq in seqs:
        out, i = [], 0
        while len(out) < max_length:
            out.append(seq[i]); i += 1
            if eos is not None and seq[i] == eos:
                break
        outs.append(out)
    return outs


@pytest.mark.parametrize("eos_at", [(None, None), (1, 9), (8, 3), (16, None), (5, 5), (40, 41)])
@pytest.mark.parametrize("max_length", [1, 7, 8, 20])
def test_greedy_loop_chunked_eos_matches_reference_loop(eos_at, max_length):
    """G2VLM._greedy_loop launches steps in chunks and looks for EOS once per chunk; its output must be what the
    reference's step-by-step loop yields, for EOS on, before and after chunk boundaries, never, and beyond max_length."""
    from g2vlm_amd.modeling.g2vlm.g2vlm import G2VLM
    EOS, B = 99, len(eos_at)
    seqs = []
    for j, e in enumerate(eos_at):
        s = [10 * (j + 1) + (i % 7) for i in range(64)]
        if e is not None:
            s[e] = EOS
        seqs.append(s)
    for eos in (EOS, None):
        tok = torch.tensor([s[0] for s in seqs], dtype=torch.int32)
        state = {"i": 0, "calls": 0}

        def step():
            state["i"] += 1; state["calls"] += 1
            tok.copy_(torch.tensor([s[state["i"]] for s in seqs], dtype=torch.int32))
            return tok
        got = G2VLM._greedy_loop(None, step, tok, B, max_length, eos)
        assert got == _reference_greedy(seqs, max_length, eos)
        assert state["calls"] <= max_length


# ----------------------------------------------------------------------------- attention schedule (host-built tables)
def _decode_plan(plan):
    import numpy as np
    tiles = plan.tiles.cpu().numpy().reshape(-1, 8)
    out = []
    for segs, ptr, n_blocks in plan.phases:
        sg = segs.cpu().numpy().reshape(-1, 8)
        pt = ptr.cpu().numpy()
        assert len(pt) == n_blocks + 1 and pt[0] == 0 and (np.diff(pt) >= 0).all()
        out.append((sg[:pt[-1]], pt))
    comb = plan.comb.cpu().numpy().reshape(-1, 4)[:plan.n_comb]
    return tiles, out, comb


@pytest.mark.parametrize("windows,Hq,tile_rows,max_blocks", [
    ([(0, 10968, 0, 10976, False)], 12, 256, None),                       # C3 MoT prefill: 516 equal items on 256 workgroups
    ([(0, 43872, 0, 43880, False)], 12, 256, None),                       # C4 unsharded
    ([(i * 1369, 1369, i * 1369, 1369, False) for i in range(8)], 16, 128, None),   # DINO windows: short items
    ([(0, 731, 0, 15000, False)], 12, 128, None),                         # ViT-token prefill: fewer items than workgroups
    ([(0, 300, 0, 900, True)], 4, 128, 7),                                # causal: unequal items, stream-K cuts
    ([(0, 40, 0, 300, True), (40, 200, 300, 200, False)], 2, 128, 5),
])
def test_attention_plan_covers_every_kv_tile_once_and_balances(windows, Hq, tile_rows, max_blocks):
    """make_attn_plan: every (descriptor, head) item's 64-key tiles are covered exactly once by the segments; a segment
    writes the output directly (slot < 0) only if it is its output tile's only piece; merge entries list each remaining
    output tile's consecutive slots; workgroup loads differ by little."""
    import numpy as np
    from g2vlm_amd import hip
    plan = hip.make_attn_plan(windows, Hq, "cpu", tile_rows=tile_rows, max_blocks=max_blocks)
    tiles, phases, comb = _decode_plan(plan)
    assert len(phases) == 1
    sg, pt = phases[0]
    nkt = {}
    for d, t in enumerate(tiles):
        q0, qr, k0, kl, shift = t[:5]
        need = min(kl, (q0 - t[5]) + qr - 1 + shift + 1) if shift < 2 ** 29 else kl
        nkt[d] = (need + 63) // 64
    cover, slots = {}, {}
    for s_ in sg:
        d, h, kt0, kt1, slot = (int(v) for v in s_[:5])
        assert 0 <= kt0 < kt1 <= nkt[d]
        cover.setdefault((d, h), []).append((kt0, kt1, slot))
    assert len(cover) == len(tiles) * Hq
    used = set()
    for (d, h), pcs in cover.items():
        pcs.sort()
        assert pcs[0][0] == 0 and pcs[-1][1] == nkt[d] and all(a[1] == b[0] for a, b in zip(pcs, pcs[1:])), (d, h, pcs)
        if len(pcs) == 1:
            assert pcs[0][2] == -1
        else:
            for p_ in pcs:
                assert p_[2] >= 0 and p_[2] not in used
                used.add(p_[2])
            slots[(d, h)] = sorted(p_[2] for p_ in pcs)
    assert used == set(range(plan.n_slots))
    assert len(comb) == len(slots)
    for d, h, s0, n in comb:
        assert slots[(int(d), int(h))] == list(range(int(s0), int(s0 + n)))
    load = np.zeros(len(pt) - 1)
    for b in range(len(pt) - 1):
        load[b] = sum(int(s_[3] - s_[2]) for s_ in sg[pt[b]:pt[b + 1]])
    assert load.sum() == sum(nkt[d] for d in nkt) * Hq
    if len(windows) == 1 and not windows[0][4]:
        assert load.max() <= 1.03 * load.mean() + 1, (load.max(), load.mean())     # long equal items: balanced within 3 %
    if windows == [(0, 10968, 0, 10976, False)]:
        assert plan.n_slots <= 160, plan.n_slots             # the first form of the schedule left 512 partial slots here


def test_attention_plan_phases_share_output_tiles():
    """The view-sharded prefill's two launches: phase 0 = the rank's own K/V block, phase 1 = prefix + the other ranks'
    blocks.  Every output tile is merged from the slots of all its descriptors; nothing is written directly."""
    from g2vlm_amd import hip
    T0, blk, world, rank = 8, 2742, 4, 1
    Lq, tot = blk, T0 + world * blk
    lo = T0 + rank * blk
    wins = [(0, Lq, lo, blk, False, 0), (0, Lq, 0, lo, False, 1), (0, Lq, lo + blk, tot - lo - blk, False, 1)]
    plan = hip.make_attn_plan(wins, 12, "cpu", tile_rows=256)
    tiles, phases, comb = _decode_plan(plan)
    assert len(phases) == 2 and plan.n_comb == 11 * 12         # 11 query tiles of 256 rows x 12 heads
    for sg, pt in phases:
        assert (sg[:, 4] >= 0).all()
    n_slots = sum(int(c[3]) for c in comb)
    assert n_slots == plan.n_slots and all(int(c[3]) >= 3 for c in comb)
