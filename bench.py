#!/usr/bin/env python
"""Headline benchmark: multi-view reconstruction views/sec of G2VLM-2B-MoT on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the reconstruction hot path over one synthetic 8-view 518x518 scene
(BASELINE.json configs[2], "C3"): text prefix prefill -> DINOv2-L encoder -> 28-layer MoT geo
prefill with global cross-view attention -> 3 Pi3 decoders -> pointmap / camera heads, i.e.
G2VLM.forward_cache_update_text + forward_cache_update_dino + reconstruct at full depth and
full width on random-init weights (no checkpoint is reachable offline).  Images are resident
in HBM when the timed region starts.  With N > 1 every rank reconstructs its own scene
(replicas, SURVEY.md §8e: "throughput scaling at 2/4/8 GPUs for C3: replicas only"), so
`scaling` is weak and `value` is the whole-job aggregate.

Rank 0 prints ONE JSON line; `roofline` prices the dominant kernel (MoT flash attention) from
HIP-event timing of that kernel at the workload's exact shapes; `cpu_baseline` is the CPU
oracle timed on this host on a bounded sample (N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_VIEWS, HW, T0 = 8, 518, 8
PEAK_BF16_TFLOPS = 2500.0           # dense, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
PEAK_HBM_GBPS = 8000.0              # spec, MI355X_MICROARCH.md "HBM3E peak BW" (6.29 TB/s measured achievable)
# HBM-side bytes of ONE MoT attention launch from the rocprofv3 PMC passes (profiles/): 2 x FETCH_SIZE (gfx950 reports
# half of wide coalesced reads, MI355X_MICROARCH.md "HBM") + WRITE_SIZE.  None until a PMC pass has been committed.
TRAFFIC_BYTES_PER_LAUNCH = 223.2e6
TRAFFIC_NOTE = ("profiles/r03q_attn_pmc.md (flash_fwd64_kernel): forward 2 x 77.9 MB FETCH + 50.5 MB WRITE, combine 2 x 8.3 + 0.3 MB; algorithmic "
                "78.6 MB (the surplus is K/V re-fetched per XCD through the fabric, served by the Infinity Cache, and 17 MB of partials); "
                "PMC MFMA busy 60.2 %")


class _Tok:
    """Fixed 7-id prompt (+bos = T0 = 8 prefix tokens); real tokenizer files are not available offline."""
    eos_token_id = 2

    def encode(self, text, add_special_tokens=False):
        return [11, 12, 13, 14, 15, 16, 17]


NEW_TOKEN_IDS = dict(bos_token_id=1, eos_token_id=2, start_of_image=3, end_of_image=4)


def flops_per_scene(dims, n, p, t0):
    """Algorithmic FLOPs: 2MNK per GEMM, 4 Lq Lk H d per attention."""
    L, D = dims["llm"], dims["dino"]
    H, F, nq, nkv = L["hidden"], L["ffn"], L["heads"] * 128, L["kv_heads"] * 128
    lq, lk = n * (p + 2), t0 + n * (p + 2)
    mot_lin = L["layers"] * lq * (2 * H * (nq + 2 * nkv) + 2 * nq * H + 3 * 2 * H * F)
    mot_att = L["layers"] * 4 * lq * lk * nq
    dh, s = D["hidden"], p + 5
    dino = D["layers"] * n * (s * (4 * 2 * dh * dh + 2 * 2 * dh * 4 * dh) + 4 * p * p * dh) + n * p * 2 * 588 * dh
    blk = 2 * H * 3 * H + 2 * H * H + 2 * 2 * H * 4 * H
    dec = n * p * 5 * (3 * blk + 4 * 2 * H * H) + 5 * n * 4 * p * p * H * 4 + n * p * 2 * H * (1024 + 512 + 1024)
    heads = n * p * (2 * 2 * 1024 * 588 + 6 * 2 * 512 * 512)
    return dict(total=mot_lin + mot_att + dino + dec + heads + n * p * 2 * dh * H, mot_attention=mot_att,
                mot_attention_per_launch=4 * lq * lk * nq)


def cpu_baseline(dims):
    """CPU oracle (oracle/g2vlm_oracle.py, kind "port") on this host: one DINO layer, one MoT layer, one block
    of each decoder and the heads at the C3 shapes, extrapolated linearly in depth."""
    from oracle import synth
    from oracle.g2vlm_oracle import NaiveCache, OracleG2VLM
    # this box's CPU share for one GPU is 16 cores; more threads than that only oversubscribe
    cores = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(cores)
    d = {"llm": dict(dims["llm"], layers=1, vocab=2048), "dino": dict(dims["dino"], layers=1),
         "vit": dict(dims["vit"], depth=0), "dec": dict(dims["dec"])}
    shapes = {k: v for k, v in synth.param_shapes(d).items() if ".blocks." not in k or ".blocks.0." in k}
    sd = synth.synth_state_dict(d, seed=0, jitter=False, shapes=shapes)
    orc = OracleG2VLM(sd, d)
    tok = _Tok()
    imgs = synth.synth_images(N_VIEWS, HW, HW, 0)
    NS = 2                                    # per-view-independent stages are timed on NS views and scaled by N_VIEWS/NS
    t = {}
    c = NaiveCache(1)
    gi, nl, nr = orc.prepare_prompts([0], [0], ["x"], tok, NEW_TOKEN_IDS, bos=True)
    orc.forward_cache_update_text(c, **gi)
    gi, nl, nr = orc.prepare_dino_images(nl, nr, imgs, NEW_TOKEN_IDS)
    cu = torch.nn.functional.pad(torch.cumsum(gi["dino_token_seqlens"], 0), (1, 0))
    sc = N_VIEWS / NS
    t0 = time.perf_counter(); orc.dino_forward(gi["packed_dino_images"][:NS], cu[:NS + 1], num_layers=0); t["dino_embed"] = (time.perf_counter() - t0) * sc
    t0 = time.perf_counter(); orc.dino_forward(gi["packed_dino_images"][:NS], cu[:NS + 1], num_layers=1)
    t["dino_layer"] = (time.perf_counter() - t0) * sc - t["dino_embed"]
    x = torch.randn((int(gi["packed_seqlens"].sum()), d["llm"]["hidden"]))
    t0 = time.perf_counter()
    last = orc.llm_forward_inference(x, gi["packed_seqlens"], gi["packed_position_ids"], gi["packed_indexes"], c, gi["key_values_lens"],
                                     gi["packed_key_value_indexes"], False, "geo", gi["packed_dino_token_indexes"], gi["packed_text_indexes"], 1)
    t["mot_layer"] = time.perf_counter() - t0
    p = (HW // 14) ** 2
    hidden = last[gi["packed_dino_token_indexes"]].reshape(N_VIEWS, p, -1)[:NS]
    pos = torch.cartesian_prod(torch.arange(HW // 14), torch.arange(HW // 14)).view(1, p, 2).expand(NS, -1, 2).clone()
    t0 = time.perf_counter()
    ph = orc.decoder("point_decoder", hidden, pos, depth=1); ch = orc.decoder("camera_decoder", hidden, pos, depth=1)
    gh = orc.decoder("global_points_decoder", hidden, pos, context=hidden[0:1].repeat(NS, 1, 1), depth=1)
    t["dec_blocks"] = (time.perf_counter() - t0) * sc
    t0 = time.perf_counter()
    orc.pts_head("point_head", ph.float(), HW, HW); orc.camera_head(ch.float()); orc.pts_head("global_point_head", gh.float(), HW, HW)
    t["heads"] = (time.perf_counter() - t0) * sc
    total = t["dino_embed"] + dims["dino"]["layers"] * t["dino_layer"] + dims["llm"]["layers"] * t["mot_layer"] + 5 * t["dec_blocks"] + t["heads"]
    return dict(value=N_VIEWS / total, unit="views/s", cores=cores, kind="port",
                sample=("C3 shapes: 1 MoT geo layer at the full 8-view sequence (Lq 10968), and 1 DINO layer, 1 block per decoder "
                        "and the heads on 2 of the 8 views (per-view independent, scaled x4); extrapolated linearly to 24/28/5 "
                        "layers; seconds per 8-view scene: " + json.dumps({k: round(v, 3) for k, v in t.items()})))


def host_prep_ms(model, n_views, reps=3):
    """host image preparation is excluded from views/s and reported separately.  n_views synthetic 1280x720
    RGB frames (no image files exist offline) go through the product's own front end - pinned upload of the decoded uint8
    frames, Pillow's LANCZOS resize to width 518 on the device (host.load_images_u8(device=), g2v_lanczos_resize_u8, bit-exact
    with the PIL call of reference data/transforms_vggt.py:411-451), device-side ToTensor / normalise (g2v_dino_preprocess)
    - and the wall time per scene is returned, device drained."""
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(0)
    frames = [Image.fromarray(rng.integers(0, 256, size=(720, 1280, 3), dtype=np.uint8)) for _ in range(n_views)]
    tok = _Tok()
    gi_text, nl, nr = model.prepare_prompts_addbos([0], [0], ["Reconstruct the 3D scene."], tok, NEW_TOKEN_IDS)
    best = None
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model.prepare_dino_images_pi3(nl, nr, frames, None, NEW_TOKEN_IDS)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None else min(best, dt)
    return round(best, 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decode-tokens", type=int, default=128)
    ap.add_argument("--overlap", type=int, default=2, help="also report (secondary key, outside the timed region) views/s with scenes issued on this many streams; 1 = skip")
    ap.add_argument("--breakdown", action="store_true", help="c5: print a per-stage wall-time breakdown of one step to stderr")
    ap.add_argument("--scenes-per-step", type=int, default=2, help="c5: scenes per rank and step (their greedy decodes run as one batch)")
    ap.add_argument("--workload", choices=["c3", "c4", "c5"], default="c3",
                    help="c3 (default, the headline): one 8-view scene per GPU, replicas.  c4: ONE scene of 4 views per GPU "
                         "sharded by view with an RCCL K/V all-gather per MoT layer (BASELINE config 4 at --gpus 8).  c5: "
                         "interleaved recon + chat, one scene per rank and step: 8-view pointmaps, then chat_with_recon over the "
                         "same 8 views (8 ViT images, 32-token question, 256 greedy tokens) (BASELINE config 5, replicas)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs for a box with fewer GPUs than ranks (never set by the driver): G2V_BENCH_DEVICE pins every rank to
    # one device, G2V_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU)
    if "G2V_BENCH_DEVICE" in os.environ:
        local = int(os.environ["G2V_BENCH_DEVICE"])
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        backend = os.environ.get("G2V_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from g2vlm_amd import hip
    from g2vlm_amd.g2vlm_utils import build_model, configs_from_dims
    from g2vlm_amd.modeling.g2vlm import NaiveCache
    from g2vlm_amd.synthetic import REAL_DIMS, SyntheticStateDict
    hip.lib()
    dims = REAL_DIMS
    model = build_model(*configs_from_dims(dims), SyntheticStateDict(dims, dev, seed=0), dev)
    tok = _Tok()
    g = torch.Generator(); g.manual_seed(1000 + rank)
    imgs = torch.rand((N_VIEWS, 3, HW, HW), generator=g)
    gi_text, nl, nr = model.prepare_prompts_addbos([0], [0], ["Reconstruct the 3D scene."], tok, NEW_TOKEN_IDS)
    gi, nl2, nr2 = model.prepare_dino_images_pi3(nl, nr, imgs, None, NEW_TOKEN_IDS)
    assert gi["packed_dino_images"].is_cuda and gi["original_images"].is_cuda    # resident in HBM before the timed region
    P = (HW // 14) ** 2
    lq = N_VIEWS * (P + 2)
    cap = T0 + lq + 256

    if a.workload == "c4":
        # BASELINE config 4.  --gpus 8: ONE 32-view scene, 4 views per rank, K/V all-gather per MoT layer.  --gpus 1: the same
        # 32-view scene unsharded on one MI355X (Lq 43 872 x Lk 43 880: the global attention is 63 % of its FLOPs), with the
        # attention kernel's roofline from HIP events around each of its 28 launches per step.
        from g2vlm_amd.sharded import LocalComm, TorchDistComm, recon_view_sharded
        if world > 1 and os.environ.get("G2V_COMM", "torch") == "abi":
            from g2vlm_amd.comm import RcclComm                 # RCCL through the C-ABI of include/g2vlm_comm.h (no process group on the data path)
            comm = RcclComm.from_env(dev)
        else:
            comm = TorchDistComm() if world > 1 else LocalComm()
        nv_total = 4 * world if world > 1 else 32
        gg = torch.Generator(); gg.manual_seed(2000)
        imgs4 = hip.h2d(torch.rand((nv_total, 3, HW, HW), generator=gg), dev)    # same scene on every rank, resident in HBM
        for _ in range(a.warmup):
            recon_view_sharded(model, comm, tok, NEW_TOKEN_IDS, imgs4, gather=False)
        attn_events = []
        if rank == 0:
            model.engine.attn_events = attn_events
        comm.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            r4 = recon_view_sharded(model, comm, tok, NEW_TOKEN_IDS, imgs4, gather=False)
        comm.barrier(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        model.engine.attn_events = None
        from g2vlm_amd import dist_util
        dt = dist_util.max_over_ranks(dt, dev)
        assert torch.isfinite(r4["points"]).all()
        if rank == 0:
            lq4 = nv_total * ((HW // 14) ** 2 + 2)
            lq_loc, tot4 = lq4 // world, T0 + lq4
            durs = [ev[0].elapsed_time(ev[1]) for ev, l_, t_ in attn_events if l_ == lq_loc and t_ == tot4]
            k_ms = sum(durs) / max(1, len(durs))
            fl_launch = 4 * lq_loc * tot4 * dims["llm"]["heads"] * 128
            ach = fl_launch / (k_ms * 1e-3) / 1e12 if durs else None
            fl4 = flops_per_scene(dims, nv_total, (HW // 14) ** 2, T0)
            print(json.dumps({"metric": "views/sec (32-view reconstruction)" if world == 1 else "views/sec (view-sharded 4 views/GPU, RCCL K/V all-gather)",
                              "value": round(nv_total * a.steps / dt, 3),
                              "unit": "views/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                              "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
                              "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                              "config": {"workload": f"C4: one {nv_total}-view 518x518 scene, " + ("unsharded on one GPU" if world == 1 else "4 views per GPU"),
                                         "Lq": lq4, "Lk": tot4, "parallelism": f"view-sharded x{world}"},
                              "tflops_per_scene": round(fl4["total"] / 1e12, 1),
                              "achieved_tflops_per_gpu": round(fl4["total"] * a.steps / dt / 1e12 / world, 1),
                              "roofline": dict(bound="mfma", kernel="flash_fwd64_kernel (+ combine), rank 0's launches", achieved=round(ach, 1) if ach else None,
                                               peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(ach / PEAK_BF16_TFLOPS, 4) if ach else None,
                                               traffic=None, launch_ms=round(k_ms, 4), launches_timed=len(durs), flops_per_launch=fl_launch)}),
                  flush=True)
        if world > 1:
            import torch.distributed as dist
            dist.barrier(); dist.destroy_process_group()
        return

    if a.workload == "c5":
        # ViT inputs through the product's own front end: random 768x768 uint8 frames -> host.QwenVL2ImageTransform on the
        # device (uint8 upload, g2v_qwen_patchify_u8) -> bf16 [2916, Kpad] patch matrices, resident before the timed region
        import numpy as np
        from PIL import Image
        from g2vlm_amd import host
        rng = np.random.default_rng(3000 + rank)
        tf = host.QwenVL2ImageTransform(768, 768, 14, device=dev, k_pad=model.weights["vit.patch.w"].shape[1])
        vit_in = []
        for _ in range(N_VIEWS):
            pv, thw = tf([Image.fromarray(rng.integers(0, 256, size=(768, 768, 3), dtype=np.uint8))])
            vit_in.append((pv, tuple(int(v) for v in thw[0])))

        class _Tok32(_Tok):
            def encode(self, text, add_special_tokens=False):
                return list(range(11, 11 + 32)) if "?" in text else [11, 12, 13, 14, 15, 16, 17]

            def decode(self, ids):
                return ""

        tok5 = _Tok32()
        n_tok = 256

        SPS = a.scenes_per_step                               # BASELINE config 5: 2 scenes per GPU, decoded together

        def scene():
            for _ in range(SPS):
                past = NaiveCache(dims["llm"]["layers"], dims["llm"]["kv_heads"], dev, capacity=cap)
                past, last = model.prefill_text_and_dino(past, gi_text, gi)
                model.reconstruct(past_key_values=past, selected_hidden_states=last, **gi)
            it = iter(vit_in * SPS)
            image_transform = lambda _im: (lambda pv, thw: (pv, torch.tensor([list(thw)])))(*next(it))
            real_eos, NEW_TOKEN_IDS["eos_token_id"] = NEW_TOKEN_IDS["eos_token_id"], -1      # decode all 256 tokens
            try:
                if SPS == 1:
                    model.chat_with_recon(tok5, NEW_TOKEN_IDS, image_transform, None, images=imgs,
                                          prompt="How far is the chair from the door?", max_length=n_tok)
                else:
                    model.chat_with_recon_batch(tok5, NEW_TOKEN_IDS, image_transform, None,
                                                [(imgs, "How far is the chair from the door?")] * SPS, max_length=n_tok)
            finally:
                NEW_TOKEN_IDS["eos_token_id"] = real_eos

        for _ in range(a.warmup):
            scene()
        torch.cuda.synchronize()
        if a.breakdown and rank == 0:
            # one extra, untimed step with a device sync around every stage: where a C5 step spends its wall time
            acc = {}

            def timed(name):
                fn = getattr(model, name)

                def wrap(*args, **kw):
                    torch.cuda.synchronize(); t = time.perf_counter()
                    r = fn(*args, **kw)
                    torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
                    return r
                setattr(model, name, wrap)
                return fn
            names = ("forward_cache_update_text", "forward_cache_update_dino", "reconstruct", "forward_cache_update_vit",
                     "forward_cache_update_vit_multi", "generate_text", "generate_text_batch")
            saved = {n: timed(n) for n in names}
            t = time.perf_counter(); scene(); torch.cuda.synchronize(); tot_b = time.perf_counter() - t
            for n, fn in saved.items():
                setattr(model, n, fn)
            print(json.dumps({"c5_breakdown_ms": {k: round(v * 1e3, 1) for k, v in acc.items()}, "step_ms": round(tot_b * 1e3, 1)}),
                  file=sys.stderr, flush=True)
        if os.environ.get("G2V_C5_SYNC"):                     # experiment: device sync after the named stages
            for n in os.environ["G2V_C5_SYNC"].split(","):
                fn = getattr(model, n)
                setattr(model, n, (lambda f: lambda *ar, **kw: (f(*ar, **kw), torch.cuda.synchronize())[0])(fn))
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        prof = None
        if os.environ.get("G2V_C5_CPROFILE"):
            import cProfile
            prof = cProfile.Profile(); prof.enable()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            scene()
        torch.cuda.synchronize()
        if prof is not None:
            import pstats
            prof.disable()
            pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(22)
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        from g2vlm_amd import dist_util
        dt = dist_util.max_over_ranks(dt, dev) if world > 1 else dt
        if rank == 0:
            print(json.dumps({"metric": "scenes/sec (interleaved 8-view recon + 8-image chat, 256 greedy tokens)",
                              "value": round(world * a.steps * SPS / dt, 4), "unit": "scenes/s", "n_gpus": world, "steps": a.steps,
                              "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 1), "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                              "views_per_s": round(world * a.steps * SPS * N_VIEWS / dt, 2), "decode_tokens_per_s_incl_prefill": round(world * a.steps * SPS * n_tok / dt, 1),
                              "config": {"workload": "C5: per scene 8-view 518x518 reconstruction, then chat over the same views (DINO geo prefill, "
                                                     f"8 ViT images of 2916 patches, 32-token question, 256 greedy tokens); {SPS} scene(s) per rank and step, decoded together",
                                         "parallelism": f"replicas x{world}"}}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    def step():
        past = NaiveCache(dims["llm"]["layers"], dims["llm"]["kv_heads"], dev, capacity=cap)
        past, last = model.prefill_text_and_dino(past, gi_text, gi)     # = forward_cache_update_text + forward_cache_update_dino (G2VLM.recon)
        pred = model.reconstruct(past_key_values=past, selected_hidden_states=last, **gi)
        return past, pred

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # the host only enqueues (~2 000 launches per scene, 8.5 ms against 75 ms of GPU work); a cyclic-GC pass over the model's
    # object graph in the middle of a step is the one host-side pause long enough to starve the stream: collect now, not then -
    # and before the warm-up steps, so that the caching allocator's pools settle AFTER the collection has released its tensors
    import gc
    gc.collect()
    gc.disable()
    attn_events, gemm_events = [], []
    for i in range(a.warmup):
        if rank == 0 and i == a.warmup - 1:            # the last warm-up step already carries the event records of the timed steps
            model.engine.attn_events = attn_events      # two event records per MoT layer; rank 0 only (max-over-ranks keeps it honest)
            model.engine.gemm_events = gemm_events      # and two around its gate/up GEMM
        past, pred = step()
    if rank == 0:
        model.engine.attn_events = attn_events
        model.engine.gemm_events = gemm_events
        torch.cuda.synchronize()
        attn_events.clear(); gemm_events.clear()        # only the timed steps' records are read below
    barrier()
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]    # per-step GPU times (diagnostic key `step_ms`)
    ms0 = torch.cuda.memory_stats(dev) if os.environ.get("G2V_STEP_DIAG") else None  # tools/step_outliers.py
    t0 = time.perf_counter()
    step_ev[0].record()
    diag_allocs = []
    for i in range(a.steps):
        past, pred = step()
        step_ev[i + 1].record()
        if ms0 is not None:
            diag_allocs.append(torch.cuda.memory_stats(dev)["num_device_alloc"] - ms0["num_device_alloc"])
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if ms0 is not None:
        ms1 = torch.cuda.memory_stats(dev)
        print("G2V_STEP_DIAG device mallocs in the timed region:", ms1["num_device_alloc"] - ms0["num_device_alloc"], "frees:",
              ms1["num_device_free"] - ms0["num_device_free"], "retries:", ms1["num_alloc_retries"] - ms0["num_alloc_retries"],
              "reserved GB:", round(ms1["reserved_bytes.all.current"] / 2 ** 30, 2), "cumulative by step:", diag_allocs,
              "reserved MB added:", round((ms1["reserved_bytes.all.current"] - ms0["reserved_bytes.all.current"]) / 2 ** 20, 1), file=sys.stderr, flush=True)
    step_ms = [round(step_ev[i].elapsed_time(step_ev[i + 1]), 2) for i in range(a.steps)]
    model.engine.attn_events = None
    model.engine.gemm_events = None
    overlap_vps = None
    if a.overlap > 1:
        # secondary figure, measured AFTER the timed region and not part of `value`: scenes of consecutive steps issued on
        # `overlap` streams of this process, so one scene's kernel tails and bandwidth-bound passes are filled by the other's
        # MFMA kernels.  Same work per scene; the per-kernel roofline above is taken on the serial steps, where a kernel has
        # the chip to itself.
        streams = [torch.cuda.Stream(device=dev) for _ in range(a.overlap)]
        for s_ in streams:
            s_.wait_stream(torch.cuda.current_stream())
        torch.cuda.synchronize(); t_o = time.perf_counter()
        for i in range(a.steps * a.overlap):
            with torch.cuda.stream(streams[i % a.overlap]):
                step()
        torch.cuda.synchronize()
        overlap_vps = N_VIEWS * a.steps * a.overlap / (time.perf_counter() - t_o)
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
    assert torch.isfinite(pred["points"]).all()
    views_per_s = world * N_VIEWS * a.steps / dt

    if rank == 0:
        fl = flops_per_scene(dims, N_VIEWS, P, T0)
        # ---- dominant kernel: MoT flash attention (hd 128, GQA 12:2, Lq 10968 x Lk 10976).  HIP events recorded on the
        # launch stream around every one of its launches INSIDE the timed steps above (28 layers x K steps); the text
        # prefill's 8-row launches are left out by shape.  A launch = flash_fwd64_kernel + its split-KV combine.
        L = dims["llm"]
        tot = T0 + lq
        durs = [ev[0].elapsed_time(ev[1]) for ev, l_, t_ in attn_events if l_ == lq and t_ == tot]
        assert len(durs) == L["layers"] * a.steps, (len(durs), L["layers"], a.steps)
        k_ms = sum(durs) / len(durs)
        ach = fl["mot_attention_per_launch"] / (k_ms * 1e-3) / 1e12
        roofline = dict(bound="mfma", kernel="flash_fwd64_kernel (+ flash_combine_kernel<128>)", achieved=round(ach, 1),
                        peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(ach / PEAK_BF16_TFLOPS, 4), traffic=TRAFFIC_BYTES_PER_LAUNCH,
                        launch_ms=round(k_ms, 4), launches_timed=len(durs), flops_per_launch=fl["mot_attention_per_launch"],
                        traffic_note=TRAFFIC_NOTE)
        # ---- the largest Linear (gate || up with the SwiGLU epilogue: 2 x Lq x 17920 x 1536 FLOP, 27 % of a step's GEMM work), same
        # method: events around its 28 x K launches inside the timed steps
        gd = [ev[0].elapsed_time(ev[1]) for ev, l_ in gemm_events if l_ == lq]
        roofline_gemm = None
        if gd:
            g_ms = sum(gd) / len(gd)
            g_fl = 2.0 * lq * 2 * L["ffn"] * L["hidden"]
            roofline_gemm = dict(bound="mfma", kernel="gemm8p_kernel<SWIGLU> (MoT gate||up Linear, geo + und groups)", achieved=round(g_fl / (g_ms * 1e-3) / 1e12, 1),
                                 peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(g_fl / (g_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                 launch_ms=round(g_ms, 4), launches_timed=len(gd), flops_per_launch=int(g_fl))
        # ---- greedy decode tokens/s on the und expert, KV = the scene's 10976 cached rows (second headline metric).
        # HBM roofline per token: und-expert weights + lm_head once, K + V rows of every layer once.
        w_bytes = L["layers"] * 2 * (L["hidden"] * (L["heads"] + 2 * L["kv_heads"]) * 128 + L["heads"] * 128 * L["hidden"]
                                     + 3 * L["hidden"] * L["ffn"]) + 2 * L["vocab"] * L["hidden"]
        kv_bytes = L["layers"] * 2 * L["kv_heads"] * 128 * 2 * tot
        tok_s, decode_roofline, decode_batch = None, None, {}
        if a.decode_tokens > 0:
            gs = dict(packed_start_tokens=torch.tensor([5]), packed_query_position_ids=torch.tensor([[nr2[0]]] * 3),
                      key_values_lens=torch.tensor([past.length], dtype=torch.int), packed_key_value_indexes=torch.arange(past.length))
            model.generate_text(past, max_length=4, **gs)
            gs["key_values_lens"] = torch.tensor([past.length], dtype=torch.int)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            model.generate_text(past, max_length=a.decode_tokens, **gs)
            torch.cuda.synchronize()
            tok_s = a.decode_tokens / (time.perf_counter() - t1)
            decode_roofline = dict(bound="hbm", kernel="batch-1 greedy decode step (28 und-expert layers + lm_head, one hipGraph replay per token)",
                                   bytes_per_token=int(w_bytes + kv_bytes), ms_per_token=round(1e3 / tok_s, 4),
                                   achieved=round((w_bytes + kv_bytes) * tok_s / 1e9, 1), peak=PEAK_HBM_GBPS, unit="GB/s",
                                   frac=round((w_bytes + kv_bytes) * tok_s / 1e9 / PEAK_HBM_GBPS, 4), kv_len=int(tot),
                                   note="whole call incl. cache hand-over and id read-back; per-kernel table: profiles/r02*_decode_kernel_stats.csv")
            # ---- batched decode (SURVEY 8f-3 / 8d "b scenes sharing weights"): B copies of the scene's cache decoded together
            for B in (2, 8):
                gsb = [dict(gs, key_values_lens=torch.tensor([past.length], dtype=torch.int)) for _ in range(B)]
                model.generate_text_batch([past] * B, gsb, 4)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                model.generate_text_batch([past] * B, gsb, a.decode_tokens)
                torch.cuda.synchronize()
                dtb = time.perf_counter() - t1
                decode_batch[str(B)] = {"tokens_per_s": round(B * a.decode_tokens / dtb, 1),
                                        "ms_per_step": round(dtb / a.decode_tokens * 1e3, 3),
                                        "hbm_gb_per_s_incl_setup": round((w_bytes + B * kv_bytes) * a.decode_tokens / dtb / 1e9, 1)}
        out = {
            "metric": "views/sec (multi-view recon, G2VLM-2B-MoT)", "value": round(views_per_s, 3), "unit": "views/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "C3: 8-view 518x518 reconstruction, full depth (24 DINO + 28 MoT + 15 decoder blocks), 1 scene per GPU",
                       "views_per_scene": N_VIEWS, "Lq": lq, "Lk": tot, "parallelism": f"replicas x{world}"},
            "tflops_per_scene": round(fl["total"] / 1e12, 2),
            "achieved_tflops_per_gpu": round(fl["total"] * a.steps / dt / 1e12, 1),
            "decode_tokens_per_s": round(tok_s, 1) if tok_s else None, "decode_kv_len": int(tot),
            "views_per_s_scenes_on_two_streams": round(overlap_vps, 2) if overlap_vps else None,
            "step_ms": step_ms, "decode_batch": decode_batch, "decode_roofline": decode_roofline, "roofline_gemm": roofline_gemm,
            "host_prep_ms": host_prep_ms(model, N_VIEWS),
            "host_prep_note": "per 8-view scene, NOT in `value`: pinned upload of 8 decoded 1280x720 uint8 frames + device LANCZOS resize to 518 wide (bit-exact with Pillow) + device normalise; image file decoding not included (no files offline)",
            "roofline": roofline,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dims)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
