"""View-sharded multi-view reconstruction (BASELINE config C4: 32 views, 4 per GPU, 8 GPUs).

The reference has no multi-GPU inference; this is the new design of SURVEY §8e:
one process per GPU, rank r owns views [lo, hi).  Per-view work (DINO, point/camera decoders,
heads) is local; the global cross-view MoT attention needs every rank's K/V, exchanged with ONE
in-place all-gather of K and of V per layer over RCCL/xGMI (equal contiguous blocks of the KV
cache, because a rank's views are contiguous in the packed sequence).  Two small one-off
exchanges: DINO boundary rows (hazard H1, below) and view 0's hidden state for the global decoder.

Hazard H1 under sharding.  The reference's DINO attention windows are [i*P, (i+1)*P) of the flat
[N*(P+5)] token axis (App. D-H1): they straddle view boundaries, but they are the SAME disjoint
row sets in every layer, so token rows never interact across windows.  The encoder is therefore
sharded by WINDOW, not by view: rank r runs windows [lo, hi) = flat rows [lo*P, hi*P) (the last
rank also carries the 5N uncovered rows).  Those rows cover views lo-1 (its last 5*lo rows) ..
hi-1 (all but its last 5*hi rows), so after the encoder every rank passes its first 5*lo token
rows to rank r-1 — one tiny exchange instead of a per-layer halo.

Collectives go through a small `Comm` interface with three implementations: `TorchDistComm`
(torch.distributed; "nccl" = RCCL on ROCm, "gloo" on CPU), `g2vlm_amd.comm.RcclComm` (RCCL through
the C-ABI of include/g2vlm_comm.h, no process group on the data path) and `ThreadSimComm`, which
runs the W ranks as threads of one process on one GPU so the sharding algebra is tested against
the unsharded engine without an 8-GPU node.
"""
import os
import threading

import torch

from . import hip, host
from .dist_util import shard_views
from .modeling.g2vlm.qwen2vl import NaiveCache


# ----------------------------------------------------------------------------- communicators
class TorchDistComm:
    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    @property
    def overlappable(self):
        """Collectives may be issued on a side stream while the main stream computes (see KVExchange): RCCL orders a
        collective behind the stream that is current when it is called and makes THAT stream wait for its end; gloo is a
        host-side exchange (staged through host memory below), nothing to overlap."""
        return self.dist.get_backend(self.group) != "gloo"

    def all_gather_blocks(self, full, block_rows):
        """All-gather into `full`: rank r's rows [r*block_rows, (r+1)*block_rows) are already valid on rank r.
        RCCL/NCCL: ONE in-place all_gather_into_tensor - the send block is the rank's own slice of the receive buffer
        (ncclAllGather's in-place form, what FSDP's all-gather uses too), so nothing is staged or copied.
        gloo (rehearsal: CPU tensors, or several ranks on one GPU): list form on host copies - gloo's own device-tensor
        support covers only broadcast / all-reduce."""
        mine = full[self.rank * block_rows:(self.rank + 1) * block_rows]
        if self.dist.get_backend(self.group) == "gloo":
            send = mine.detach().cpu().contiguous()          # a blocking copy: the stream's earlier writes have landed
            parts = [torch.empty_like(send) for _ in range(self.world)]
            self.dist.all_gather(parts, send, group=self.group)
            for r, p in enumerate(parts):
                if r != self.rank:
                    full[r * block_rows:(r + 1) * block_rows].copy_(p)
        else:
            self.dist.all_gather_into_tensor(full, mine, group=self.group)

    def broadcast(self, t, src):
        if self.dist.get_backend(self.group) == "gloo" and t.is_cuda:
            h = t.detach().cpu()
            self.dist.broadcast(h, src=src, group=self.group)
            if self.rank != src:
                t.copy_(h)
            return
        self.dist.broadcast(t, src=src, group=self.group)

    def barrier(self):
        self.dist.barrier(group=self.group)


class LocalComm:
    """world = 1: every collective is a no-op."""
    rank, world = 0, 1
    overlappable = False

    def all_gather_blocks(self, full, block_rows):
        pass

    def broadcast(self, t, src):
        pass

    def barrier(self):
        pass


class ThreadSimComm:
    """W simulated ranks = W host threads sharing one device, every rank on its OWN stream (run_thread_sim), as W processes
    would be.  Two forms of every collective:

    * drained (default): a device-wide synchronize + a host barrier on both sides of the copies - nothing to get wrong,
      the form the size / algebra checks use;
    * `overlap=True`: stream-ordered, the way RCCL behaves - no device synchronize anywhere.  A rank records a `ready`
      event on the stream the collective is called on, the host threads meet, every rank makes that stream wait for its
      peers' `ready` events, copies their blocks, records `done`, and the collective completes on a rank's stream only once
      every peer has read its block (a send buffer may be reused after the collective).  `overlappable` is then true, so
      KVExchange takes its side-stream branch: the event ordering of the RCCL path (cache write -> gather on the side
      stream, local-block attention meanwhile -> join -> remote blocks) runs under test on one GPU.  `delay_cycles` holds
      the side stream back before the copies, so an attention launched without the join would read blocks that have not
      arrived."""

    class _Shared:
        def __init__(self, world):
            self.world, self.barrier, self.slots = world, threading.Barrier(world), {}

    def __init__(self, shared, rank, overlap=False, delay_cycles=0):
        self.sh, self.rank, self.world = shared, rank, shared.world
        self.overlappable, self.delay, self.seq = bool(overlap), int(delay_cycles), 0

    def _sync(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.sh.barrier.wait()

    # ---- stream-ordered form
    def _exchange(self, buf, copy_from_peers, senders):
        """One collective on the current stream: publish (buf, ready), meet, wait for the senders' events, run
        `copy_from_peers(slots)`, then hold my stream until every peer is done reading."""
        cur = torch.cuda.current_stream()
        seq, self.seq = self.seq, self.seq + 1
        for k in list(self.sh.slots):                            # (list(dict) is one atomic step under the GIL)
            if isinstance(k, tuple) and len(k) == 3 and k[1] == self.rank and k[2] < seq - 1:
                del self.sh.slots[k]                             # everyone is past collective seq-2 once seq-1's barriers fell
        ready = torch.cuda.Event()
        ready.record(cur)
        self.sh.slots[("buf", self.rank, seq)] = (buf, ready)
        self.sh.barrier.wait()
        if self.delay:
            torch.cuda._sleep(self.delay)
        peers = {r: self.sh.slots[("buf", r, seq)] for r in senders if r != self.rank}
        for r, (_, ev) in peers.items():
            cur.wait_event(ev)
        copy_from_peers({r: b for r, (b, _) in peers.items()})
        done = torch.cuda.Event()
        done.record(cur)
        self.sh.slots[("done", self.rank, seq)] = done
        self.sh.barrier.wait()
        for r in range(self.world):
            if r != self.rank:
                cur.wait_event(self.sh.slots[("done", r, seq)])

    def all_gather_blocks(self, full, block_rows):
        if self.overlappable and full.is_cuda:
            def copies(peers):
                for r, src in peers.items():
                    full[r * block_rows:(r + 1) * block_rows].copy_(src[r * block_rows:(r + 1) * block_rows], non_blocking=True)
            return self._exchange(full, copies, range(self.world))
        self.sh.slots[("ag", self.rank)] = full
        self._sync()
        for r in range(self.world):
            if r != self.rank:
                src = self.sh.slots[("ag", r)]
                full[r * block_rows:(r + 1) * block_rows].copy_(src[r * block_rows:(r + 1) * block_rows])
        self._sync()

    def broadcast(self, t, src):
        if self.overlappable and t.is_cuda:
            def copies(peers):
                if self.rank != src:
                    t.copy_(peers[src], non_blocking=True)
            return self._exchange(t, copies, (src,))
        if self.rank == src:
            self.sh.slots["bc"] = t
        self._sync()
        if self.rank != src:
            t.copy_(self.sh.slots["bc"])
        self._sync()

    def barrier(self):
        self._sync()


class KVExchange:
    """Per-layer K/V exchange of the view-sharded prefill, overlapped with the local-block attention.
    start(i): the all-gathers of layer i's K and V blocks are issued on a communication stream that waits for the main
    stream's cache write; wait(i): the main stream waits for them.  Between the two calls the engine runs the attention
    over the rank's own block.  On a communicator that cannot overlap (thread simulation, gloo on CPU tensors, world 1)
    start() does the exchange synchronously; `overlap=False` (recon_view_sharded(kv_overlap=False)) forces that form on
    any communicator.

    Which branch runs where: RCCL (`TorchDistComm` on backend "nccl") takes the side-stream branch - it has NOT run on
    hardware yet (no multi-GPU box in this pipeline); the same branch runs under test with `ThreadSimComm(overlap=True)`;
    gloo and the drained thread simulation take the synchronous branch."""

    def __init__(self, comm, past, first_row, total_rows, block_rows, device, overlap=True):
        self.comm, self.past, self.r0, self.n, self.blk = comm, past, first_row, total_rows, block_rows
        use_side = overlap and bool(getattr(comm, "overlappable", False)) and torch.device(device).type == "cuda"
        self.side = torch.cuda.Stream(device=device) if use_side else None
        self.done = None
        self.overlapped_layers = 0                                           # introspection for the tests

    def _gather(self, i):
        k, v = self.past.k[i][self.r0:self.r0 + self.n], self.past.v[i][self.r0:self.r0 + self.n]
        if hasattr(self.comm, "all_gather_kv"):              # g2vlm_amd.comm.RcclComm: K and V as one grouped RCCL launch
            self.comm.all_gather_kv(k, v, self.blk)
        else:
            self.comm.all_gather_blocks(k, self.blk)
            self.comm.all_gather_blocks(v, self.blk)

    def start(self, i):
        if self.side is None:
            self._gather(i)
            return
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            self._gather(i)
            self.done = torch.cuda.Event()
            self.done.record(self.side)
        self.overlapped_layers += 1

    def wait(self, i):
        if self.side is not None:
            torch.cuda.current_stream().wait_event(self.done)


# ----------------------------------------------------------------------------- the sharded forward
@torch.no_grad()
def geo_prefill_view_sharded(model, comm, past, newlens, new_rope, images, new_token_ids, kv_overlap=True, text_inputs=None):
    """G2VLM.prepare_dino_images_pi3 + forward_cache_update_dino (reference g2vlm.py:868-1039) with the views sharded over
    comm.world ranks.  `past` holds the (replicated) stages before the views; on return it holds, on EVERY rank, the K/V
    rows of all N views as well (the per-layer all-gather leaves the full cache everywhere), so any later stage - the
    ViT images, the question, the decode of chat_with_recon - runs on any one rank unchanged.

    Returns (gi, newlens, new_rope, hidden, shape): gi = the reference's generation-input dict for ALL views, hidden = fp32
    [nv*P, H] final-norm rows of this rank's geo tokens (view-major), shape = dict(N, lo, hi, P, gh, gw, Hh, Ww, overlapped).

    Encoder variants: DINOv2 (1 + 4 prefix tokens per view) and DINOv3 (`use_dinov3`: 1 + R prefix tokens, patch 16).  Both
    are called with cumulative PATCH counts as window boundaries (hazard H1), so both are sharded by window and exchange
    their `pre * lo` boundary rows once.

    text_inputs: the generation inputs of the (replicated) text stage in front of the views, NOT yet run: it is then run here,
    on a side stream under the encoder (G2VLM.prefill_text_and_dino: the prefix does not depend on the encoder; the cache is
    sized on the caller's stream first), and joined in front of the first MoT kernel."""
    eng, hp, w = model.engine, hip, model.weights
    dev, H = model.device, model.hidden_size
    rank, world = comm.rank, comm.world
    T0 = newlens[0]
    assert T0 == past.length + (text_inputs["packed_text_ids"].numel() if text_inputs is not None else 0)

    # ---- global bookkeeping (host ints), then this rank's subset
    gi, newlens, new_rope = model.prepare_dino_images_pi3(newlens, new_rope, images, None, new_token_ids)
    imgs = gi["packed_dino_images"]
    N, _, Hh, Ww = imgs.shape
    assert N % world == 0, "views must divide evenly over ranks (equal K/V blocks for the in-place all-gather)"
    ps = model.dino_patch_size
    gh, gw = Hh // ps, Ww // ps
    pre = 1 + (w.dinov3.config.num_register_tokens if model.use_dinov3 else 4)      # cls + registers in front of a view's patches
    P, S = gh * gw, gh * gw + pre
    lo, hi = shard_views(N, world, rank)
    nv = hi - lo
    assert pre * N < P, "boundary exchange assumes a window straddles at most two views"
    blk = nv * (P + 2)                                                   # packed rows per rank
    Lq = N * (P + 2)
    past.reserve(T0 + Lq)
    join_text = None
    if text_inputs is not None:
        if os.environ.get("G2V_TEXT_OVERLAP", "1") == "0" or torch.cuda.is_current_stream_capturing():
            model.forward_cache_update_text(past, **text_inputs)
        else:
            cur = torch.cuda.current_stream()
            side = eng.side_stream(cur)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                model.forward_cache_update_text(past, **text_inputs)
            join_text = lambda: cur.wait_stream(side)                    # noqa: E731

    # ---- encoder, sharded by window (H1): flat rows [lo*P, hi*P) (+ the uncovered tail on the last rank)
    va = max(lo - 1, 0)                                                  # first view whose tokens we touch
    f0, f1 = lo * P, hi * P + (pre * N if rank == world - 1 else 0)      # global flat rows of this rank
    views = hip.h2d(imgs[va:hi], dev, torch.float32).contiguous()
    if model.use_dinov3:
        v3 = w.dinov3
        x_views, _, _ = v3.embed(views)                                  # [(hi-va)*S, C]
        cos, sin = v3._rope_rows(hi - va, gh, gw)
        sl = slice(f0 - va * S, f1 - va * S)
        tok = v3.run_layers(x_views[sl].contiguous(), cos[sl].contiguous(), sin[sl].contiguous(), [i * P for i in range(nv + 1)])
        tok32 = hp.linear(hp.cast_bf16(tok), w["dino2llm.w"], w["dino2llm.b"], hp.EPI_RES_F32)      # fp32 [f1-f0, H]
    else:
        x_views = eng.dino_embed(views)                                  # [(hi-va)*S, C]
        tok = eng.dino_layers(x_views[f0 - va * S: f1 - va * S].contiguous(), nv, P)                 # bf16 [f1-f0, C]
        tok32 = hp.linear(tok, w["dino2llm.w"], w["dino2llm.b"], hp.EPI_RES_F32)                    # fp32 [f1-f0, H]
    # boundary exchange: my first pre*lo rows belong to view lo-1 (rank r-1); I need the next rank's first pre*hi rows
    bmax = pre * N
    send = torch.zeros((world * bmax, H), dtype=torch.float32, device=dev)
    if lo > 0:
        send[rank * bmax: rank * bmax + pre * lo].copy_(tok32[:pre * lo])
    comm.all_gather_blocks(send, bmax)
    x = torch.empty((blk, H), dtype=torch.float32, device=dev)           # local MoT rows: [nv*P geo | 2*nv und]
    # patch rows of view v = global flat rows [v*S+pre, (v+1)*S)
    for v in range(lo, hi):
        g0, g1 = v * S + pre, (v + 1) * S
        dst = x[(v - lo) * P:(v - lo + 1) * P]
        have1 = min(g1, f1)                                              # rows available locally
        dst[:have1 - g0].copy_(tok32[g0 - f0: have1 - f0])
        if have1 < g1:                                                   # tail lives in the next rank's boundary block
            need = g1 - have1
            assert need == pre * hi and v == hi - 1
            dst[have1 - g0:].copy_(send[(rank + 1) * bmax:(rank + 1) * bmax + need])

    # ---- MoT geo prefill on local rows; K/V all-gathered per layer
    geo_idx, text_idx = gi["packed_dino_token_indexes"].long(), gi["packed_text_indexes"].long()
    geo_loc = geo_idx[lo * P: hi * P]
    text_loc = text_idx[2 * lo: 2 * hi]
    perm = torch.cat([geo_loc, text_loc])                                # local order, global packed indices
    eng.embed(model._dev_i32(gi["packed_text_ids"][2 * lo: 2 * hi]), x[nv * P:])
    pos = model._dev_i32(gi["packed_position_ids"][:, perm])
    kv_rows = model._dev_i32(gi["packed_indexes"][perm])
    if join_text is not None:
        join_text()

    overlapped = 0
    if world > 1:
        exchange = KVExchange(comm, past, T0, Lq, blk, dev, overlap=kv_overlap)
        last = eng.llm_forward(x, nv * P, pos, kv_rows, past, T0, causal=False, und_rounding=0, kv_total=T0 + Lq, kv_exchange=exchange,
                               local_kv=(T0 + rank * blk, blk))
        overlapped = exchange.overlapped_layers
    else:                                                                # one rank: nothing to exchange, one attention launch
        last = eng.llm_forward(x, nv * P, pos, kv_rows, past, T0, causal=False, und_rounding=0, kv_total=T0 + Lq)
    hidden = last[:nv * P].contiguous()                                  # geo rows of my views, view-major
    return gi, newlens, new_rope, hidden, dict(N=N, lo=lo, hi=hi, P=P, gh=gh, gw=gw, Hh=Hh, Ww=Ww, overlapped=overlapped)


@torch.no_grad()
def recon_view_sharded(model, comm, tokenizer, new_token_ids, images, gather=True, kv_overlap=True):
    """G2VLM.recon (reference g2vlm.py:1240-1303) with views sharded over comm.world ranks.

    `images`: the full [N,3,H,W] tensor in [0,1] (or paths/PIL list) on every rank; N % world == 0.
    Returns this rank's slice of the reference's output dict (keys as G2VLM.recon, leading dims
    [1, n_local, ...]) plus 'view_range'; with gather=True the per-view tensors are all-gathered so every
    rank holds all N views.  kv_overlap=False: the per-layer K/V exchange runs on the main stream even on a communicator
    that could overlap it (KVExchange).
    """
    eng, dev, H = model.engine, model.device, model.hidden_size
    rank = comm.rank
    L = model.dims["llm"]
    # ---- replicated text prefix (identical on every rank)
    past = NaiveCache(L["layers"], L["kv_heads"], dev)
    gi_text, newlens, new_rope = model.prepare_prompts_addbos([0], [0], ["Reconstruct the 3D scene."], tokenizer, new_token_ids)
    gi, newlens, new_rope, hidden, sh = geo_prefill_view_sharded(model, comm, past, newlens, new_rope, images, new_token_ids, kv_overlap,
                                                                 text_inputs=gi_text)
    N, lo, hi, P, gh, gw, Hh, Ww = (sh[k] for k in ("N", "lo", "hi", "P", "gh", "gw", "Hh", "Ww"))
    nv = hi - lo

    # ---- decoders: local views; the global decoder's context is view 0 (rank 0)
    ctx = torch.empty((P, H), dtype=torch.float32, device=dev)
    if rank == 0:
        ctx.copy_(hidden[:P])
    comm.broadcast(ctx, 0)
    _, _, _, points, local, poses, glob = eng.decoders_and_heads(hidden, ctx, nv, gh, gw, Hh, Ww)
    out = dict(points=points, local_points=local, camera_poses=poses, global_points=glob,
               images=hip.h2d(gi["original_images"][lo:hi], dev))
    if gather:
        full = {}
        for k, t in out.items():
            buf = torch.empty((N,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
            buf[lo:hi].copy_(t)
            comm.all_gather_blocks(buf.view(N, -1), nv)
            full[k] = buf
        out = full
        lo, hi = 0, N
    res = {k: v.unsqueeze(0) for k, v in out.items()}
    res["conf"] = None
    res["view_range"] = (lo, hi)
    res["past_key_values"] = past
    res["kv_layers_overlapped"] = sh["overlapped"]
    return res


@torch.no_grad()
def chat_view_sharded(model, comm, tokenizer, new_token_ids, image_transform, images, prompt, max_length, decode_rank=0,
                      kv_overlap=True, return_ids=False):
    """G2VLM.chat_with_recon (reference g2vlm.py:1305-1410) with the geometry prefill of the N views sharded over the ranks
    (second half: decode after a view-sharded prefill).

    The reference is one process, batch 1 (g2vlm.py:1006, 1137).  Here the system prompt is prefilled on every rank
    (replicated, 20-odd rows), the DINO + geo-expert prefill - the N (P + 2) rows that dominate the cache - is sharded by view,
    and its per-layer K/V all-gather leaves the FULL cache on every rank (28 672 B per row: 1.26 GB at the 44 k rows of C4,
    nothing next to 288 GB).  So no split-KV decode and no collective per token is needed: `decode_rank` runs the ViT
    images, the question and the greedy decode exactly as the unsharded method does, over a cache whose geo rows came from
    all ranks; the other ranks are free for the next scene.  The ids are broadcast so every rank returns the answer.
    `images`: list of PIL images / paths (as chat_with_recon) or an [N,3,H,W] tensor with image_transform ignoring its
    argument (tests)."""
    dev = model.device
    L = model.dims["llm"]
    past = NaiveCache(L["layers"], L["kv_heads"], dev)
    sys_p = "<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n<|im_start|>user\n"
    gi_text, newlens, new_rope = model.prepare_prompts_pure_text([0], [0], [sys_p], tokenizer, new_token_ids)
    _, newlens, new_rope, _, _ = geo_prefill_view_sharded(model, comm, past, newlens, new_rope,
                                                          list(images) if not torch.is_tensor(images) else images, new_token_ids, kv_overlap,
                                                          text_inputs=gi_text)
    n_ids = torch.zeros(1, dtype=torch.int64, device=dev)
    ids = None
    if comm.rank == decode_rank:
        past, gi = model._chat_suffix(past, newlens, new_rope, tokenizer, new_token_ids, image_transform, images, prompt)
        ids = model.generate_text(past_key_values=past, max_length=max_length, end_token_id=new_token_ids["eos_token_id"], **gi)[:, 0].to(dev)
        n_ids[0] = ids.numel()
    comm.broadcast(n_ids, decode_rank)
    buf = torch.zeros(int(n_ids[0]), dtype=torch.int64, device=dev)
    if comm.rank == decode_rank:
        buf.copy_(ids)
    comm.broadcast(buf, decode_rank)
    ids = buf.cpu()
    return ids if return_ids else tokenizer.decode(ids[1:])


def run_thread_sim(model, world, tokenizer, new_token_ids, images, gather=True, overlap=False, delay_cycles=0, fn=None):
    """Run recon_view_sharded (or `fn(comm)`) on `world` simulated ranks of one device; returns the per-rank results.

    A rank = a host thread with its own stream, as a process would have: the ranks' launches interleave on the device and
    share nothing but the weights and the engine's read-only tables (kernel scratch is owned per (stream, thread),
    hip.scratch_owner).  overlap / delay_cycles: see ThreadSimComm."""
    shared = ThreadSimComm._Shared(world)
    results, errors = [None] * world, []
    on_gpu = model.device.type == "cuda"
    dev = torch.device("cuda", model.device.index if model.device.index is not None else torch.cuda.current_device()) if on_gpu else None
    parent = torch.cuda.current_stream(dev) if on_gpu else None

    def worker(r):
        try:
            comm = ThreadSimComm(shared, r, overlap=overlap, delay_cycles=delay_cycles)
            run = (lambda: fn(comm)) if fn is not None else \
                (lambda: recon_view_sharded(model, comm, tokenizer, new_token_ids, images, gather=gather))
            if on_gpu:
                torch.cuda.set_device(dev)
                s = torch.cuda.Stream(device=dev)
                s.wait_stream(parent)                                     # the caller's inputs were produced on its stream
                with torch.cuda.stream(s):
                    results[r] = run()
                s.synchronize()
            else:
                results[r] = run()
        except BaseException as e:                                        # noqa: BLE001
            errors.append(e)
            shared.barrier.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errors:
        raise errors[0]
    return results
