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

Collectives go through a small `Comm` interface with two implementations: `TorchDistComm`
(torch.distributed; "nccl" = RCCL on ROCm, "gloo" on CPU) and `ThreadSimComm`, which runs the W
ranks as threads of one process on one GPU so the sharding algebra is tested against the
unsharded engine without an 8-GPU node.
"""
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

    # collectives may be issued on a side stream while the main stream computes (see KVExchange)
    overlappable = True

    def all_gather_blocks(self, full, block_rows):
        """All-gather into `full`: rank r's rows [r*block_rows, (r+1)*block_rows) are already valid on rank r.
        RCCL/NCCL: ONE in-place all_gather_into_tensor - the send block is the rank's own slice of the receive buffer
        (ncclAllGather's in-place form, what FSDP's all-gather uses too), so nothing is staged or copied.
        gloo (CPU rehearsal): list form."""
        mine = full[self.rank * block_rows:(self.rank + 1) * block_rows]
        if self.dist.get_backend(self.group) == "gloo":
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            self.dist.all_gather(parts, mine.clone(), group=self.group)
            for r, p in enumerate(parts):
                if r != self.rank:
                    full[r * block_rows:(r + 1) * block_rows].copy_(p)
        else:
            self.dist.all_gather_into_tensor(full, mine, group=self.group)

    def broadcast(self, t, src):
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
    """W simulated ranks = W threads sharing one device.  Collectives rendezvous on a barrier, with a device
    synchronize first so every rank's kernels have finished before buffers are read across ranks."""

    class _Shared:
        def __init__(self, world):
            self.world, self.barrier, self.slots = world, threading.Barrier(world), {}

    overlappable = False

    def __init__(self, shared, rank):
        self.sh, self.rank, self.world = shared, rank, shared.world

    def _sync(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.sh.barrier.wait()

    def all_gather_blocks(self, full, block_rows):
        self.sh.slots[("ag", self.rank)] = full
        self._sync()
        for r in range(self.world):
            if r != self.rank:
                src = self.sh.slots[("ag", r)]
                full[r * block_rows:(r + 1) * block_rows].copy_(src[r * block_rows:(r + 1) * block_rows])
        self._sync()

    def broadcast(self, t, src):
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
    start() does the exchange synchronously."""

    def __init__(self, comm, past, first_row, total_rows, block_rows, device):
        self.comm, self.past, self.r0, self.n, self.blk = comm, past, first_row, total_rows, block_rows
        use_side = getattr(comm, "overlappable", False) and torch.device(device).type == "cuda" and \
            comm.dist.get_backend(comm.group) != "gloo"
        self.side = torch.cuda.Stream(device=device) if use_side else None
        self.done = None

    def _gather(self, i):
        self.comm.all_gather_blocks(self.past.k[i][self.r0:self.r0 + self.n], self.blk)
        self.comm.all_gather_blocks(self.past.v[i][self.r0:self.r0 + self.n], self.blk)

    def start(self, i):
        if self.side is None:
            self._gather(i)
            return
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            self._gather(i)
            self.done = torch.cuda.Event()
            self.done.record(self.side)

    def wait(self, i):
        if self.side is not None:
            torch.cuda.current_stream().wait_event(self.done)


# ----------------------------------------------------------------------------- the sharded forward
@torch.no_grad()
def recon_view_sharded(model, comm, tokenizer, new_token_ids, images, gather=True):
    """G2VLM.recon (reference g2vlm.py:1240-1303) with views sharded over comm.world ranks.

    `images`: the full [N,3,H,W] tensor in [0,1] (or paths/PIL list) on every rank; N % world == 0.
    Returns this rank's slice of the reference's output dict (keys as G2VLM.recon, leading dims
    [1, n_local, ...]) plus 'view_range'; with gather=True the per-view tensors are all-gathered so every
    rank holds all N views.
    """
    eng, hp, w = model.engine, hip, model.weights
    dev, H = model.device, model.hidden_size
    rank, world = comm.rank, comm.world
    L = model.dims["llm"]

    if model.use_dinov3:
        raise NotImplementedError("view-sharded prefill covers the DINOv2 encoder (its H1 boundary-row exchange is laid out for "
                                  "1 + 4 prefix tokens); the use_dinov3 variant runs unsharded")
    # ---- replicated text prefix (identical on every rank)
    past = NaiveCache(L["layers"], L["kv_heads"], dev)
    gi, newlens, new_rope = model.prepare_prompts_addbos([0], [0], ["Reconstruct the 3D scene."], tokenizer, new_token_ids)
    past = model.forward_cache_update_text(past, **gi)
    T0 = past.length

    # ---- global bookkeeping (host ints), then this rank's subset
    gi, newlens, new_rope = model.prepare_dino_images_pi3(newlens, new_rope, images, None, new_token_ids)
    imgs = gi["packed_dino_images"]
    N, _, Hh, Ww = imgs.shape
    assert N % world == 0, "views must divide evenly over ranks (equal K/V blocks for the in-place all-gather)"
    gh, gw = Hh // 14, Ww // 14
    P, S = gh * gw, gh * gw + 5
    lo, hi = shard_views(N, world, rank)
    nv = hi - lo
    assert 5 * N < P, "boundary exchange assumes a window straddles at most two views"
    blk = nv * (P + 2)                                                   # packed rows per rank
    Lq = N * (P + 2)

    # ---- DINO, sharded by window (H1): flat rows [lo*P, hi*P) (+ the uncovered tail on the last rank)
    va = max(lo - 1, 0)                                                  # first view whose tokens we touch
    x_views = eng.dino_embed(hip.h2d(imgs[va:hi], dev, torch.float32).contiguous())     # [(hi-va)*S, C]
    f0, f1 = lo * P, hi * P + (5 * N if rank == world - 1 else 0)        # global flat rows of this rank
    x_loc = x_views[f0 - va * S: f1 - va * S].contiguous()
    tok = eng.dino_layers(x_loc, nv, P)                                  # bf16 [f1-f0, C]
    tok32 = hp.linear(tok, w["dino2llm.w"], w["dino2llm.b"], hp.EPI_RES_F32)       # fp32 [f1-f0, H]
    # boundary exchange: my first 5*lo rows belong to view lo-1 (rank r-1); I need the next rank's first 5*hi rows
    bmax = 5 * N
    send = torch.zeros((world * bmax, H), dtype=torch.float32, device=dev)
    if lo > 0:
        send[rank * bmax: rank * bmax + 5 * lo].copy_(tok32[:5 * lo])
    comm.all_gather_blocks(send, bmax)
    x = torch.empty((blk, H), dtype=torch.float32, device=dev)           # local MoT rows: [nv*P geo | 2*nv und]
    # patch rows of view v = global flat rows [v*S+5, (v+1)*S)
    for v in range(lo, hi):
        g0, g1 = v * S + 5, (v + 1) * S
        dst = x[(v - lo) * P:(v - lo + 1) * P]
        have1 = min(g1, f1)                                              # rows available locally
        dst[:have1 - g0].copy_(tok32[g0 - f0: have1 - f0])
        if have1 < g1:                                                   # tail lives in the next rank's boundary block
            need = g1 - have1
            assert need == 5 * hi and v == hi - 1
            dst[have1 - g0:].copy_(send[(rank + 1) * bmax:(rank + 1) * bmax + need])

    # ---- MoT geo prefill on local rows; K/V all-gathered per layer
    geo_idx, text_idx = gi["packed_dino_token_indexes"].long(), gi["packed_text_indexes"].long()
    geo_loc = geo_idx[lo * P: hi * P]
    text_loc = text_idx[2 * lo: 2 * hi]
    perm = torch.cat([geo_loc, text_loc])                                # local order, global packed indices
    eng.embed(model._dev_i32(gi["packed_text_ids"][2 * lo: 2 * hi]), x[nv * P:])
    pos = model._dev_i32(gi["packed_position_ids"][:, perm])
    kv_rows = model._dev_i32(gi["packed_indexes"][perm])
    past.reserve(T0 + Lq)

    if world > 1:
        exchange = KVExchange(comm, past, T0, Lq, blk, dev)
        last = eng.llm_forward(x, nv * P, pos, kv_rows, past, T0, causal=False, und_rounding=0, kv_total=T0 + Lq, kv_exchange=exchange,
                               local_kv=(T0 + rank * blk, blk))
    else:                                                                # one rank: nothing to exchange, one attention launch
        last = eng.llm_forward(x, nv * P, pos, kv_rows, past, T0, causal=False, und_rounding=0, kv_total=T0 + Lq)
    hidden = last[:nv * P].contiguous()                                  # geo rows of my views, view-major

    # ---- decoders: local views; the global decoder's context is view 0 (rank 0)
    ctx = torch.empty((P, H), dtype=torch.float32, device=dev)
    if rank == 0:
        ctx.copy_(hidden[:P])
    comm.broadcast(ctx, 0)
    point_hidden = eng.decoder("point_decoder", hidden, nv, gh, gw)
    camera_hidden = eng.decoder("camera_decoder", hidden, nv, gh, gw)
    global_hidden = eng.decoder("global_points_decoder", hidden, nv, gh, gw, context=ctx)
    points, local, poses, glob = eng.heads(point_hidden, camera_hidden, global_hidden, nv, Hh, Ww)
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
    return res


def run_thread_sim(model, world, tokenizer, new_token_ids, images, gather=True):
    """Run recon_view_sharded on `world` simulated ranks (threads) of one device; returns the list of per-rank results."""
    shared = ThreadSimComm._Shared(world)
    results, errors = [None] * world, []

    def worker(r):
        try:
            if model.device.type == "cuda" and model.device.index is not None:
                torch.cuda.set_device(model.device)
            results[r] = recon_view_sharded(model, ThreadSimComm(shared, r), tokenizer, new_token_ids, images, gather=gather)
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
