"""The collective half of the C-ABI (include/g2vlm_comm.h, SURVEY 8(b): kv_allgather_{init,run,destroy} wrapping ncclComm_t).

CPU: the library loads and exports every declared symbol.  GPU (one device per box: world = 1): RCCL is loaded, a communicator is
created on the real device, the in-place all-gather / the grouped K+V form / the broadcast run on a torch stream and leave the
buffers intact, and recon_view_sharded through RcclComm equals the unsharded engine bit for bit (one rank = the same kernels).
World > 1 over xGMI needs a multi-GPU node: unmeasured on hardware."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_comm_library_exports_every_declared_symbol():
    from g2vlm_amd import build, comm
    build.build_comm()
    hdr = open(os.path.join(ROOT, "include", "g2vlm_comm.h")).read()
    declared = set(re.findall(r"\bint\s+(g2v_\w+)\s*\(", hdr))
    assert declared == set(comm.EXPORTS) and len(declared) == 8
    lib = comm.lib()
    for name in declared:
        assert getattr(lib, name) is not None


@pytest.mark.gpu
def test_rccl_comm_single_rank_on_device():
    from g2vlm_amd.comm import RcclComm
    comm = RcclComm(1, 0, RcclComm.unique_id(), "cuda:0")
    try:
        assert comm.world == 1 and comm.rank == 0 and comm.overlappable
        k = torch.randn((10, 2, 128), device="cuda").bfloat16()
        v = torch.randn((10, 2, 128), device="cuda").bfloat16()
        k0, v0 = k.clone(), v.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # the stream KVExchange would use
            comm.all_gather_blocks(k, 10)
            comm.all_gather_kv(k, v, 10)
            comm.broadcast(v, 0)
        torch.cuda.current_stream().wait_stream(side)
        comm.barrier()
        assert torch.equal(k, k0) and torch.equal(v, v0)
    finally:
        comm.close()


@pytest.mark.gpu
def test_recon_view_sharded_through_rccl_comm_world_one():
    from oracle import dims as D, synth                    # inputs / fake tokenizer only
    from g2vlm_amd.comm import RcclComm
    from g2vlm_amd.g2vlm_utils import build_model, configs_from_dims
    from g2vlm_amd.sharded import recon_view_sharded
    dims = D.TINY
    model = build_model(*configs_from_dims(dims), synth.synth_state_dict(dims, seed=33), "cuda")
    tok = synth.FakeTokenizer(dims["llm"]["vocab"])
    imgs = synth.synth_images(2, 56, 70, 33)
    ref = model.recon(tok, tok.new_token_ids, None, imgs)
    comm = RcclComm.from_env()
    try:
        res = recon_view_sharded(model, comm, tok, tok.new_token_ids, imgs, gather=True)
        for k in ("points", "local_points", "global_points", "camera_poses", "images"):
            assert torch.equal(res[k], ref[k]), k
    finally:
        comm.close()
