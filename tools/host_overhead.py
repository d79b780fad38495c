"""How long the host needs to ENQUEUE one reconstruction step (no sync) vs the GPU time of the step: the margin that
keeps N ranks on one node GPU-bound (bench.py --gpus N, replicas)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from g2vlm_amd import hip
from g2vlm_amd.g2vlm_utils import build_model, configs_from_dims
from g2vlm_amd.modeling.g2vlm import NaiveCache
from g2vlm_amd.synthetic import REAL_DIMS, SyntheticStateDict
dev = torch.device("cuda", 0)
dims = REAL_DIMS
model = build_model(*configs_from_dims(dims), SyntheticStateDict(dims, dev, seed=0), dev)
tok = B._Tok()
imgs = torch.rand((8, 3, 518, 518))
gi_text, nl, nr = model.prepare_prompts_addbos([0], [0], ["Reconstruct the 3D scene."], tok, B.NEW_TOKEN_IDS)
gi, nl2, nr2 = model.prepare_dino_images_pi3(nl, nr, imgs, None, B.NEW_TOKEN_IDS)
gi["packed_dino_images"] = gi["packed_dino_images"].to(dev); gi["original_images"] = gi["original_images"].to(dev)
def step():
    past = NaiveCache(dims["llm"]["layers"], dims["llm"]["kv_heads"], dev, capacity=8 + 8 * 1371 + 256)
    past = model.forward_cache_update_text(past, **gi_text)
    past, last = model.forward_cache_update_dino(past, **gi)
    return model.reconstruct(past_key_values=past, selected_hidden_states=last, **gi)
for _ in range(2): step()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.1f} ms, until GPU done {1e3*(t2-t0):.1f} ms")
if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3): step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
    st.sort_stats("tottime").print_stats(25)
