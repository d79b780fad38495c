import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from g2vlm_amd import hip
import test_e2e_gpu as T
from oracle import synth
name = "recon_tiny_3v_56x56"
meta, g = T.load(os.path.join(ROOT, "tests", "golden"), name)
dims = meta["dims"]
model, sd = T.build(dims, meta["seed"])
tok = synth.FakeTokenizer(dims["llm"]["vocab"])
imgs = synth.synth_images(meta["n"], meta["h"], meta["w"], meta["seed"])
outs = {}
eng = model.engine
orig = eng.decoder
cap = {}
def dec(name_, hidden, N, gh, gw, context=None, depth=None):
    r = orig(name_, hidden, N, gh, gw, context=context, depth=depth)
    cap[name_] = r.float().cpu().clone()
    cap[name_ + ".in"] = hidden.float().cpu().clone()
    return r
eng.decoder = dec
for tag, fl in (("skinny", 0), ("small", 2)):
    hip._DEBUG_GEMM_FLAGS = fl
    cap.clear()
    gi, out = T.run_recon(model, tok, imgs)
    outs[tag] = {k: v.float().cpu().clone() for k, v in out.items() if torch.is_tensor(v)}
    outs[tag].update({k: v for k, v in cap.items()})
for k in outs["skinny"]:
    a, b = outs["skinny"][k], outs["small"][k]
    print(k, tuple(a.shape), "rel skinny-vs-small", T.rel(a, b), "maxabs", float((a - b).abs().max()))
prec = T.precise_recon(sd, dims, tok, imgs)
for k in ("local_points", "global_points", "points"):
    print(k, "vs precise: skinny", T.rel(outs["skinny"][k], prec[k]), "small", T.rel(outs["small"][k], prec[k]), "ref", T.rel(g["ref." + k].float(), prec[k]))
a, b = outs["skinny"]["local_points"], outs["small"]["local_points"]
d = (a - b).abs()
print("local diff per view", [float(d[0, v].max()) for v in range(d.shape[1])], "per chan", [float(d[..., c].max()) for c in range(3)])
idx = (d == d.max()).nonzero()[0].tolist(); print("argmax", idx, a[tuple(idx)].item(), b[tuple(idx)].item(), prec["local_points"][tuple(idx)].item())
