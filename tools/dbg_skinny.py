import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g2vlm_amd import hip
import torch.nn.functional as F
hip.lib()
def rnd(*s, seed=0, scale=1.0):
    g = torch.Generator(); g.manual_seed(seed); return torch.randn(s, generator=g) * scale
for (M, N, K) in [(48, 768, 256), (48, 256, 256), (48, 1024, 256), (48, 256, 1024), (16, 512, 256), (63, 384, 128), (48, 512, 256)]:
    x, w, b = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2, scale=K ** -0.5).bfloat16(), rnd(N, seed=3, scale=0.1).bfloat16()
    ref = F.linear(x.double(), w.double(), b.double())
    for name, fl in (("skinny", 0), ("small", hip.FORCE_SMALL_TILE)):
        got = hip.linear(x.cuda(), w.cuda(), b.cuda(), hip.EPI_BF16, flags=fl).double().cpu()
        err = (got - ref).abs()
        tol = 2.0 ** -8 * ref.abs().clamp_min(1e-3)
        bad = err > tol
        print(M, N, K, name, "max err", float(err.max()), "bad", int(bad.sum()), "of", bad.numel(),
              "bad rows", sorted(set(bad.nonzero()[:, 0].tolist()))[:8], "bad cols", sorted(set(bad.nonzero()[:, 1].tolist()))[:8])
    res = rnd(M, N, seed=5)
    r2 = res.cuda().clone()
    hip.linear(x.cuda(), w.cuda(), b.cuda(), hip.EPI_RES_F32, out=r2, res=r2)
    r3 = res.cuda().clone()
    hip.linear(x.cuda(), w.cuda(), b.cuda(), hip.EPI_RES_F32, out=r3, res=r3, flags=hip.FORCE_SMALL_TILE)
    print("   res_f32 inplace skinny vs small max diff", float((r2 - r3).abs().max()))
    g1 = hip.linear(x.cuda(), w.cuda(), b.cuda(), hip.EPI_GELU)
    g2 = hip.linear(x.cuda(), w.cuda(), b.cuda(), hip.EPI_GELU, flags=hip.FORCE_SMALL_TILE)
    print("   gelu skinny vs small max diff", float((g1.float() - g2.float()).abs().max()), "n diff", int((g1 != g2).sum()))
