import sys, torch
sys.path.insert(0, '/root/repo')
from g2vlm_amd import hip
hip.lib()
torch.manual_seed(1)
for (Lq, Lk) in ((256, 64), (256, 256), (256, 1024), (300, 900)):
    Hq, Hkv, D = 2, 1, 128
    q = torch.randn((Lq, Hq * D), device="cuda").bfloat16()
    k = torch.randn((Lk, Hkv * D), device="cuda").bfloat16()
    v = torch.randn((Lk, Hkv * D), device="cuda").bfloat16()
    o = torch.zeros_like(q)
    plan = hip.make_attn_plan([(0, Lq, 0, Lk, False)], Hq, "cuda", tile_rows=256)
    hip.flash_attn(q, k, v, o, plan, Hq, Hkv, D)
    torch.cuda.synchronize()
    s = torch.einsum("qhd,kd->hqk", q.view(Lq, Hq, D).double(), k.double()) * D ** -0.5
    ref = torch.einsum("hqk,kd->qhd", torch.softmax(s, -1), v.double()).reshape(Lq, Hq * D)
    err = (o.double() - ref)
    print("shape", Lq, Lk, "nan", int(torch.isnan(o).sum()), "of", o.numel(), "n_blocks", plan.n_blocks, "n_comb", plan.n_comb)
    for h in range(Hq):
        for qb0 in range(0, Lq, 32):
            e = err[qb0:qb0 + 32, h * D:(h + 1) * D]
            blocks = [float(torch.nan_to_num(e[:, d:d + 32], nan=99.).abs().max()) for d in range(0, D, 32)]
            print(f"  h{h} rows {qb0:4d}: " + " ".join(f"{b:8.3g}" for b in blocks))
