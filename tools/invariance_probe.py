"""Is a block's (mu, scale) independent of its position in the micro-batch and of its companions?"""
import sys, torch
sys.path.insert(0, "/root/repo")
from bitswap_amd import workload
dev = "cuda"
torch.backends.cudnn.deterministic = True
for nb, algo, minb in ((16, "winograd", 24), (32, "winograd", 24), (16, "winograd", 1), (16, "miopen", 24), (64, "winograd", 24)):
    model, _, _ = workload.build("imagenetcrop4", dev, quantbits=8, nn_batch=None, ppb=1)
    model.compress(True); model.conv_algo = algo; model.gemm_min_batch = minb
    g = torch.Generator().manual_seed(0)
    bad = []
    with torch.no_grad():
        for kind in ("infer", "generate"):
            for i in range(model.nz):
                D = model.xdim if (kind == "infer" and i == 0) else model.zdim_flat
                x = torch.randn((nb, D), generator=g).to(dev)
                fn = getattr(model, kind)(i)
                mu, sc = fn(x)
                perm = torch.randperm(nb, generator=g).to(dev)
                mu2, sc2 = fn(x[perm].contiguous())
                pos = torch.equal(mu[perm], mu2) and torch.equal(sc.expand_as(mu)[perm], sc2.expand_as(mu2))
                xz = torch.zeros_like(x); xz[0] = x[3]
                mu3, sc3 = fn(xz)
                comp = torch.equal(mu3[0], mu[3]) and torch.equal(sc3.expand_as(mu3)[0], sc.expand_as(mu)[3])
                if not (pos and comp):
                    bad.append(f"{kind}({i}): position-invariant={pos} companion-invariant={comp} maxdiff={float((mu3[0]-mu[3]).abs().max()):.2e}")
    print(f"batch {nb} algo {algo} min_batch {minb}: {'OK' if not bad else bad}", flush=True)
