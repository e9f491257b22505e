import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ["BITSWAP_GEMM_ARITH"] = os.environ.get("ARITH", "bf16x3")
from bitswap_amd import workload, hip
model, zend, zcen = workload.build("cifar8", "cuda", quantbits=10)
model.compress(True)
B = 32
g = torch.Generator().manual_seed(1)
z1 = torch.randn((B, model.zdim_flat), generator=g).cuda()
z0 = torch.randn((B, model.zdim_flat), generator=g).cuda()
names = ["conv3_wino", "wino_fused", "wino_gemm", "wino_gemm_bf16x3", "head_params", "small_k_gemm"]
orig = {n: getattr(hip, n) for n in names}
logs = {}
active = [False]
def wrap(n):
    def f(*a, **k):
        out = orig[n](*a, **k)
        if active[0]:
            outs = out if isinstance(out, tuple) else (out,)
            logs.setdefault(torch.cuda.current_stream().cuda_stream, []).append((n, [o.clone() for o in outs if torch.is_tensor(o)], [tuple(x.shape) if torch.is_tensor(x) else x for x in a][:2]))
        return out
    return f
for n in names:
    setattr(hip, n, wrap(n))
A, S2 = torch.cuda.Stream(), torch.cuda.Stream()
big = torch.randn(8192, 8192, device="cuda")
def run(which):
    logs.clear()
    torch.cuda.synchronize()
    active[0] = True
    with torch.no_grad():
        for _ in range(3):
            big @ big                      # blocker on the default stream (~ms)
        cur = torch.cuda.current_stream()
        A.wait_stream(cur); S2.wait_stream(cur)
        if "a" in which:
            with torch.cuda.stream(A):
                mu, sc = model.infer(1)(z1)
        if "s" in which:
            with torch.cuda.stream(S2):
                mu0, sc0 = model.generate(0)(z0)
    torch.cuda.synchronize()
    active[0] = False
    return {k: list(v) for k, v in logs.items()}
solo_a = run("a")[A.cuda_stream]
solo_s = run("s")[S2.cuda_stream]
for rep in range(6):
    con = run("as")
    for nm, solo, key in (("infer(1)@A", solo_a, A.cuda_stream), ("generate(0)@S", solo_s, S2.cuda_stream)):
        bad = None
        for i, (a, b) in enumerate(zip(solo, con[key])):
            for x, y in zip(a[1], b[1]):
                if not torch.equal(x, y):
                    bad = (i, a[0], a[2], float((x - y).abs().max()), int((x != y).sum()), x.numel())
                    break
            if bad:
                break
        print("rep", rep, nm, "first differing call:", bad, "of", len(solo), flush=True)
