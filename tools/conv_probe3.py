"""5x5 conv as x-expansion + 5 strided-batched GEMMs (one per kernel row) against MIOpen."""
import torch, torch.nn.functional as F
dev = "cuda"
torch.backends.cudnn.deterministic = True
def bench(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/it
for n in (13, 25, 50, 100, 200):
    C = 252
    x = torch.randn(n, C, 16, 16, device=dev); w = torch.randn(C, C, 5, 5, device=dev) * 0.02
    t_ref = bench(lambda: F.conv2d(x, w, None, padding=2))
    ref = F.conv2d(x, w, None, padding=2)
    # weights per kernel row: W_dy [co, (ci, dx)]
    wd = [w[:, :, dy, :].reshape(C, C * 5).contiguous() for dy in range(5)]
    def expand():
        xp = F.pad(x, (2, 2, 2, 2))                                   # [n, C, 20, 20]
        ax = torch.stack([xp[:, :, :, dx:dx + 16] for dx in range(5)], dim=2)   # [n, C, 5, 20, 16]
        return ax.reshape(n, C * 5, 20, 16)
    ax = expand()
    t_exp = bench(expand)
    out = torch.empty(n, C, 256, device=dev)
    def gemms():
        for dy in range(5):
            a = ax[:, :, dy:dy + 16, :].reshape(n, C * 5, 256) if False else ax[:, :, dy:dy + 16, :].flatten(2)
            if dy == 0:
                torch.bmm(wd[dy].unsqueeze(0).expand(n, C, C * 5), a, out=out)
            else:
                out.baddbmm_(wd[dy].unsqueeze(0).expand(n, C, C * 5), a)
        return out
    o = gemms().view(n, C, 16, 16)
    err = (o - ref).abs().max().item()
    t_g = bench(gemms)
    fl = 2 * n * 256 * C * C * 25
    print(f"n={n}: MIOpen {t_ref:.3f} ms ({fl/t_ref/1e9:.1f} TF)  expand(torch) {t_exp:.3f} ms  5 bmm {t_g:.3f} ms ({fl/t_g/1e9:.1f} TF)  maxerr {err:.2e}", flush=True)
    # one big GEMM with a [K, n*256] operand for comparison (layout that a custom expansion could write)
    axk = ax.permute(1, 0, 2, 3).contiguous()                          # [K5, n, 20, 16]
    big = torch.randn(C * 5, n * 256, device=dev)
    t_big = bench(lambda: torch.mm(wd[0], big))
    print(f"      one [252 x 1260] @ [1260 x {n*256}] GEMM: {t_big:.3f} ms ({2*C*C*5*n*256/t_big/1e9:.1f} TF)", flush=True)
