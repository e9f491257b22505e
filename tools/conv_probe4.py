"""3x3 convs at 200 images: MIOpen vs three strided-batched GEMMs; small-channel convs (heads, input convs)."""
import torch, torch.nn.functional as F
dev = "cuda"
torch.backends.cudnn.deterministic = True
def bench(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/it
for n in (100, 200):
    C = 252
    x = torch.randn(n, C, 16, 16, device=dev); w = torch.randn(C, C, 3, 3, device=dev) * 0.02
    t_ref = bench(lambda: F.conv2d(x, w, None, padding=1))
    ref = F.conv2d(x, w, None, padding=1)
    wd = [w[:, :, dy, :].reshape(C, C * 3).contiguous() for dy in range(3)]
    xp = F.pad(x, (1, 1, 1, 1))
    ax = torch.stack([xp[:, :, :, dx:dx + 16] for dx in range(3)], dim=2).reshape(n, C * 3, 18, 16)
    out = torch.empty(n, C, 256, device=dev)
    def gemms():
        for dy in range(3):
            a = ax[:, :, dy:dy + 16, :].flatten(2)
            if dy == 0: torch.bmm(wd[dy].unsqueeze(0).expand(n, C, C * 3), a, out=out)
            else: out.baddbmm_(wd[dy].unsqueeze(0).expand(n, C, C * 3), a)
        return out
    err = (gemms().view(n, C, 16, 16) - ref).abs().max().item()
    t_g = bench(gemms)
    fl = 2 * n * 256 * C * C * 9
    print(f"3x3 n={n}: MIOpen {t_ref:.3f} ms ({fl/t_ref/1e9:.1f} TF)  3 bmm {t_g:.3f} ms ({fl/t_g/1e9:.1f} TF) + expansion ~{(n*C*3*288*4 + n*C*256*4)/3.2e9:.3f} ms  maxerr {err:.1e}", flush=True)
    for (ci, co, k) in ((252, 16, 3), (252, 12, 3), (8, 252, 3), (12, 252, 5)):
        xs = torch.randn(n, ci, 16, 16, device=dev); ws = torch.randn(co, ci, k, k, device=dev) * 0.02
        t = bench(lambda: F.conv2d(xs, ws, None, padding=k // 2))
        print(f"   n={n} {ci}->{co} k={k}: MIOpen {t*1e3:.1f} us ({2*n*256*ci*co*k*k/t/1e9:.1f} TF)", flush=True)
