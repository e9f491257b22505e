import torch, time, sys
import torch.nn.functional as F
dev = "cuda"
def bench(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/it
for bench_mode in (False, True):
    torch.backends.cudnn.benchmark = bench_mode
    for (B, C, k) in ((100, 252, 3), (100, 252, 5), (50, 252, 3), (100, 256, 3), (100, 254, 3)):
        x = torch.randn(B, C, 16, 16, device=dev); w = torch.randn(C, C, k, k, device=dev) * 0.02; b = torch.randn(C, device=dev)
        t = bench(lambda: F.conv2d(x, w, b, padding=k // 2))
        fl = 2 * B * 256 * C * C * k * k
        xl = x.contiguous(memory_format=torch.channels_last); wl = w.contiguous(memory_format=torch.channels_last)
        t2 = bench(lambda: F.conv2d(xl, wl, b, padding=k // 2))
        print(f"benchmark={bench_mode} B={B} C={C} k={k}: nchw {t:.3f} ms ({fl/t/1e9:.1f} TF/s)  nhwc {t2:.3f} ms ({fl/t2/1e9:.1f} TF/s)", flush=True)
# im2col + matmul
B, C, k = 100, 252, 3
x = torch.randn(B, C, 16, 16, device=dev); w = torch.randn(C, C * k * k, device=dev) * 0.02
def im2col():
    cols = F.unfold(x, k, padding=1)            # [B, C*9, 256]
    return torch.matmul(w, cols)
t = bench(im2col); print(f"unfold+matmul: {t:.3f} ms")
cols = F.unfold(x, k, padding=1)
t = bench(lambda: torch.matmul(w, cols)); print(f"matmul only: {t:.3f} ms ({2*B*256*C*C*9/t/1e9:.1f} TF/s)")
# elu + bias fused cost
y = torch.randn(B, C, 16, 16, device=dev)
t = bench(lambda: F.elu(y)); print(f"elu: {t*1e3:.1f} us")
