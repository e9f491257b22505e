import torch, torch.nn.functional as F
dev = "cuda"
torch.backends.cudnn.deterministic = True
def bench(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/it
for (B, C, k) in ((50, 252, 5), (50, 256, 5), (100, 252, 5), (100, 256, 5), (50, 252, 3), (50, 256, 3), (100, 256, 3), (25, 256, 5), (25, 256, 3), (50, 254, 5), (50, 254, 3)):
    x = torch.randn(B, C, 16, 16, device=dev); w = torch.randn(C, C, k, k, device=dev) * 0.02; b = torch.randn(C, device=dev)
    t = bench(lambda: F.conv2d(x, w, None, padding=k // 2))
    tb = bench(lambda: F.conv2d(x, w, b, padding=k // 2))
    fl = 2 * B * 256 * C * C * k * k
    print(f"B={B} C={C} k={k}: nobias {t:.3f} ms ({fl/t/1e9:.1f} TF/s)  bias {tb:.3f} ms", flush=True)
# small convs
for (B, ci, co, k) in ((50, 12, 252, 5), (50, 8, 252, 3), (50, 252, 8, 3), (50, 252, 16, 3), (50, 252, 12, 3), (50, 256, 16, 3), (50, 8, 256, 3)):
    x = torch.randn(B, ci, 16, 16, device=dev); w = torch.randn(co, ci, k, k, device=dev) * 0.02
    t = bench(lambda: F.conv2d(x, w, None, padding=k // 2))
    print(f"B={B} {ci}->{co} k={k}: {t*1e3:.1f} us", flush=True)
y = torch.randn(50, 252, 16, 16, device=dev); bb = torch.randn(1, 252, 1, 1, device=dev)
print("elu %.1f us, add bias %.1f us, add %.1f us" % (bench(lambda: F.elu(y))*1e3, bench(lambda: y + bb)*1e3, bench(lambda: y + y)*1e3))
