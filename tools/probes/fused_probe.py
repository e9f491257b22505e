import torch, sys, os
sys.path.insert(0, os.getcwd())
from bitswap_amd import hip
M = torch.randn(36, 256, 6400, device="cuda"); x = torch.randn(400, 256, 16, 16, device="cuda"); b = torch.randn(256, device="cuda")
M8 = torch.randn(64, 256, 6400, device="cuda")
def t(fn, n=300):
    for _ in range(50): fn()
    torch.cuda.synchronize(); a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return a.elapsed_time(e) / n * 1e3
f66 = lambda: hip.wino_fused(M, (400, 256, 16, 16), 6, b, x, True, ts_out=6)
f66s = lambda: hip.wino_fused(M, (400, 256, 16, 16), 6, b, x, True, want_sum=True, ts_out=6)
f88 = lambda: hip.wino_fused(M8, (400, 256, 16, 16), 8, b, x, True, ts_out=8)
print("fused<6,6> %.1f us (%.2f TB/s) | <6,6>+sum %.1f us | <8,8> %.1f us (%.2f TB/s)" % (t(f66), (2*36*256*6400*4+105e6)/t(f66)/1e6, t(f66s), t(f88), (2*64*256*6400*4+105e6)/t(f88)/1e6))
