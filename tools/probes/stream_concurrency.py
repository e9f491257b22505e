import torch, time, os
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(streams, n=40, cyc=2_000_000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        for s in streams:
            with torch.cuda.stream(s):
                torch.cuda._sleep(cyc)
    torch.cuda.synchronize(); return time.perf_counter() - t0
run([s1, s2], 3)
a = run([s1]); b = run([s1, s2])
print(f"sleep kernels: one stream {a*1e3:.1f} ms, two streams {b*1e3:.1f} ms (ratio {b/a:.2f}; 1.0 = concurrent, 2.0 = serialized)")
x = torch.randn(64, 1024, device="cuda")
def work(s):
    with torch.cuda.stream(s):
        y = x
        for _ in range(200): y = torch.tanh(y)
torch.cuda.synchronize(); t0 = time.perf_counter(); work(s1); torch.cuda.synchronize(); a = time.perf_counter() - t0
t0 = time.perf_counter(); work(s1); work(s2); torch.cuda.synchronize(); b = time.perf_counter() - t0
print(f"tiny torch kernels: one stream {a*1e3:.2f} ms, two streams {b*1e3:.2f} ms")
print({k: v for k, v in os.environ.items() if any(t in k for t in ("HIP", "HSA", "AMD", "ROC", "GPU"))})
