import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ["BITSWAP_GEMM_ARITH"] = os.environ.get("ARITH", "bf16x3")
from bitswap_amd import workload, hip
from bitswap_amd.codec import BitSwapCodec, initial_states
model, zend, zcen = workload.build("cifar8", "cuda", quantbits=10)
B, n = 32, 1
images = workload.synthetic_blocks(B * n, model.xs, seed=19).view(B, n, -1).to(torch.int32).cuda()
mode = os.environ.get("MODE", "none")
codec = BitSwapCodec(model, zend, zcen, quantbits=10, bitswap=True)
codec.use_graphs = False; codec.fork = "1"
stash = []
if mode == "keepall":
    # every tensor any hip.* call or torch op allocates stays alive until the run ends: no block is ever reused
    real_empty = torch.empty
    def empty(*a, **k):
        t = real_empty(*a, **k); stash.append(t); return t
    torch.empty = empty
    orig = codec._net
    def wrapped(fn, given):
        out = orig(fn, given); stash.extend(out); stash.append(given); return out
    codec._net = wrapped
ok = []
side = torch.cuda.Stream()
for rep in range(3):
    if mode == "side":
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            state, met = codec.compress(images)
            out = codec.decompress(state, n)
    else:
        state, met = codec.compress(images)
        out = codec.decompress(state, n)
    torch.cuda.synchronize()
    ok.append(bool(torch.equal(out, images)))
print("mode", mode, os.environ.get("GPU_MAX_HW_QUEUES", ""), "lossless", ok, "stash", len(stash))
