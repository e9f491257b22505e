import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
os.environ["BITSWAP_GEMM_ARITH"] = os.environ.get("ARITH", "bf16x3")
from bitswap_amd import workload, hip
from bitswap_amd.codec import BitSwapCodec, initial_states
model, zend, zcen = workload.build("cifar8", "cuda", quantbits=10)
B, n = 32, 1
images = workload.synthetic_blocks(B * n, model.xs, seed=19).view(B, n, -1).to(torch.int32).cuda()
for fork in ("0", "1"):
    codec = BitSwapCodec(model, zend, zcen, quantbits=10, bitswap=True)
    codec.use_graphs = False; codec.fork = fork
    rec = {}
    orig = codec._net
    phase = ["enc"]
    def wrapped(fn, given, orig=orig, rec=rec, phase=phase):
        out = orig(fn, given)
        rec.setdefault((phase[0],) + fn.bs_key, []).append((given.clone(), out[0].clone(), out[1].clone()))
        return out
    codec._net = wrapped
    state, met = codec.compress(images)
    phase[0] = "dec"
    out = codec.decompress(state, n)
    torch.cuda.synchronize()
    print("fork", fork, "lossless", bool(torch.equal(out, images)), "unwound", state.to_lists() == initial_states(B))
    for key in sorted(k for k in rec if k[0] == "enc"):
        e = rec[key][0]; d = rec[("dec",) + key[1:]][0]
        same_in = torch.equal(e[0], d[0])
        dmu = float((e[1] - d[1]).abs().max()); dsc = float((e[2] - d[2]).abs().max())
        # recompute now, alone, with the encoder's input
        with torch.no_grad():
            fn = (model.infer if key[1] == "infer" else model.generate)(key[2])
            mu2, sc2 = orig(fn, e[0])
        print(key[1:], "same input", same_in, "enc-dec dmu", dmu, "dsc", dsc, "| enc vs recomputed-alone dmu", float((e[1] - mu2).abs().max()), "| dec vs alone", float((d[1] - mu2).abs().max()))
