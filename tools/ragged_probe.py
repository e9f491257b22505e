import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from bitswap_amd import workload
from bitswap_amd.codec import BitSwapCodec
dev = "cuda"
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
model, zend, zcen = workload.build("imagenetcrop4", dev, quantbits=10, nn_batch=nb)
lens = [3, 2, 2, 1] + [1] * 20
chains = [workload.synthetic_blocks(n, model.xs, seed=60 + i).to(torch.int32) for i, n in enumerate(lens)]
codec = BitSwapCodec(model, zend, zcen, quantbits=10, bitswap=True)
rec = []
orig = codec._net
def wrapped(fn, given):
    out = orig(fn, given); rec.append((given.clone(), out[0].clone(), out[1].expand_as(out[0]).clone())); return out
codec._net = wrapped
state, order, met = codec.compress_ragged(chains)
batched = list(rec); lists = state.to_lists()
for k in (0, 1, 5):
    rec.clear()
    alone, _, _ = codec.compress_ragged([chains[order[k]]])
    same = alone.to_lists()[0] == lists[k]
    first = None
    for j, ((g1, m1, s1), (g2, m2, s2)) in enumerate(zip(rec, batched)):
        if k < g2.shape[0]:
            eq_in = torch.equal(g1[0], g2[k]); eq_mu = torch.equal(m1[0], m2[k]); eq_sc = torch.equal(s1[0], s2[k])
            if not (eq_in and eq_mu and eq_sc):
                first = (j, eq_in, eq_mu, eq_sc, float((m1[0] - m2[k]).abs().max())); break
    print(f"nn_batch {nb} chain sorted#{k} (len {lens[order[k]]}): stream equal {same}; first differing net call {first}", flush=True)
