import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitswap_amd import hip, workload
from bitswap_amd.codec import BitSwapCodec, HipBackend
name = sys.argv[1] if len(sys.argv) > 1 else "mnist2"
dev = torch.device("cuda")
model, zend, zcen = workload.build(name, dev, quantbits=10)
B, n = 100, 2
images = workload.synthetic_blocks(B * n, model.xs, seed=1000).view(B, n, -1).to(torch.int32).to(dev)
class Dbg(HipBackend):
    def _chk(self, state, what, *ts):
        st = state.status.cpu()
        info = [(float(t.float().min()), float(t.float().max()), bool(torch.isnan(t.float()).any())) for t in ts]
        print(what, "status nonzero:", int((st != 0).sum()), "codes", sorted(set(st.tolist())), "len", int(state.len.min()), int(state.len.max()), info)
    def pop(self, state, cdf, K, bits, centres=None):
        r = super().pop(state, cdf, K, bits, centres); self._chk(state, f"pop K={K} D={r[0].shape[1]}", r[0]); return r
    def push_params(self, state, e, mu, sc, sym, q, bits):
        super().push_params(state, e, mu, sc, sym, q, bits); self._chk(state, f"push_params q={q}", mu, sc, sym)
    def push_table(self, state, cdf, sym, K, bits):
        super().push_table(state, cdf, sym, K, bits); self._chk(state, "push_table", sym)
codec = BitSwapCodec(model, zend, zcen, quantbits=10, bitswap=True, backend=Dbg(dev))
state = codec.new_states(B, n)
for xi in range(n):
    codec.encode_block(state, images[:, xi])
for xi in range(n):
    x = codec.decode_block(state)
    print("decoded equal", torch.equal(x, images[:, n - 1 - xi]))
