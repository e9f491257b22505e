"""Drop-in `ANS` class: same constructor and list-in/list-out methods as the reference's
(mnist_compress.py:13-68 and its five verbatim copies), executed by the HIP kernels.

It exists for API compatibility and parity tests (one chain, host round trip per call); the
throughput path is bitswap_amd.codec, which keeps B chains resident in HBM.
"""
import numpy as np
import torch

from . import hip


class ANS:
    def __init__(self, pmfs, bits=31, quantbits=8):
        if not pmfs.is_cuda:
            raise hip.BitswapHipError("ANS needs pmfs on a HIP device (no CPU fallback)")
        self.device = pmfs.device
        self.bits = bits
        self.quantbits = quantbits
        self.mask = (1 << bits) - 1
        self.lbound = 1 << 32
        self.tail_bits = (1 << 32) - 1
        self.seq_len, self.support = pmfs.shape
        self._f, self._cdf, status = hip.table_rows(pmfs, bits, quantbits, ld=self.support + 1)
        # the reference asserts these on the host copy (mnist_compress.py:46-47)
        assert self._cdf.shape == (self.seq_len, self.support + 1)
        assert int(status.abs().max()) == 0, "cdf[:, -1] != 2**bits"

    # the reference exposes int64 numpy tables (mnist_compress.py:43-44)
    @property
    def pmfs(self):
        return self._f.cpu().numpy().view(np.uint32).astype(np.int64)

    @property
    def cdfs(self):
        return self._cdf.cpu().numpy().view(np.uint32).astype(np.int64)

    def encode(self, x, symbols):
        sym = torch.as_tensor([int(s) for s in symbols] if not torch.is_tensor(symbols) else symbols)
        sym = sym.to(self.device, torch.int32).view(1, -1)
        st = hip.RansState.from_lists([x], cap=len(x) + sym.shape[1] + 8, device=self.device)
        hip.rans_push_table(st, self._cdf, sym, self.support, self.bits)
        st.check("ANS.encode")
        x[:] = st.to_lists()[0]
        return x

    def decode(self, x):
        st = hip.RansState.from_lists([x], cap=len(x) + 8, device=self.device)
        sym, _ = hip.rans_pop(st, self._cdf.unsqueeze(0), self.support, self.bits)
        code = int(st.status[0])
        if code == hip.ST_UNDERFLOW:
            raise IndexError("pop from empty list")  # what list.pop raises at mnist_compress.py:66
        st.check("ANS.decode")
        x[:] = st.to_lists()[0]
        return x, sym[0].long()
