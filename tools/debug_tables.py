import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O
from bitswap_amd import hip
g = np.load("tests/golden/tables_rans.npz")
for name in ("ztop", "zuni", "x"):
    q = int(g[f"{name}_quantbits"]); e, mu, sc = g[f"{name}_endpoints"], g[f"{name}_mu"], g[f"{name}_scale"]
    K = e.shape[1] + 1
    for ptype in (torch.float32, torch.float64):
        for ld in (K + 1, hip.aligned_ld(K)):
            cdf = hip.logistic_tables(torch.from_numpy(e).cuda(), torch.from_numpy(mu[None]).to(ptype).cuda(), torch.from_numpy(sc[None]).to(ptype).cuda(), 31, q, ld=ld)
            got = cdf.cpu().numpy().view(np.uint32)[0, :, :K + 1].astype(np.int64)
            _, want, _ = O.tables(O.logistic_pmf(e, mu, sc, O.MODE_DET), 31, q)
            want = want.astype(np.int64)
            bad = np.argwhere(got != want)
            print(name, ptype, ld, "mismatches", len(bad))
            if len(bad):
                rows = sorted(set(bad[:, 0]))
                print("  rows", rows[:10])
                r = rows[0]; cols = bad[bad[:, 0] == r][:, 1]
                print("  row", r, "mu", mu[r], "sc", sc[r], "cols", cols[:5], "...", cols[-3:], "n", len(cols))
                c0 = cols[0]
                print("  got", got[r, max(0,c0-2):c0+3], "want", want[r, max(0,c0-2):c0+3])
                fg, fw = np.diff(got[r]), np.diff(want[r])
                d = np.argwhere(fg != fw)[:, 0]
                print("  f diff at", d[:10], "got", fg[d[:10]], "want", fw[d[:10]], "argmax got/want", fg.argmax(), fw.argmax())
