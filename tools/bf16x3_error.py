#!/usr/bin/env python3
"""Error table of the bf16x3 conv arithmetic (VERDICT r3 #5 / r5 #2c: evidence before any promotion).  Full-width models of ALL
FOUR workloads (--workload cifar8 | imagenet4 | mnist2 | imagenetcrop4 | all; round 4 measured cifar8 only),
32 blocks per call, every infer(i) / generate(i) stack: (mu, scale) of the fp32-MFMA route and of the bf16x3 route(s) against
a float64 evaluation of the plain torch modules (same folded weights); and what the difference means for the rate -- the
ideal code length of the same symbols under the two routes' integer tables (HIP table kernels, CDF spec 2).
    python tools/bf16x3_error.py [out.json] [--workload all]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitswap_amd import hip, workload  # noqa: E402
from bitswap_amd.bins import uniform_step  # noqa: E402


def build(arith, name):
    os.environ["BITSWAP_GEMM_ARITH"] = arith
    m, zend, zcen = workload.build(name, "cuda", quantbits=10)
    assert m.gemm_arith == arith
    m.compress(True)
    return m, zend


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out_path = args[0] if args else "/tmp/bf16x3_error.json"
    wl = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "cifar8"
    names = ["cifar8", "imagenet4", "mnist2", "imagenetcrop4"] if wl == "all" else [wl]
    res = {name: one_workload(name) for name in names}
    ok = all(r["summary"]["bf16x3"]["max_abs_ideal_bits_per_dim_delta"] <= 1e-4 for r in res.values())
    json.dump({"what": "full width, 32 blocks per call, every stack against a float64 evaluation of the torch modules; "
                       "ideal code length of the same symbols under each route's integer tables (default CDF spec)",
               "bits_per_dim_within_1e-4_of_the_fp32_route_everywhere": ok, "workloads": res}, open(out_path, "w"), indent=1)
    for name, r in res.items():
        print(name, json.dumps(r["summary"]))


def one_workload(name):
    routes = {a: build(a, name) for a in ("fp32", "bf16x3", "bf16x3x9")}
    base, zend = routes["fp32"]
    # float64 reference: the same module tree in double precision, unfused
    import copy
    ref = copy.deepcopy(base).double()
    ref.fused = False
    ref.fold()
    ref.compress(False)          # (compress mode casts its input to float32: call the stacks directly)
    N = 32
    g = torch.Generator().manual_seed(5)
    rows = []
    for i in range(base.nz):
        x = ((torch.randint(0, 256, (N, base.xdim), generator=g).float() - 127.5) / 127.5).cuda()
        z = torch.randn((N, base.zdim_flat), generator=g).cuda()
        for kind in ("infer", "generate"):
            inp = x if (kind == "infer" and i == 0) else z
            with torch.no_grad():
                h64 = inp.double().view((-1,) + (ref.xs if (kind == "infer" and i == 0) else ref.zdim))
                mu64, sc64 = (ref._infer_stack if kind == "infer" else ref._gen_stack)(i, h64)
                mu64, sc64 = mu64.reshape(N, -1), sc64.expand_as(mu64).reshape(N, -1)
                rng = float(mu64.abs().max())
                row = {"stack": f"{kind}({i})", "mu_range": rng}
                outs = {}
                for a, (m, _) in routes.items():
                    mu, sc = getattr(m, kind)(i)(inp)
                    sc = sc.expand_as(mu)
                    outs[a] = (mu, sc)
                    row[a] = {"max_dmu_over_range": float((mu.double() - mu64).abs().max()) / rng,
                              "rms_dmu_over_range": float((mu.double() - mu64).pow(2).mean().sqrt()) / rng,
                              "max_dscale_rel": float(((sc.double() - sc64) / sc64).abs().max())}
                # rate: ideal code length of the same symbols (the bin of mu64, +- a few bins) under each route's tables
                if kind == "infer" and i < base.nz - 1 or kind == "generate" and i > 0:
                    li = i if kind == "infer" else i - 1
                    e = zend[li]
                    step = torch.from_numpy(uniform_step(e.cpu().numpy())).cuda()
                    K = e.shape[1] + 1
                    centre = (e[None, :, :] < mu64[:, :, None]).sum(-1).clamp(0, K - 1).to(torch.int32)
                    off = torch.randint(-6, 7, centre.shape, generator=g).cuda().to(torch.int32)
                    sym = (centre + off).clamp(0, K - 1).contiguous()
                    bits = {}
                    for a, (mu, sc) in outs.items():
                        st = torch.zeros(N, dtype=torch.int32, device="cuda")
                        f, _ = hip.logistic_fc(e, mu.contiguous(), sc.contiguous(), sym, st, 31, 10, step=step)
                        bits[a] = 31.0 - torch.log2(f.double())
                    for a in ("bf16x3", "bf16x3x9"):
                        row[a]["ideal_bits_per_dim_minus_fp32_route"] = float((bits[a] - bits["fp32"]).mean())
                        row[a]["max_abs_bits_per_symbol_minus_fp32_route"] = float((bits[a] - bits["fp32"]).abs().max())
                rows.append(row)
                print(json.dumps(row), flush=True)
    summary = {a: {"max_dmu_over_range": max(r[a]["max_dmu_over_range"] for r in rows),
                   "rms_dmu_over_range": float(np.sqrt(np.mean([r[a]["rms_dmu_over_range"] ** 2 for r in rows]))),
                   "max_dscale_rel": max(r[a]["max_dscale_rel"] for r in rows),
                   "max_abs_ideal_bits_per_dim_delta": max((abs(r[a].get("ideal_bits_per_dim_minus_fp32_route", 0.0)) for r in rows), default=0.0)}
               for a in routes}
    nfr = {a: len(m._ufrags) for a, (m, _) in routes.items()}
    return {"summary": summary, "products_on_bf16x3": nfr, "stacks": rows}


if __name__ == "__main__":
    main()
