#!/bin/bash
# round 6, visit N: the same with r workgroups per CU in all (units spread evenly: no ragged tail), r = 2, 3, 4, 6
# between the non-persistent kernel (every unit pays its prologue and epilogue, other kernels slip in between 17-us workgroups) and
# one workgroup per CU (nothing slips in).  Alone and in the pipeline.
TAG=${1:-r06N}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for k in 2 3 4 6; do BITSWAP_BF16X3_WGS_PER_CU=$k python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from bitswap_amd import hip
def t_us(fn, warm=60, reps=100):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for (T, Cout, Cin, cols) in [(36, 256, 256, 8000), (64, 256, 256, 8000), (36, 256, 256, 1600)]:
    torch.manual_seed(T + cols)
    U = (torch.randn(T, Cout, Cin, device="cuda") * torch.exp(torch.randn(T, 1, Cin, device="cuda"))).contiguous()
    V = (torch.randn(T, Cin, cols, device="cuda") * torch.exp(0.5 * torch.randn(T, Cin, 1, device="cuda"))).contiguous()
    Uf = hip.frags_bf16x3(U)
    out = torch.empty(T, Cout, cols, device="cuda")
    k = os.environ.pop("BITSWAP_BF16X3_WGS_PER_CU")
    os.environ["BITSWAP_BF16X3_PERSISTENT"] = "0"
    ref = hip.wino_gemm_bf16x3(Uf, V, 6).clone()
    t1 = t_us(lambda: hip.wino_gemm_bf16x3(Uf, V, 6, out=out))
    os.environ["BITSWAP_BF16X3_PERSISTENT"] = "1"
    tp = t_us(lambda: hip.wino_gemm_bf16x3(Uf, V, 6, out=out))
    del os.environ["BITSWAP_BF16X3_PERSISTENT"]
    os.environ["BITSWAP_BF16X3_WGS_PER_CU"] = k
    assert torch.equal(hip.wino_gemm_bf16x3(Uf, V, 6), ref)
    tk = t_us(lambda: hip.wino_gemm_bf16x3(Uf, V, 6, out=out))
    print(f"T{T} {Cout}x{Cin}x{cols}: one unit per workgroup {t1:7.1f} us   one workgroup per CU {tp:7.1f} us   {k} workgroups per CU {tk:7.1f} us", flush=True)
PY
done > $OUT/${TAG}_gemm_units_alone.txt 2>&1
cat $OUT/${TAG}_gemm_units_alone.txt
B="python bench.py --steps 6 --warmup 2 --no-extra --no-cpu-baseline --no-roofline --full-record /dev/null"
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"   {d['ms_per_step']:8.2f} ms/step  {d['value']/1e6:6.3f} Mpixel/s  lossless={d['lossless']}")
except Exception as e:
    print("   failed:", e, open(sys.argv[1]).read()[-300:], open('/tmp/o.err').read()[-600:])
PY
}
{
for rep in 1 2 3; do
  echo "1000 chains, default"; timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  for k in 2 3 4 6; do echo "1000 chains, $k workgroups per CU"; BITSWAP_BF16X3_WGS_PER_CU=$k timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
done
echo "imagenet4 1000 chains, default"; timeout 400 $B --workload imagenet4 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
for k in 3 4; do echo "imagenet4 1000 chains, $k workgroups per CU"; BITSWAP_BF16X3_WGS_PER_CU=$k timeout 400 $B --workload imagenet4 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
} > $OUT/${TAG}_gemm_units_pipeline.txt 2>&1
cat $OUT/${TAG}_gemm_units_pipeline.txt
