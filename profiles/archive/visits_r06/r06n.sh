#!/bin/bash
# round 6, visit n: two producer wavefronts per workgroup (384 threads: two SIMDs of a CU keep 288 free registers) against four
TAG=${1:-r06n}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_codec_gpu.py -m gpu -q -x -k "bf16x3_gemm" > $OUT/${TAG}_pytest.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/${TAG}_pytest.log
python - > $OUT/${TAG}_gemm_times.txt 2>&1 <<'PY'
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
for (T, Cout, Cin, cols) in [(36, 256, 256, 8000), (64, 256, 256, 8000), (36, 256, 256, 1600), (64, 256, 256, 1600), (36, 256, 256, 400), (36, 256, 256, 208)]:
    torch.manual_seed(T + cols)
    U = (torch.randn(T, Cout, Cin, device="cuda") * torch.exp(torch.randn(T, 1, Cin, device="cuda"))).contiguous()
    V = (torch.randn(T, Cin, cols, device="cuda") * torch.exp(0.5 * torch.randn(T, Cin, 1, device="cuda"))).contiguous()
    Uf = hip.frags_bf16x3(U)
    out = torch.empty(T, Cout, cols, device="cuda")
    fl = 2.0 * T * Cout * Cin * cols
    row = {}
    ref = None
    for name, pers, ring in (("ws 4 producers", "0", "4"), ("ws 2 producers", "0", "2"), ("persistent 4 producers", "1", "4"), ("persistent 2 producers", "1", "2")):
        os.environ["BITSWAP_BF16X3_SHAPE"], os.environ["BITSWAP_BF16X3_PERSISTENT"], os.environ["BITSWAP_BF16X3_PRODUCERS"] = "3", pers, ring
        m = hip.wino_gemm_bf16x3(Uf, V, 6).clone()
        ref = m if ref is None else ref
        assert torch.equal(m, ref), name
        row[name] = t_us(lambda: hip.wino_gemm_bf16x3(Uf, V, 6, out=out))
    print(f"T{T} {Cout}x{Cin}x{cols}: " + "  ".join(f"{k} {v:7.1f} us ({fl / v / 1e6:6.1f} TF-eq)" for k, v in row.items()), flush=True)
PY
cat $OUT/${TAG}_gemm_times.txt
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
  for ring in 4 2; do echo "producers $ring, 1000 chains"; BITSWAP_BF16X3_PRODUCERS=$ring timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
done
for ring in 4 2; do echo "producers $ring, imagenet4 1000 chains"; BITSWAP_BF16X3_PRODUCERS=$ring timeout 400 $B --workload imagenet4 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
for ring in 4 2; do echo "producers $ring, 100 chains"; BITSWAP_BF16X3_PRODUCERS=$ring timeout 400 $B --scaling strong --total-chains 100 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
} > $OUT/${TAG}_pipeline_ab.txt 2>&1
cat $OUT/${TAG}_pipeline_ab.txt
