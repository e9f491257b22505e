#!/bin/bash
# round 6, visit Q: grid order of k_wino_fused (BITSWAP_FUSED_ORDER=1: image groups fastest instead of channels fastest): alone and
# in the pipeline
TAG=${1:-r06Q}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python - > $OUT/${TAG}_fused_order_alone.txt 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from bitswap_amd import hip
def t_us(fn, warm=40, reps=100):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for (N, C, H, W, ts) in [(500, 256, 16, 16, 6), (500, 256, 32, 32, 8), (100, 256, 16, 16, 6)]:
    T = (H // 4) * (W // 4)
    torch.manual_seed(N + ts)
    M = torch.randn(ts * ts, C, N * T, device="cuda")
    bias = torch.randn(C, device="cuda")
    res = torch.randn(N, C, H, W, device="cuda")
    out = {}
    for order in ("0", "1"):
        os.environ["BITSWAP_FUSED_ORDER"] = order
        s, a, V = hip.wino_fused(M, (N, C, H, W), ts_in=ts, bias=bias, res=res, act=3, want_sum=True, ts_out=ts)
        if order == "0": ref = (s.clone(), V.clone())
        else: assert torch.equal(s, ref[0]) and torch.equal(V, ref[1])
        out[order] = t_us(lambda: hip.wino_fused(M, (N, C, H, W), ts_in=ts, bias=bias, res=res, act=3, want_sum=True, ts_out=ts))
    print(f"k_wino_fused<{ts},{ts}> N {N} C {C} {H}x{W}: channels fastest {out['0']:7.1f} us   image groups fastest {out['1']:7.1f} us", flush=True)
PY
cat $OUT/${TAG}_fused_order_alone.txt
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
  for o in 0 1; do echo "1000 chains, order $o"; BITSWAP_FUSED_ORDER=$o timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
done
for o in 0 1; do echo "imagenet4 1000 chains, order $o"; BITSWAP_FUSED_ORDER=$o timeout 400 $B --workload imagenet4 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
for o in 0 1; do echo "100 chains, order $o"; BITSWAP_FUSED_ORDER=$o timeout 400 $B --scaling strong --total-chains 100 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
} > $OUT/${TAG}_fused_order_pipeline.txt 2>&1
cat $OUT/${TAG}_fused_order_pipeline.txt
