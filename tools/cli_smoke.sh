#!/bin/bash
# End-to-end run of the reference-named scripts on one GPU with synthetic weights (no datasets/checkpoints offline).
#   gpurun --timeout 900 -- 'bash tools/cli_smoke.sh'
set -e
T=$(mktemp -d)
python - <<PY
import numpy as np
from PIL import Image
rng = np.random.RandomState(0)
yy, xx = np.mgrid[0:150, 0:200]
img = np.stack([127 + 80 * np.sin(yy / (9. + c)) * np.cos(xx / (11. + c)) for c in range(3)], -1) + rng.randn(150, 200, 3) * 5
Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save("$T/pic.png")
PY
for fmt in reference wave64; do
  python demo_compress.py --image $T/pic.png --gpu 0 --synthetic --format $fmt | tail -1
  python demo_decompress.py --file $T/pic_bitswap.npy --gpu 0 --synthetic | tail -2
done
python cifar_compress.py --synthetic --experiments 6 --ndatapoints 3 --decompress 1 --outdir $T/out | tail -2
cat $T/out/bitstreams/cifar/nz8/Bit-Swap/stream_meta.json | tr -d '\n' | cut -c1-400; echo
# a receiver in another process: reads the pickles + stream_meta.json, refuses a CDF-spec mismatch, then decodes
python cifar_compress.py --synthetic --decompress-only --outdir $T/out | tail -1
if python cifar_compress.py --synthetic --decompress-only --cdf-spec 1 --outdir $T/out > $T/mismatch.log 2>&1; then echo "MISMATCH NOT REFUSED"; exit 1; else grep -o "StreamMismatch.*" $T/mismatch.log | cut -c1-160; fi
python cifar_compress.py --synthetic --experiments 6 --ndatapoints 3 --decompress 1 --format wave64 --outdir $T/out64 | tail -1
python mnist_compress.py --synthetic --nz 2 --bitswap 0 --experiments 4 --ndatapoints 2 --decompress 1 --outdir $T/out | tail -1
python imagenetcrop_compress.py --synthetic --nimages 12 | tail -4
echo CLI_SMOKE_OK
