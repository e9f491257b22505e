#!/bin/bash
# BASELINE.json configs 2, 3, 5 at the reference's own shape (100 experiments x 100 blocks, mnist_compress.py:102-103)
# through the reference-named scripts, sender + receiver with the lossless / state-restored asserts, synthetic weights.
#   gpurun --timeout 1500 -- 'bash tools/config_runs.sh > gpurun_out/r02k_configs.txt 2>&1'
T=$(mktemp -d)
echo "== config 2: cifar_compress.py --nz 8 --bitswap 1 (100 chains x 100 blocks)"
python cifar_compress.py --synthetic --decompress 1 --outdir $T 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -3
echo "== config 3: imagenet_compress.py (nz 2 and 4, 100 chains x 100 blocks = 10k blocks each)"
python imagenet_compress.py --synthetic --decompress 1 --outdir $T 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -6
echo "== config 5 (one GPU's view): imagenet_compress.py --bitswap 0 (BB-ANS)"
python imagenet_compress.py --synthetic --decompress 1 --bitswap 0 --outdir $T 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -6
echo "== config 1 shape on the GPU: mnist_compress.py --nz 2"
python mnist_compress.py --synthetic --decompress 1 --outdir $T 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -3
