"""Time bs_conv3_wino_f32 alone at the headline's launch size (500 chains, Cin 8 -> 256 channels, 16 x 16 planes) and check it
against torch's conv + the transform it fuses.  Used to compare the product kernel with the lab build -DBS_CONV3_MFMA
(profiles/archive/visits_r06/r06w.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bitswap_amd import hip

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
N, Cin, C = 500, 8, 256
x = torch.randn((N, Cin, 16, 16), generator=g).to(dev)
w = (torch.randn((C, Cin, 3, 3), generator=g) / (Cin * 9) ** 0.5).to(dev)
b = torch.randn((C,), generator=g).to(dev)
h, V = hip.conv3_wino(x, w, b)
want = torch.nn.functional.elu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1))
print("max |h - float64 conv| / max|h|:", float((h.double() - want).abs().max() / want.abs().max()))
for _ in range(5):
    hip.conv3_wino(x, w, b)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(50):
    hip.conv3_wino(x, w, b)
ev[1].record()
torch.cuda.synchronize()
print(f"bs_conv3_wino_f32, {N} blocks: {ev[0].elapsed_time(ev[1]) / 50 * 1e3:.1f} us per launch   lib={hip.load()._name}")
