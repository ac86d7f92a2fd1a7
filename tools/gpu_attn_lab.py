"""Ablation timing of the attention forward kernel (lab switches in attention_v2.h; results are wrong by design)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from tools.gpu_attn_bench import bench
dev = torch.device("cuda:0")
for B in (32, 4):
    H, T, D = 12, 499, 64
    d = H * D
    qkv = (torch.randn(B * T, 3 * d) * 0.7).to(dev, torch.bfloat16)
    o = torch.empty(B * T, d, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B * H * T, device=dev)
    for name, seed in (("full", 0), ("no tile loads", 0xdead0001), ("no loads, no barrier", 0xdead0002)):
        desc = ops.AttnDesc(B, H, T, T, D, False, D ** -0.5)
        desc.p.drop_seed = seed
        desc.set("Q", qkv, 0, T * 3 * d, 3 * d); desc.set("K", qkv, d, T * 3 * d, 3 * d); desc.set("V", qkv, 2 * d, T * 3 * d, 3 * d)
        desc.set("O", o, 0, T * d, d)
        t = bench(lambda: ops.attention_fwd(desc, lse, ops.BF16))
        print(f"B={B} {name}: {t:.1f} us", flush=True)
