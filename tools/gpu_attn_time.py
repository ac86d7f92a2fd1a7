"""Event timing of the encoder-shape attention forward / backward (B 32, H 12, T 499, D 64, bf16, with and without dropout 0.1)
on fused-QKV buffers, as the engine lays them out - A/B builds via SMX_LIB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
dev = torch.device("cuda:0")
B, H, T, D = 32, 12, 499, 64
d = H * D
torch.manual_seed(0)
qkv = torch.randn(B * T, 3 * d, device=dev).bfloat16()
o = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B * H * T, device=dev)
delta = torch.empty(B * H * T, device=dev)
do = torch.randn(B * T, d, device=dev).bfloat16()
dqkv = torch.empty_like(qkv)
for drop in (None, (0.1, 1234)):
    desc = ops.AttnDesc(B, H, T, T, D, False, D ** -0.5, drop=drop)
    for name, off in (("Q", 0), ("K", d), ("V", 2 * d)):
        desc.set(name, qkv, off, T * 3 * d, 3 * d)
        desc.set("d" + name, dqkv, off, T * 3 * d, 3 * d)
    desc.set("O", o, 0, T * d, d)
    desc.set("dO", do, 0, T * d, d)
    for name, fn in (("fwd (+ mask kernel on the first call)", lambda: ops.attention_fwd(desc, lse, ops.BF16)),
                     ("bwd", lambda: ops.attention_bwd(desc, lse, delta, ops.BF16))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record(); torch.cuda.synchronize()
        print(f"dropout={drop is not None} {name}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us", flush=True)
