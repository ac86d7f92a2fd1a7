"""Where a tile of the attention forward goes, in shader cycles per 16-query x 64-key tile and wave (the SMX_ATTN_TRACE build of attention.hip:
    tools/lab/build_variant.sh attntrace attention.hip "-DSMX_ATTN_TRACE=1 -DSMX_TU=attention";  SMX_LIB=tools/lab/libsmx_attntrace.so python tools/gpu_attn_trace.py)
for the tile-staged kernel (SMX_ATTN_V3=0) and the resident-operand kernel (=1).  Phases: 0 K fragment reads + score MFMAs issued, 1 V fragment
reads issued, 2 row maximum (waits for the scores), 3 exponentials + row sums, 4 rescale + second V reads, 5 P V MFMAs issued, 6 everything
between two tile bodies (staging, barrier, loop)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from speechmix_amd import ops, _lib as L
dev = torch.device("cuda:0")
lib = L.lib()
if not hasattr(lib, "smx_attn_trace_read"):
    sys.exit("needs the SMX_ATTN_TRACE build")
B, H, T, D = 32, 12, 499, 64
d = H * D
g = torch.Generator(device="cpu").manual_seed(0)
qkv = (torch.randn(B * T, 3 * d, generator=g) * 0.7).to(dev, torch.bfloat16)
names = ["K reads + S MFMAs", "V reads (1st half)", "row max (S wait)", "exp + row sum", "rescale + V reads (2nd)", "P V MFMAs", "between tiles"]
for v3, drop in (("0", None), ("1", None), ("0", (0.1, 7)), ("1", (0.1, 7))):
    os.environ["SMX_ATTN_V3"] = v3
    desc = ops.AttnDesc(B, H, T, T, D, False, D ** -0.5, drop=drop)
    desc.set("Q", qkv, 0, T * 3 * d, 3 * d); desc.set("K", qkv, d, T * 3 * d, 3 * d); desc.set("V", qkv, 2 * d, T * 3 * d, 3 * d)
    o = torch.empty(B * T, d, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B * H * T, device=dev)
    desc.set("O", o, 0, T * d, d)
    for _ in range(3):
        ops.attention_fwd(desc, lse, ops.BF16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.attention_fwd(desc, lse, ops.BF16)
    e1.record()
    torch.cuda.synchronize()
    buf = np.zeros(64 * 16 * 8, dtype=np.uint64)
    assert lib.smx_attn_trace_read(buf.ctypes.data_as(C.c_void_p), C.c_ulonglong(buf.nbytes)) == 0
    t = buf.reshape(64, 16, 8).astype(np.float64)
    nw = 8 if v3 == "1" else 4
    sel = t[:, :nw, :]
    tiles = sel[..., 7].sum()
    per = sel[..., :7].sum(axis=(0, 1)) / max(tiles, 1)
    print(f"SMX_ATTN_V3={v3} drop={drop}: launch {e0.elapsed_time(e1) * 100:.1f} us (traced build); tiles per wave {sel[..., 7].mean():.1f}; cycles per tile {per.sum():.0f}")
    print("   " + " | ".join(f"{n} {c:.0f}" for n, c in zip(names, per)), flush=True)
