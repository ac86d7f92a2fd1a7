"""colsum: single-stage atomic vs two-stage, per shape (us, GB/s)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from tools.gpu_check_pp import bench
dev = torch.device("cuda:0")
for (M, N) in ((15968, 768), (15968, 2304), (15968, 3072), (7968, 768), (7968, 3072), (511968, 512), (1024, 768)):
    x = torch.randn(M, N, device=dev).bfloat16()
    out = torch.zeros(N, dtype=torch.float32, device=dev)
    os.environ["SMX_COLSUM"] = "atomic"
    t0 = bench(lambda: ops.colsum(x, out, M, N, N, ops.BF16), n=20)
    os.environ["SMX_COLSUM"] = ""
    t1 = bench(lambda: ops.colsum(x, out, M, N, N, ops.BF16), n=20)
    print(f"M={M} N={N}: atomic {t0:.1f} us ({M*N*2/t0/1e3:.0f} GB/s)   two-stage {t1:.1f} us ({M*N*2/t1/1e3:.0f} GB/s)", flush=True)
