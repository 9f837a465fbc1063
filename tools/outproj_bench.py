import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import ops
M = 61500
a = (torch.rand(M, 512, device="cuda") * 2 - 1).to(torch.bfloat16); w = (torch.rand(512, 512, device="cuda") * 0.1).to(torch.bfloat16)
rs = torch.randn(M, 512, device="cuda"); out = torch.empty(M, 512, device="cuda")
for pol in (0, 1):
    ops.gemm_tile_policy(pol)
    for _ in range(3): ops.gemm(a, w, M, 512, 512, kind=ops.EPI_RESID_F32, out=out, resid=rs, dropout=ops.Dropout(0.1, 1, 2))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(20): ops.gemm(a, w, M, 512, 512, kind=ops.EPI_RESID_F32, out=out, resid=rs, dropout=ops.Dropout(0.1, 1, 2))
    e.record(); torch.cuda.synchronize()
    print("policy", pol, "%.1f us" % (s.elapsed_time(e) / 20 * 1000))
ops.gemm_tile_policy(1)
