import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import ops
def timeit(fn, reps=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1000 / reps
M = 81920
for (N, K) in ((128, 512),):
    a = (torch.rand(M, K, device="cuda") - 0.5).bfloat16()
    w = (torch.rand(N, K, device="cuda") - 0.5).bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    y2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    r = torch.randn(M, N, device="cuda")
    o = torch.empty(M, N, device="cuda")
    d = ops.Dropout(0.1, 5, 3)
    print(N, K, "decode_gemm gelu %.1f us" % timeit(lambda: ops.decode_gemm(a, w, y, M, N, K, gelu=True)),
          "| gemm128 GELU_BF16+drop %.1f us" % timeit(lambda: ops.gemm(a, w, M, N, K, kind=ops.EPI_GELU_BF16, out=y, out2=y2, dropout=d)),
          flush=True)
    for pol in (0, 1):
        ops.gemm_tile_policy(pol)
        print("policy", pol, "GELU_BF16+drop %.1f us" % timeit(lambda: ops.gemm(a, w, M, N, K, kind=ops.EPI_GELU_BF16, out=y, out2=y2, dropout=d)),
              "GELU_BWD+drop %.1f us" % timeit(lambda: ops.gemm(a, w, M, N, K, kind=ops.EPI_GELU_BWD_BF16, out=y, resid=y2, dropout=d)),
              "STORE %.1f us" % timeit(lambda: ops.gemm(a, w, M, N, K, out=y)), "tile", ops.gemm_last_tile(), flush=True)
