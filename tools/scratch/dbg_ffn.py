import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_ffn as T
from novic_amd import ops
from novic_amd.ops import Dropout
E, K = 512, 128
for M, p, limit in ((61, 0.1, None), (61, 0.0, None), (32, 0.1, None), (8192, 0.0, None), (4096+16, 0.0, None)):
    xmid, g2, gn, w1, w2 = T._inputs(M, seed=M + 3)
    lim = None
    seed = 0x1234567887654321
    ref = T._unfused(xmid, g2, gn, w1, w2, M, p, seed, lim)
    x = torch.zeros(M, E, device="cuda"); ln2, lnn = (torch.zeros(M, E, dtype=torch.bfloat16, device="cuda") for _ in range(2))
    hpre, hact = (torch.zeros(M, K, dtype=torch.bfloat16, device="cuda") for _ in range(2))
    ops.ffn_fwd(xmid, g2, w1, w2, x, M, E, K, gamma_next=gn, ln_next=lnn, ln2=ln2, hpre=hpre, hact=hact, dropout=Dropout(p, seed, 0), site_gelu=7, site_out=8, row_limit=lim)
    torch.cuda.synchronize()
    print("M", M, "p", p)
    for name, got, want in zip(("x", "ln2", "hpre", "hact", "ln_next"), (x, ln2, hpre, hact, lnn), ref):
        d = (got.float() - want.float()).abs()
        bad = d > 0
        if bad.any():
            idx = bad.nonzero()
            rows = idx[:, 0].unique()
            print("  ", name, "mismatches", int(bad.sum()), "max", float(d.max()), "rows", rows[:10].tolist(), "n_rows", len(rows), "cols", idx[:8, 1].tolist())
        else:
            print("  ", name, "equal")
