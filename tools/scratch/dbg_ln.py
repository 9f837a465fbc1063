import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_ffn as T
from novic_amd import ops
E, K = 512, 128
M = 8192
xmid, g2, gn, w1, w2 = T._inputs(M, seed=M + 3)
ln_ref = torch.zeros(M, E, dtype=torch.bfloat16, device="cuda")
ops.layernorm_fwd(xmid, g2, ln_ref, M, E)
ln32 = torch.zeros(M, E, device="cuda")
ops.layernorm_fwd(xmid, g2, None, M, E, out_f32=ln32)
x = torch.zeros(M, E, device="cuda"); ln2 = torch.zeros(M, E, dtype=torch.bfloat16, device="cuda")
ops.ffn_fwd(xmid, g2, w1, w2, x, M, E, K, ln2=ln2)
torch.cuda.synchronize()
bad = (ln2 != ln_ref).nonzero()
print("mismatches", len(bad))
xd = xmid.double().cpu(); gd = g2.double().cpu()
for r, c in bad.tolist()[:8]:
    row = xd[r]; mean = row.mean(); var = ((row - mean) ** 2).mean(); val = (row[c] - mean) / torch.sqrt(var + 1e-5) * gd[c]
    print(r, c, "ffn", float(ln2[r, c]), "ref", float(ln_ref[r, c]), "ref_f32", float(ln32[r, c]), "exact", float(val), "tile", r // 16, "row_in_tile", r % 16, "lane", (c % 256) // 4, "e", c % 4)
