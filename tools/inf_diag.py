"""diagnostic: deflate S shards, inflate, report first mismatch per stream"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zlib_rs_amd.engine import Engine, uniform_layout
e = Engine(0)
S = int(os.environ.get("DIAG_S", "64")); B = int(os.environ.get("DIAG_B", str(1 << 20)))
data = e.gen_shards(S, B)
off, ln = uniform_layout(S, B, e.device)
out, olen, st = e.deflate_batch(data, off, ln, B, level=6)
back = torch.zeros(S * B, dtype=torch.uint8, device=e.device)
cap = torch.full((S,), B, dtype=torch.int32, device=e.device)
ooff = torch.arange(S, dtype=torch.int64, device=e.device) * B
coff = torch.arange(S, dtype=torch.int64, device=e.device) * out.stride(0)
blen, bst = e.inflate_batch(out, coff, olen, back, ooff, cap)
torch.cuda.synchronize()
print("status nonzero:", [(i, int(x)) for i, x in enumerate(bst.cpu().tolist()) if x][:20])
d = data.view(S, B); b = back.view(S, B)
ne = (d != b)
for i in range(S):
    if bool(ne[i].any()):
        idx = torch.nonzero(ne[i]).flatten()
        k = int(idx[0]); n = int(idx.numel())
        print("stream", i, "class", i % 8, "first diff", k, "ndiff", n, "last diff", int(idx[-1]), "blen", int(blen[i]),
              "got", bytes(b[i, k:k + 12].cpu().tolist()).hex(), "want", bytes(d[i, k:k + 12].cpu().tolist()).hex())
e.close()
