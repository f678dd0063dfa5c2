import sys, time, zlib, torch, numpy as np
sys.path.insert(0, '.')
from zlib_rs_amd.engine import Engine, uniform_layout
e = Engine(0)
S, B = 8192, 1 << 20
data = e.gen_shards(S, B)
off, ln = uniform_layout(S, B, e.device)
for adler, crc in ((True, False), (False, True)):
    e.checksums(data, off, ln, adler=adler, crc=crc); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        a, c = e.checksums(data, off, ln, adler=adler, crc=crc)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print("adler" if adler else "crc", "%.2f ms per %d MiB  %.2f TB/s" % (dt * 1e3, S, S * B / dt / 1e12))
a, c = e.checksums(data, off, ln)
h = data[:4 * B].cpu().numpy().tobytes()
for i in range(4):
    assert int(c[i].item()) & 0xFFFFFFFF == zlib.crc32(h[i * B:(i + 1) * B]), i
    assert int(a[i].item()) & 0xFFFFFFFF == zlib.adler32(h[i * B:(i + 1) * B]), i
print("values ok")
