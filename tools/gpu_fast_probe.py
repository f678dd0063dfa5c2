#!/usr/bin/env python3
"""Torch-free GPU probe (ctypes over libamdhip64 + the engine library): per-kernel times, ratio and a device round trip
per level, in a few seconds -- no `import torch` (1-2 minutes on a fresh box).  For kernel A/B work:

    ZMI_LIB=variants/libzmi355_x.so python tools/gpu_fast_probe.py --shards 2048 --levels 1,6,9 --real --classes

--real adds the three real fixtures (lcet10.txt, paper-100k.pdf, fireworks.jpg, each tiled to 1 MiB) with their ratios.
Every deflate run is verified on the device: the streams are inflated back and the Adler-32 of every shard is compared
with that of its input (zmi_checksum_batch_dev both sides); --host-verify N additionally expands N streams with Python's zlib.
"""
import argparse
import ctypes as C
import json
import lzma
import os
import sys
import time
import zlib

os.environ.setdefault("ZMI_TUNING", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]


def ck(rc, what):
    if rc != 0:
        raise RuntimeError("%s: hip error %d" % (what, rc))


def dmalloc(n):
    p = C.c_void_p()
    ck(hip.hipMalloc(C.byref(p), max(int(n), 16)), "hipMalloc(%d)" % n)
    return p.value


def h2d(dst, b):
    ck(hip.hipMemcpy(dst, C.c_char_p(bytes(b)) if not isinstance(b, C.Array) else b, len(b) if not isinstance(b, C.Array) else C.sizeof(b), 1), "h2d")


def d2h(src, n):
    buf = (C.c_uint8 * n)()
    ck(hip.hipMemcpy(buf, src, n, 2), "d2h")
    return bytes(buf)


def d2h_u32(src, n):
    buf = (C.c_uint32 * n)()
    ck(hip.hipMemcpy(buf, src, 4 * n, 2), "d2h")
    return list(buf)


def load_lib():
    path = os.environ.get("ZMI_LIB", os.path.join(ROOT, "zlib_rs_amd", "libzmi355.so"))
    L = C.CDLL(os.path.abspath(path))
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.zmi_last_error.restype = C.c_char_p
    L.zmi_ctx_create.argtypes = [C.POINTER(vp), i32]
    L.zmi_ctx_set_scratch_limit.argtypes = [vp, u64]
    L.zmi_ctx_set_inflate_out_limit.argtypes = [vp, u64]
    L.zmi_ctx_set_timing.argtypes = [vp, i32]
    L.zmi_ctx_get_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(u32)]
    L.zmi_deflate_bound.restype = u64
    L.zmi_deflate_bound.argtypes = [u64, i32]
    L.zmi_deflate_batch_dev.argtypes = [vp, vp, vp, vp, u32, u32, i32, i32, i32, vp, u64, vp, vp, vp]
    L.zmi_inflate_batch_dev.argtypes = [vp, vp, vp, vp, u32, i32, vp, vp, vp, vp, vp, vp]
    L.zmi_checksum_batch_dev.argtypes = [vp, vp, vp, vp, u32, i32, vp, vp, vp]
    L.zmi_gen_shards_dev.argtypes = [vp, vp, u64, u32, u32, u32, vp]
    return L, path


def timing(L, ctx):
    sums = (C.c_double * 8)()
    cnts = (C.c_uint32 * 8)()
    L.zmi_ctx_get_timing(ctx, sums, cnts)
    return list(sums)


def real_fixtures(B):
    d = os.path.join(ROOT, "tests", "golden", "fixtures")
    out = []
    for name in ("lcet10.txt", "paper-100k.pdf", "fireworks.jpg"):
        raw = lzma.decompress(open(os.path.join(d, name + ".xz"), "rb").read())
        out.append((name, (raw * (B // len(raw) + 1))[:B]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=2048)
    ap.add_argument("--levels", default="6")
    ap.add_argument("--real", action="store_true")
    ap.add_argument("--classes", action="store_true")
    ap.add_argument("--host-verify", type=int, default=4)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--tag", default="")
    ap.add_argument("--groups", type=int, default=1, help="cut the batch into this many launch groups (scratch = 1/groups of the batch)")
    ap.add_argument("--no-check", action="store_true", help="ablation builds (output invalid on purpose): times only, no status / round-trip checks")
    ap.add_argument("--class-times", action="store_true", help="also time the eight data classes of the generator separately (level 6 unless --levels has one entry)")
    a = ap.parse_args()
    L, path = load_lib()
    ctx = C.c_void_p()
    if L.zmi_ctx_create(C.byref(ctx), 0) != 0:
        sys.exit("zmi_ctx_create: " + L.zmi_last_error().decode())
    B = 1 << 20
    S = a.shards
    fx = real_fixtures(B) if a.real else []
    NT = S + len(fx)
    L.zmi_ctx_set_scratch_limit(ctx, max(64 << 20, NT * B * 4 // max(1, a.groups)))
    L.zmi_ctx_set_inflate_out_limit(ctx, NT * B + (1 << 20))
    L.zmi_ctx_set_timing(ctx, 1)
    stride = int(L.zmi_deflate_bound(B, 1))
    d_in = dmalloc(NT * B)
    d_out = dmalloc(NT * stride)
    d_back = dmalloc(NT * B)
    d_off = dmalloc(NT * 8); d_len = dmalloc(NT * 4); d_olen = dmalloc(NT * 4); d_st = dmalloc(NT * 4)
    d_coff = dmalloc(NT * 8); d_cap = dmalloc(NT * 4); d_blen = dmalloc(NT * 4); d_bst = dmalloc(NT * 4)
    d_a0 = dmalloc(NT * 4); d_a1 = dmalloc(NT * 4); d_c = dmalloc(NT * 4)
    h2d(d_off, (C.c_uint64 * NT)(*[i * B for i in range(NT)]))
    h2d(d_len, (C.c_uint32 * NT)(*([B] * NT)))
    h2d(d_coff, (C.c_uint64 * NT)(*[i * stride for i in range(NT)]))
    h2d(d_cap, (C.c_uint32 * NT)(*([B] * NT)))
    for s0 in range(0, S, 16384):
        L.zmi_gen_shards_dev(ctx, d_in + s0 * B, 0x5A4C4942, s0, min(16384, S - s0), B, None)
    for i, (_, raw) in enumerate(fx):
        h2d(d_in + (S + i) * B, raw)
    hip.hipDeviceSynchronize()
    L.zmi_checksum_batch_dev(ctx, d_in, d_off, d_len, NT, 1, d_a0, d_c, None)
    hip.hipDeviceSynchronize()
    want = d2h_u32(d_a0, NT)
    print("# lib %s  shards %d (+%d real)  %s" % (os.path.relpath(path, ROOT), S, len(fx), a.tag))
    res = {"lib": os.path.relpath(path, ROOT), "shards": S, "tag": a.tag, "levels": {}}
    for lvl in [int(x) for x in a.levels.split(",")]:
        best = None
        for rep in range(a.reps):
            timing(L, ctx)
            t = time.perf_counter()
            rc = L.zmi_deflate_batch_dev(ctx, d_in, d_off, d_len, NT, B, lvl, 0, 1, d_out, stride, d_olen, d_st, None)
            if rc != 0:
                sys.exit("deflate: " + L.zmi_last_error().decode())
            hip.hipDeviceSynchronize()
            dt = time.perf_counter() - t
            tm = timing(L, ctx)
            if best is None or dt < best[0]:
                best = (dt, tm)
        dt, tm = best
        olen = d2h_u32(d_olen, NT)
        st = d2h_u32(d_st, NT)
        if a.no_check:
            print("L%d: lz77 %.2f ms  encode %.2f ms  (NO CHECK: ablation build)" % (lvl, tm[1], tm[2]))
            continue
        assert not any(st), ("deflate status", [s for s in st if s][:4])
        # device round trip
        hip.hipMemset(d_back, 0, NT * B)
        timing(L, ctx)
        rc = L.zmi_inflate_batch_dev(ctx, d_out, d_coff, d_olen, NT, 1, d_back, d_off, d_cap, d_blen, d_bst, None)
        if rc != 0:
            sys.exit("inflate: " + L.zmi_last_error().decode())
        hip.hipDeviceSynchronize()
        itm = timing(L, ctx)
        bst = d2h_u32(d_bst, NT)
        blen = d2h_u32(d_blen, NT)
        L.zmi_checksum_batch_dev(ctx, d_back, d_off, d_blen, NT, 1, d_a1, d_c, None)
        hip.hipDeviceSynchronize()
        got = d2h_u32(d_a1, NT)
        ok = not any(bst) and blen == [B] * NT and got == want
        for i in list(range(0, S, max(1, S // max(1, a.host_verify))))[:a.host_verify] + list(range(S, NT)):
            comp = d2h(d_out + i * stride, olen[i])
            ok = ok and zlib.adler32(zlib.decompress(comp)) == want[i]
        ratio = S * B / float(sum(olen[:S])) if S else 0.0
        line = "L%d: lz77 %.2f ms  encode %.2f ms  wall %.2f ms (%.1f GiB/s)  ratio %.4f  roundtrip %s  [inflate decode %.2f resolve %.2f ms]" % (
            lvl, tm[1], tm[2], dt * 1e3, NT * B / 2**30 / dt, ratio, "ok" if ok else "FAILED", itm[3], itm[6])
        r = {"lz77_ms": tm[1], "encode_ms": tm[2], "ratio": ratio, "ok": ok, "decode_ms": itm[3], "resolve_ms": itm[6]}
        if a.classes and S >= 8:
            r["class_ratio"] = [B * len(olen[c:S:8]) / float(sum(olen[c:S:8])) for c in range(8)]
            line += "\n    class ratios: " + " ".join("%.3f" % x for x in r["class_ratio"])
        for i, (name, _) in enumerate(fx):
            r[name] = B / float(olen[S + i])
            line += "\n    %-16s %.4f" % (name, r[name])
        print(line)
        sys.stdout.flush()
        res["levels"][str(lvl)] = r
        if not ok:
            print("ROUND TRIP FAILED at level", lvl)
    if a.class_times and S >= 64:
        lvl = int(a.levels.split(",")[0])
        d_off2 = dmalloc(S * 8); d_len2 = dmalloc(S * 4)
        for c in range(8):
            idx = list(range(c, S, 8))
            h2d(d_off2, (C.c_uint64 * len(idx))(*[i * B for i in idx]))
            h2d(d_len2, (C.c_uint32 * len(idx))(*([B] * len(idx))))
            best = None
            for rep in range(2):
                timing(L, ctx)
                L.zmi_deflate_batch_dev(ctx, d_in, d_off2, d_len2, len(idx), B, lvl, int(os.environ.get('PROBE_STRATEGY', '0')), 1, d_out, stride, d_olen, d_st, None)
                hip.hipDeviceSynchronize()
                tm = timing(L, ctx)
                if best is None or tm[1] + tm[2] < best[1] + best[2]:
                    best = tm
            ol = d2h_u32(d_olen, len(idx))
            ibest = None
            for rep in range(0 if a.no_check else 2):
                timing(L, ctx)
                L.zmi_inflate_batch_dev(ctx, d_out, d_coff, d_olen, len(idx), 1, d_back, d_off, d_cap, d_blen, d_bst, None)
                hip.hipDeviceSynchronize()
                itm = timing(L, ctx)
                if ibest is None or itm[3] + itm[6] < ibest[3] + ibest[6]:
                    ibest = itm
            if ibest is None:
                ibest = [0.0] * 8
            print("    class %d L%d (%d shards): lz77 %.2f ms  encode %.2f ms  ratio %.3f   inflate decode %.2f resolve %.2f ms" % (
                c, lvl, len(idx), best[1], best[2], B * len(idx) / float(sum(ol)), ibest[3], ibest[6]))
            res.setdefault("class_times", {})[str(c)] = {"lz77_ms": best[1], "encode_ms": best[2], "decode_ms": ibest[3], "resolve_ms": ibest[6]}
    print("JSON " + json.dumps(res))


if __name__ == "__main__":
    main()
