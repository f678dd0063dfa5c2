"""GPU probe: per-kernel timings across levels / batch sizes (PROBE_S, PROBE_LEVELS, PROBE_CLASSES).
Run on the MI355X box through gpurun; writes human-readable lines to stdout."""
import ctypes as C
import hashlib
import json
import os
import sys

os.environ.setdefault("ZMI_TUNING", "1")   # the ZMI_* overrides are honoured only with this set
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
if os.environ.get("PROBE_LIB"):   # a measurement build of the engine (build_prof/): same ABI, extra counters
    import zlib_rs_amd._lib as _zl
    _zl.LIB = os.path.abspath(os.environ["PROBE_LIB"])
from zlib_rs_amd.engine import Engine, uniform_layout  # noqa: E402


def timing(e):
    sums = (C.c_double * 8)()
    cnts = (C.c_uint32 * 8)()
    e.L.zmi_ctx_get_timing(e._ctx, sums, cnts)
    return list(sums), list(cnts)


def main():
    sg = os.environ.get("PROBE_SCRATCH_GIB")   # the engine's scratch = one launch group (bench.py: 70 GiB = 16 865 shards)
    e = Engine(0, scratch_bytes=int(float(sg) * 2**30) if sg else None)
    props = torch.cuda.get_device_properties(0)
    print("device", props.name, "CUs", props.multi_processor_count, "mem GiB", props.total_memory / 2**30)
    B = 1 << 20
    e.L.zmi_ctx_set_timing(e._ctx, 1)
    for S in [int(x) for x in os.environ.get("PROBE_S", "256,2048").split(",")]:
        data = e.gen_shards(S, B)
        off, ln = uniform_layout(S, B, e.device)
        for lvl in [int(x) for x in os.environ.get("PROBE_LEVELS", "1,3,6,9").split(",")]:
            out, olen, st = e.deflate_batch(data, off, ln, B, level=lvl)
            torch.cuda.synchronize()
            timing(e)
            t = time.perf_counter()
            out, olen, st = e.deflate_batch(data, off, ln, B, level=lvl, out=out, out_len=olen, status=st)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            sums, cnts = timing(e)
            ratio = S * B / max(1.0, float(olen.to(torch.int64).sum().item()))
            print("deflate S=%d L%d: %.2f GiB/s wall  ratio %.3f  ms: checksum %.2f lz77 %.2f parse %.2f encode %.2f" %
                  (S, lvl, S * B / 2**30 / dt, ratio, sums[0], sums[1], sums[5], sums[2]))
            if hasattr(e.L, "zmi_enc_prof_read"):
                pr = (C.c_ulonglong * 8)()
                e.L.zmi_enc_prof_read(pr)
                tot = float(pr[0]) or 1.0
                print("    encode wave-cycles (both passes): kernel %.3g | recurrence %.1f %% chain %.1f %% walk %.1f %% prices %.1f %% emission %.1f %%" %
                      (tot, 100 * pr[1] / tot, 100 * pr[2] / tot, 100 * pr[3] / tot, 100 * pr[4] / tot, 100 * pr[5] / tot))
            if lvl == 6 and os.environ.get("PROBE_CLASSES"):
                for cls in range(8):
                    idx = torch.arange(cls, S, 8, device=e.device)
                    o2, l2, s2 = e.deflate_batch(data, off[idx].contiguous(), ln[idx].contiguous(), B, level=lvl)
                    torch.cuda.synchronize(); timing(e)
                    o2, l2, s2 = e.deflate_batch(data, off[idx].contiguous(), ln[idx].contiguous(), B, level=lvl)
                    torch.cuda.synchronize()
                    sm, _ = timing(e)
                    print("    class %d: lz77 %.2f ms parse %.2f ms encode %.2f ms for %d shards" % (cls, sm[1], sm[5], sm[2], idx.numel()))
            if lvl == 6:
                # per-class ratio
                l = olen.cpu().numpy().astype("int64")
                print("  per-class ratio L6:", ["%.2f" % (B * len(l[c::8]) / l[c::8].sum()) for c in range(8)])
                # inflate of our own output
                back = torch.empty(S * B, dtype=torch.uint8, device=e.device)
                cap = torch.full((S,), B, dtype=torch.int32, device=e.device)
                ooff = torch.arange(S, dtype=torch.int64, device=e.device) * B
                coff = torch.arange(S, dtype=torch.int64, device=e.device) * out.stride(0)
                e.inflate_batch(out, coff, olen, back, ooff, cap)
                torch.cuda.synchronize(); timing(e)
                t = time.perf_counter()
                blen, bst = e.inflate_batch(out, coff, olen, back, ooff, cap)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t
                sums, cnts = timing(e)
                print("inflate S=%d: %.2f GiB/s wall  ok=%s  ms: decode %.2f resolve %.2f checksum %.2f" %
                      (S, S * B / 2**30 / dt, bool(torch.equal(back, data)) and int((bst != 0).sum()) == 0, sums[3], sums[6], sums[0]))
                if os.environ.get("PROBE_CLASSES"):
                    for cls in range(8):
                        idx = torch.arange(cls, S, 8, device=e.device)
                        a = (coff[idx].contiguous(), olen[idx].contiguous(), back, ooff[idx].contiguous(), cap[idx].contiguous())
                        e.inflate_batch(out, *a)
                        torch.cuda.synchronize(); timing(e)
                        e.inflate_batch(out, *a)
                        torch.cuda.synchronize()
                        sm, _ = timing(e)
                        print("    class %d: decode %.2f resolve %.2f ms for %d streams" % (cls, sm[3], sm[6], idx.numel()))
        del data
    e.close()


if __name__ == "__main__":
    main()
