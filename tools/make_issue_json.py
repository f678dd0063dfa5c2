#!/usr/bin/env python3
"""profiles/rNN_rocprofv3_summary.csv (the counter section tools/prof_final.sh prints) -> profiles/rNN_issue.json: per kernel the
SQ counters of ONE 16 384-shard launch, in the form bench.py replays as roofline.issue (wave-instructions per byte, SIMD cycles
per instruction, VALU lane utilisation).  usage: make_issue_json.py SUMMARY.csv OUT.json TAG"""
import datetime
import json
import sys

WANT = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_THREAD_CYCLES_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAVES")


def main(src, dst, tag):
    rows, on = {}, False
    for line in open(src):
        line = line.strip()
        if line.startswith("kernel,counter,"):
            on = True
            continue
        if not on or line.startswith("#") or not line:
            continue
        k, c, v, n = line.split(",")
        if c in WANT:
            rows.setdefault(k, {})[c] = float(v)
            rows[k]["dispatches"] = int(n)
    out = {"_collected": "%s, %s" % (tag, datetime.date.today().isoformat()),
           "_command": "rocprofv3 --pmc SQ_* -- python bench.py --shards 16384 --steps 1 --warmup 0 --no-cpu --verify 0 --no-extras --no-stitch "
                       "(tools/prof_final.sh): deflate kernels = one launch of 16 384 x 1 MiB shards, inflate kernels = the round trip of the "
                       "same 16 384 streams (+ a 64-stream warm-up launch, 0.4 %)",
           "bytes_per_launch": 16384 << 20, "kernels": {}}
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out["_csrc_sha16"] = bench.csrc_sha16()   # (run right after the counter passes, on the same tree)
    for k, r in sorted(rows.items()):
        if "SQ_INSTS_VALU" not in r:
            continue
        out["kernels"][k] = {"valu": r.get("SQ_INSTS_VALU", 0), "salu": r.get("SQ_INSTS_SALU", 0), "lds": r.get("SQ_INSTS_LDS", 0),
                             "thread_cycles_valu": r.get("SQ_THREAD_CYCLES_VALU", 0), "busy_cycles": r.get("SQ_BUSY_CYCLES", 0),
                             "wave_cycles": r.get("SQ_WAVE_CYCLES", 0), "wait_any": r.get("SQ_WAIT_ANY", 0), "waves": r.get("SQ_WAVES", 0)}
    json.dump(out, open(dst, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "r??")
