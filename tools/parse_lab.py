"""ANALYSIS TOOL (not on any product path): what can a cost-based parse buy over the three-deep lazy rule, given exactly
the per-position matches lz77.hip reports?  Runs the match search on the CPU emulator build, then compares parse rules by
the zeroth-order cost of their token streams (tools/parse_lab.c).  usage: python tools/parse_lab.py [chain] [fixture]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_checks  # noqa: E402
import zmi_ctypes  # noqa: E402


class LzParams(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in ("max_chain", "nice_len", "good_len", "max_dist", "claim", "hash6", "producers", "dbg",
                                           "carry", "dict_len", "barren_chain", "far4", "far5")]


def matches(L, data, chain=4, nice=128, good=16):
    n = len(data)
    buf = np.frombuffer(data + b"\0" * 64, dtype=np.uint8).copy()
    off = np.zeros(1, dtype=np.uint64)
    ln = np.array([n], dtype=np.uint32)
    stride = (n + 63) & ~63
    m = np.zeros(stride + 64, dtype=np.uint32)
    prm = LzParams(chain, nice, good, 32768, 64, 1, 2 if chain <= 8 else 1, 0, 0, 0, 1, 32768, 32768)
    L.zmi_launch_lz77.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, LzParams, C.c_void_p]
    rc = L.zmi_launch_lz77(buf.ctypes.data, off.ctypes.data, ln.ctypes.data, 0, 1, m.ctypes.data, stride, prm, None)
    assert rc == 0
    return m[:n].copy()


def main():
    chain = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    which = sys.argv[2] if len(sys.argv) > 2 else "lcet10.txt"
    L = zmi_ctypes.load_emu(rebuild=False)
    so = os.path.join(ROOT, "gpurun_out", "parse_lab.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "parse_lab.c"), "-lm"], check=True)
    P = C.CDLL(so)
    P.cost_tokens.restype = C.c_double
    P.cost_tokens.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
    V, U, I = C.c_void_p, C.c_uint32, C.c_int
    P.parse_lazy3.argtypes = [V, U, U, U, U, V, V]
    P.prices_from.argtypes = [V, I, I, V, V]
    P.parse_dp.argtypes = [V, U, U, V, V, I, U, V, V, V, V, I]
    if which.startswith("shard"):
        import oracle_lib
        data = oracle_lib.load(rebuild=False).gen_shard(int(which[5:]), 1 << 20)
    else:
        raw = dict(parity_checks.real_fixtures())[which]
        data = parity_checks.tile(raw)
    n = len(data)
    m = matches(L, data, chain)
    tok = np.zeros(n + 8, dtype=np.uint32)
    tpos = np.zeros(n + 8, dtype=np.uint32)
    cost = np.zeros(n + 8, dtype=np.uint32)
    step = np.zeros(n + 8, dtype=np.uint16)
    BLK = 16384
    nt = P.parse_lazy3(m.ctypes.data, n, 32, 1, 2, tok.ctypes.data, tpos.ctypes.data)
    base = P.cost_tokens(tok.ctypes.data, nt, BLK, 640.0)
    print("%s chain %d: lazy3  tokens %7d  bytes %8.0f  ratio %.4f" % (which, chain, nt, base / 8, n / (base / 8)))
    P.parse_lazy3_trunc.argtypes = [V, U, U, U, U, I, I, V, V]
    if os.environ.get("LAB_TRUNC"):
        for margin in (0, 1, 2, 3):
            for depth in (1, 2, 3):
                tk = np.zeros(n + 8, dtype=np.uint32); tp = np.zeros(n + 8, dtype=np.uint32)
                ntt = P.parse_lazy3_trunc(m.ctypes.data, n, 32, 1, 2, margin, depth, tk.ctypes.data, tp.ctypes.data)
                c = P.cost_tokens(tk.ctypes.data, ntt, BLK, 640.0)
                print("  lazy3 + truncate margin %d depth %d: tokens %7d ratio %.4f (%+.2f %%)" % (margin, depth, ntt, n / (c / 8), 100.0 * (base / c - 1.0)))
        return
    P.parse_strips.argtypes = [V, U, U, U, I, I, I, V, V, V, I]
    dec = np.zeros(n + 8, dtype=np.uint8)
    P.set_ext.argtypes = [V, V, I, U]
    P.set_price_model.argtypes = [C.c_double, I]
    if os.environ.get("LAB_EXT"):
        lt = tok[:nt].copy(); lpz = tpos[:nt].copy()
        P.set_ext(lt.ctypes.data, lpz.ctypes.data, nt, int(os.environ["LAB_EXT"]))
    P.set_stat0.argtypes = [I, I]
    if os.environ.get("LAB_STAT0"): P.set_stat0(int(os.environ["LAB_STAT0"]), int(os.environ.get("LAB_MINL", "3")))
    P.set_sample.argtypes = [I, I]
    if os.environ.get("LAB_HALF"): P.set_sample(int(os.environ["LAB_HALF"]), int(os.environ.get("LAB_EVERY", "1")))
    P.set_lmax.argtypes = [I]
    if os.environ.get("LAB_LMAX"): P.set_lmax(int(os.environ["LAB_LMAX"]))
    P.set_force.argtypes = [I]
    if os.environ.get("LAB_FORCE"): P.set_force(int(os.environ["LAB_FORCE"]))
    P.set_flat.argtypes = [I]
    if os.environ.get("LAB_FLAT"): P.set_flat(int(os.environ["LAB_FLAT"]))
    P.set_pen.argtypes = [I, I]
    if os.environ.get("LAB_PEN"):
        P.set_pen(int(os.environ["LAB_PEN"]), int(os.environ.get("LAB_PEN_SHORT", "0")))
    if os.environ.get("LAB_SMOOTH"):
        P.set_price_model(float(os.environ["LAB_SMOOTH"]), int(os.environ.get("LAB_PMAX", "15")))
    for S in (64,):
        for ncand in (1, 2, 4):
            for first in (0, 1, 2, 3, 4, 5, 6):
                for decay in (0, 1):
                    if (first not in (1, 3, 4, 5, 6) or decay) and not (S == 64 and ncand == 4): continue
                    nt2 = 0
                    tok2 = np.zeros(n + 8, dtype=np.uint32); tpos2 = np.zeros(n + 8, dtype=np.uint32)
                    PIECE = int(os.environ.get('LAB_PIECE', '65536'))
                    for p0 in range(0, n, PIECE):
                        nt2 = P.parse_strips(m.ctypes.data, p0, min(n, p0 + PIECE), S, ncand, first, decay, dec.ctypes.data, tok2.ctypes.data, tpos2.ctypes.data, nt2)
                    c = P.cost_tokens(tok2.ctypes.data, nt2, BLK, 640.0)
                    print("  strips S=%4d ncand %d first %d decay %d: tokens %7d bytes %8.0f ratio %.4f (%+.2f %%)" % (S, ncand, first, decay, nt2, c / 8, n / (c / 8), 100.0 * (base / c - 1.0)))
    if os.environ.get("LAB_STRIPS_ONLY"): return
    lazy_tok = tok[:nt].copy()
    lazy_pos = tpos[:nt].copy()
    for mode, minlen, name in ((0, 3, "dp full-length only"), (2, 3, "dp len..len-3"), (1, 3, "dp every length >= 3"), (1, 4, "dp every length >= 4")):
        for src in ("prev-piece lazy stats", "same-piece lazy stats", "iterated x2"):
            # pieces of 64 KiB as the encoder has them; prices from the lazy parse of the previous / the same piece
            lp = np.zeros(288, dtype=np.uint32)
            dp = np.zeros(32, dtype=np.uint32)
            nt2 = 0
            tok2 = np.zeros(n + 8, dtype=np.uint32)
            tpos2 = np.zeros(n + 8, dtype=np.uint32)
            for p0 in range(0, n, 65536):
                p1 = min(n, p0 + 65536)
                if src == "prev-piece lazy stats":
                    q0, q1 = max(0, p0 - 65536), p0 if p0 else p1   # the first piece prices itself
                else:
                    q0, q1 = p0, p1
                t0, t1 = int(np.searchsorted(lazy_pos, q0)), int(np.searchsorted(lazy_pos, q1))
                P.prices_from(lazy_tok.ctypes.data, t0, t1, lp.ctypes.data, dp.ctypes.data)
                start = nt2
                nt2 = P.parse_dp(m.ctypes.data, p0, p1, lp.ctypes.data, dp.ctypes.data, mode, minlen, cost.ctypes.data, step.ctypes.data,
                                 tok2.ctypes.data, tpos2.ctypes.data, nt2)
                if src == "iterated x2":
                    P.prices_from(tok2.ctypes.data, start, nt2, lp.ctypes.data, dp.ctypes.data)
                    nt2 = P.parse_dp(m.ctypes.data, p0, p1, lp.ctypes.data, dp.ctypes.data, mode, minlen, cost.ctypes.data, step.ctypes.data,
                                     tok2.ctypes.data, tpos2.ctypes.data, start)
            c = P.cost_tokens(tok2.ctypes.data, nt2, BLK, 640.0)
            print("  %-24s %-22s tokens %7d  bytes %8.0f  ratio %.4f  (%+.2f %%)" % (name, src, nt2, c / 8, n / (c / 8), 100.0 * (base / c - 1.0)))


if __name__ == "__main__":
    main()
