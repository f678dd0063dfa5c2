"""CPU test of the C-ABI multi-GPU stitch (include/zmi355.h: zmi_comm_*, zmi_exchange_sizes, zmi_stitch_plan_dev,
zmi_exchange_slabs, zmi_exchange_slabs_round; csrc/exchange.hip) with world_size 2 and 3: separate PROCESSES run the real
host code of the emulator build (device pointers = host pointers) against tests/emu/libmock_rccl.so, a stand-in for RCCL
that carries the bytes through files.  What is checked is what a non-Python host relies on: the table layout, the plan's
offsets, the grouped send / receive pattern in whole-slab and bounded-round form, all-gather and gather-to-root, and that
the scattered result is the multi-member gzip file of all shards in global order (the contract of the reference's
parallel-deflate recipe, zlib-rs/src/deflate.rs:4145-4221)."""
import ctypes as C
import gzip
import multiprocessing as mp
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHARD = 1 << 12


def _p(a):
    return a.ctypes.data


def _worker(rank, world, n_total, wire, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["ZMI_RCCL_LIB"] = os.path.join(ROOT, "tests", "emu", "libmock_rccl.so")
        os.environ["ZMI_MOCK_RCCL_DIR"] = wire
        import oracle_lib
        import zmi_ctypes
        L = zmi_ctypes.load_emu(rebuild=False)
        o = oracle_lib.load(rebuild=False)
        eng = zmi_ctypes.Engine(L)
        ok = lambda rc, what: (_ for _ in ()).throw(RuntimeError("%s: %d %s" % (what, rc, L.zmi_last_error().decode()))) if rc else None
        # the communicator: rank 0 makes the id, a file carries it (a real host: MPI_Bcast, torch's store ...)
        idf = os.path.join(wire, "uid")
        uid = C.create_string_buffer(128)
        if rank == 0:
            ok(L.zmi_comm_unique_id(uid), "zmi_comm_unique_id")
            open(idf + ".tmp", "wb").write(uid.raw)
            os.rename(idf + ".tmp", idf)
        else:
            import time
            for _ in range(3000):
                if os.path.exists(idf):
                    break
                time.sleep(0.01)
            uid = C.create_string_buffer(open(idf, "rb").read(), 128)
        comm = C.c_void_p()
        ok(L.zmi_comm_create(C.byref(comm), eng.ctx, world, rank, uid), "zmi_comm_create")
        assert L.zmi_comm_world(comm) == world and L.zmi_comm_rank(comm) == rank

        # this rank's shards (round-robin), compressed by the CPU oracle as gzip members, in compress_bound-strided slots
        mine = list(range(rank, n_total, world))
        n_local = len(mine)
        members = [o.deflate(o.gen_shard(g, SHARD), 6, 2)[1] for g in mine]
        stride = int(L.zmi_deflate_bound(SHARD, 2))
        slots = np.full(n_local * stride + 64, 0xA5, dtype=np.uint8)
        for j, m in enumerate(members):
            slots[j * stride:j * stride + len(m)] = np.frombuffer(m, dtype=np.uint8)
        sizes = np.array([len(m) for m in members], dtype=np.uint32)
        slab = np.full(int(sizes.sum()) + 64, 0xEE, dtype=np.uint8)
        soff_own = np.zeros(n_local + 1, dtype=np.uint64)
        ok(L.zmi_pack_slab_dev(eng.ctx, _p(slots), stride, _p(sizes), n_local, _p(slab), int(sizes.sum()), _p(soff_own), None), "pack")

        # 1. size tables
        table = np.zeros((world, n_local), dtype=np.uint32)
        ok(L.zmi_exchange_sizes(comm, _p(sizes), n_local, _p(table), None), "zmi_exchange_sizes")
        assert (table[rank] == sizes).all()
        # 2. plan
        goff = np.zeros((world, n_local), dtype=np.uint64)
        soff = np.zeros((world, n_local + 1), dtype=np.uint64)
        d_tot = np.zeros(world + 1, dtype=np.uint64)
        totals = np.zeros(world + 1, dtype=np.uint64)
        ok(L.zmi_stitch_plan_dev(eng.ctx, _p(table), world, n_local, _p(goff), _p(soff), _p(d_tot), _p(totals), None), "plan")
        flat = table.T.reshape(-1).astype(np.uint64)           # global order g = j * world + r
        want_goff = (np.cumsum(flat) - flat).reshape(n_local, world).T
        assert (goff == want_goff).all()
        assert (soff[:, :-1] == np.cumsum(table.astype(np.uint64), 1) - table).all() and (soff[:, -1] == table.sum(1)).all()
        assert (totals[:world] == table.sum(1)).all() and int(totals[world]) == int(flat.sum())
        assert (soff[rank] == soff_own).all()
        total = int(totals[world])
        want = b"".join(o.gen_shard(g, SHARD) for g in range(n_total))
        max_len = int(table.max())

        def scatter(out, r, src, src_off_row):
            ok(L.zmi_copy_ranges_dev(eng.ctx, _p(src), _p(src_off_row), 0, _p(table[r]), n_local, max_len, _p(out), _p(goff[r]), total, None), "scatter")

        # 3a. whole slabs, all-gather and gather-to-root, several chunk sizes (rounds)
        for root, chunk in ((-1, 1 << 30), (-1, 1000), (0, 4096), (world - 1, 333)):
            recv = [np.full(int(totals[p]) + 16, 0xCC, dtype=np.uint8) for p in range(world)]
            ptrs = (C.c_void_p * world)(*[None if p == rank else _p(recv[p]) for p in range(world)])
            ok(L.zmi_exchange_slabs(comm, _p(slab), _p(totals), ptrs, chunk, root, None), "zmi_exchange_slabs")
            if root >= 0 and rank != root:
                assert all((recv[p] == 0xCC).all() for p in range(world))     # nothing arrives at a non-root
                continue
            out = np.zeros(total + 16, dtype=np.uint8)
            for r in range(world):
                src = slab if r == rank else recv[r]
                assert r == rank or (recv[r][int(totals[r]):] == 0xCC).all()   # nothing behind a slab's end
                scatter(out, r, src, soff[r][:-1].copy())
            assert gzip.decompress(out[:total].tobytes()) == want, (root, chunk)
        # 3b. bounded rounds: chunk bytes of staging per peer, consumed (scattered) round by round
        chunk = 777
        stage = [np.zeros(chunk + 16, dtype=np.uint8) for _ in range(world)]
        sptr = (C.c_void_p * world)(*[None if p == rank else _p(stage[p]) for p in range(world)])
        peer_bytes = {p: bytearray() for p in range(world) if p != rank}
        lo = 0
        while lo < int(totals[:world].max()):
            ok(L.zmi_exchange_slabs_round(comm, _p(slab), _p(totals), lo, chunk, sptr, -1, None), "zmi_exchange_slabs_round")
            for p in peer_bytes:
                n = max(0, min(chunk, int(totals[p]) - lo))
                peer_bytes[p] += stage[p][:n].tobytes()
            lo += chunk
        out = np.zeros(total + 16, dtype=np.uint8)
        scatter(out, rank, slab, soff[rank][:-1].copy())
        for p, b in peer_bytes.items():
            assert len(b) == int(totals[p])
            scatter(out, p, np.frombuffer(bytes(b) + b"\0" * 16, dtype=np.uint8).copy(), soff[p][:-1].copy())
        assert gzip.decompress(out[:total].tobytes()) == want
        # argument errors are reported, not crashed on
        assert L.zmi_exchange_slabs(comm, _p(slab), _p(totals), ptrs, 0, -1, None) == -103
        assert L.zmi_exchange_slabs(comm, _p(slab), _p(totals), ptrs, 4096, world, None) == -103
        # a receiving rank that leaves a peer's entry NULL is refused on the spot -- that peer would send into nothing and the
        # group would hang on RCCL (the mock blocks on an unmatched send too); every rank does it here, so nothing is posted
        nulls = (C.c_void_p * world)(*[None] * world)
        assert L.zmi_exchange_slabs(comm, _p(slab), _p(totals), nulls, 4096, -1, None) == -103
        assert L.zmi_exchange_slabs_round(comm, _p(slab), _p(totals), 0, 4096, nulls, -1, None) == -103
        assert L.zmi_exchange_slabs(comm, _p(slab), _p(totals), None, 4096, -1, None) == -103
        # ... but a rank that does not receive (gather-to-root) needs no room at all
        ok(L.zmi_exchange_slabs(comm, _p(slab), _p(totals), ptrs if rank == 0 else None, 4096, 0, None), "gather to root, NULL d_recv elsewhere")
        # the ranks must agree on n_local (short ranks pad with zero sizes): a mismatch is reported on EVERY rank, before the
        # table's all-gather could hang on it
        assert L.zmi_exchange_sizes(comm, _p(sizes), n_local - (1 if rank == 0 else 0), _p(table), None) == -103
        ok(L.zmi_comm_destroy(comm), "zmi_comm_destroy")
        eng.close()
        q.put((rank, "ok"))
    except Exception as ex:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL " + repr(ex) + "\n" + traceback.format_exc()))


def _run(world, n_total):
    import oracle_lib
    import zmi_ctypes
    oracle_lib.load()            # build once in the parent
    zmi_ctypes.load_emu()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory(prefix="zmi_wire_") as wire:
        procs = [ctx.Process(target=_worker, args=(r, world, n_total, wire, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=300) for _ in range(world)]
        for p in procs:
            p.join(60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def test_c_abi_exchange_two_ranks():
    _run(2, 12)


def test_c_abi_exchange_three_ranks():
    _run(3, 12)


def test_rccl_missing_is_reported(monkeypatch):
    """no RCCL -> ZMI_E_NORCCL with a message, never a crash (a single-GPU user of the library never needs it)"""
    import subprocess
    code = ("import os,sys,ctypes as C; sys.path.insert(0, %r); os.environ['ZMI_RCCL_LIB']='/nonexistent/librccl.so';"
            "import zmi_ctypes; L = zmi_ctypes.load_emu(rebuild=False); b = C.create_string_buffer(128);"
            "rc = L.zmi_comm_unique_id(b); print(rc, L.zmi_last_error().decode())") % os.path.join(ROOT, "tests")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("-105 RCCL is not available"), out.stdout + out.stderr
