"""bench.py's N > 1 control flow with two ranks, before it meets eight GPUs: the functions the multi-GPU line
is made of -- stitch_leg_multi (communicator, size tables, plan, slab packing, exchange rounds, the overlapped step),
with_deadline and attach_stitch (value_with_stitch) -- run here as two PROCESSES on torch.distributed's gloo backend, with the
engine object bound to the CPU build of the kernels (tests/emu: device pointers = host pointers) and RCCL replaced by
tests/emu/libmock_rccl.so (files as the wire).  The stitched output is read back by gzip: the shards of both ranks in global
order (the reference's parallel-deflate recipe, zlib-rs/src/deflate.rs:4149-4221).  What this cannot cover is the hardware: the
scaling curve itself stays unmeasured until a multi-GPU node runs it."""
import gzip
import json
import os
import socket
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import contextlib, json, os, sys
ROOT = %(root)r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["ZMI_RCCL_LIB"] = os.path.join(ROOT, "tests", "emu", "libmock_rccl.so")
os.environ["ZMI_MOCK_RCCL_DIR"] = %(wire)r
import torch
import torch.distributed as dist
import zlib_rs_amd._lib as zl
zl.LIB = os.path.join(ROOT, "tests", "emu", "libzmi355_emu.so")      # the same C ABI, compiled for the CPU (test infrastructure)
import zlib_rs_amd.engine as eng_mod
eng_mod._stream_ptr = lambda: None
import bench

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)


class EmuEngine(eng_mod.Engine):
    def __init__(self):
        import ctypes as C
        self.device = torch.device("cpu")
        self.L = zl.lib()
        self._ctx = C.c_void_p()
        zl.check(self.L.zmi_ctx_create(C.byref(self._ctx), 0), "zmi_ctx_create")


class _Stream:
    def synchronize(self):
        pass


class _Cuda:
    @staticmethod
    def set_device(d): pass
    @staticmethod
    def synchronize(): pass
    @staticmethod
    def Stream(device=None): return _Stream()
    @staticmethod
    def stream(s): return contextlib.nullcontext()


class TorchShim:
    cuda = _Cuda()
    def __getattr__(self, name): return getattr(torch, name)


e = EmuEngine()
S, B = 6, 8192
dev = torch.device("cpu")
data = e.gen_shards(S, B, first_shard=rank, shard_step=world)        # round-robin ownership: local j = global j * world + rank
off, ln = eng_mod.uniform_layout(S, B, dev)
stride = e.deflate_bound(B, eng_mod.WRAP_GZIP)
out = torch.empty((S, stride), dtype=torch.uint8)
olen = torch.empty(S, dtype=torch.int32)
st = torch.empty(S, dtype=torch.int32)
def step():
    e.deflate_batch(data, off, ln, B, level=6, wrap=eng_mod.WRAP_GZIP, out=out, out_len=olen, status=st)
step()
assert int((st != 0).sum()) == 0
tot = torch.tensor([int(olen.sum())], dtype=torch.int64)
dist.all_reduce(tot)
stitched = torch.zeros(int(tot.item()) + 16, dtype=torch.uint8)
stitch_obj = bench.with_deadline(lambda: bench.stitch_leg_multi(e, dist, TorchShim(), out, olen, dev, world, rank, chunk_bytes=1 << 20, step=step,
                                                                step_bytes=S * B, step_s=0.01, scatter_out=stitched), 240.0)
line = bench.attach_stitch({"metric": "test", "n_gpus": world}, stitch_obj)
if rank == 0:
    open(%(outfile)r, "wb").write(bytes(stitched[:int(tot.item())].numpy()))
    print(json.dumps(line), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_bench_stitch_flow_on_the_cpu_build():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")], check=True)
    with tempfile.TemporaryDirectory(prefix="zmi_bench_emu_") as td:
        wire = os.path.join(td, "wire")
        os.mkdir(wire)
        outfile = os.path.join(td, "stitched.gz")
        code = WORKER % {"root": ROOT, "wire": wire, "outfile": outfile}
        port = _free_port()
        procs = []
        for rank in range(2):
            env = dict(os.environ)
            env.update({"RANK": str(rank), "WORLD_SIZE": "2", "LOCAL_RANK": str(rank), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
            procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = [p.communicate(timeout=600) for p in procs]
        for p, (so, se) in zip(procs, outs):
            assert p.returncode == 0, se[-3000:]
        lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
        assert len(lines) == 1, outs[0][0]
        line = json.loads(lines[0])
        assert line["n_gpus"] == 2
        st = line["stitch"]
        assert not st.get("failed"), st
        assert st["overlap"] and not st["overlap"].get("failed"), st["overlap"]
        assert line["value_with_stitch"] == st["overlap"]["value_with_stitch"] > 0
        assert st["received_GB_per_rank"] > 0 and "checked against the owner's byte sum (1 peers)" in st["exchange"]
        # the stitched output: every shard of both ranks, in global order, as one multi-member gzip file
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        o = oracle_lib.load(rebuild=False)
        want = b"".join(o.gen_shard(g, 8192) for g in range(12))
        assert st["stitched_bytes"] == os.path.getsize(outfile)
        assert gzip.open(outfile, "rb").read() == want
