"""Randomised kernel round trips on the CPU SIMT emulator (tests/emu): random shard mixes, levels, strategies and
wrappers through deflate, system zlib as the reader; inflate of our own and of the system's streams.
usage: python tools/emu_stress.py SEED SECONDS   (test infrastructure; the product path needs an MI355X)"""
import os
import sys, zlib, random, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import zmi_ctypes, oracle_lib
o=oracle_lib.load()
eng=zmi_ctypes.Engine(zmi_ctypes.load_emu(False))
rnd=random.Random(int(sys.argv[1]))
t=time.time(); rounds=0
wb={0:-15,1:15,2:31}
while time.time()-t < float(sys.argv[2]):
    n=rnd.randrange(1,6)
    shards=[]
    for i in range(n):
        ln=rnd.choice([0,1,2,3,63,64,65,1000,4095,4096,4097,rnd.randrange(70000),rnd.randrange(300000)])
        kind=rnd.random()
        if kind<0.6: d=o.gen_shard(rnd.randrange(8),ln)
        elif kind<0.75: d=bytes(rnd.randrange(256) for _ in range(min(ln,20000)))
        elif kind<0.9: d=(bytes([rnd.randrange(256)])*rnd.randrange(1,300)+o.gen_shard(3,50))*(ln//200+1); d=d[:ln]
        else: d=bytes(ln)
        shards.append(d)
    lvl=rnd.randrange(0,10); strat=rnd.choice([0,0,0,1,2,3,4]); wrap=rnd.randrange(3)
    outs,st=eng.deflate(shards,level=lvl,strategy=strat,wrap=wrap)
    assert st==[0]*n,(st,lvl,strat)
    for c,d in zip(outs,shards):
        assert zlib.decompressobj(wb[wrap]).decompress(c)==d,(lvl,strat,wrap,len(d))
    back,st2=eng.inflate(outs,[len(d)+rnd.choice([0,1,100]) for d in shards],wrap)
    assert st2==[0]*n and back==shards,(st2,lvl,strat,wrap)
    # streams of the system's zlib
    comps=[]
    for d in shards:
        co=zlib.compressobj(rnd.choice([0,1,4,6,9]),zlib.DEFLATED,wb[wrap],rnd.choice([1,8,9]),rnd.choice([0,1,2,3,4]))
        comps.append(co.compress(d)+co.flush())
    back,st3=eng.inflate(comps,[len(d) for d in shards],wrap)
    assert st3==[0]*n and back==shards,(st3,wrap)
    rounds+=1
print("emu stress ok:",rounds,"rounds, seed",sys.argv[1])
