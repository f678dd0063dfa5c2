"""Pure-Python DEFLATE token dumper (analysis tool, not on any product path)."""
import sys

LBASE = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEXT = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DBASE = [1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DEXT = [0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]

class Bits:
    def __init__(s, data, pos=0): s.d=data; s.p=pos*8
    def get(s, n):
        v=0
        for i in range(n):
            v |= ((s.d[s.p>>3]>>(s.p&7))&1)<<i; s.p+=1
        return v

def mktable(lens):
    cnt=[0]*16
    for l in lens: cnt[l]+=1
    cnt[0]=0; code=0; nxt=[0]*16
    for b in range(1,16):
        code=(code+cnt[b-1])<<1; nxt[b]=code
    t={}
    for s,l in enumerate(lens):
        if l: t[(l,nxt[l])]=s; nxt[l]+=1
    return t

def decode(b,t):
    code=0
    for l in range(1,16):
        code=(code<<1)|b.get(1)
        if (l,code) in t: return t[(l,code)]
    raise ValueError('bad code')

def dump(raw, start=0, verbose=False):
    b=Bits(raw,start); blocks=[]; out=bytearray()
    while True:
        fin=b.get(1); typ=b.get(2); p0=b.p
        info={'final':fin,'type':typ,'lits':0,'matches':0,'mlen':0,'hdr_bits':0,'maxdist':0}
        if typ==0:
            b.p=(b.p+7)&~7; l=b.get(16); b.get(16)
            out+=raw[b.p>>3:(b.p>>3)+l]; b.p+=8*l; info['stored']=l
        else:
            if typ==1:
                lt=mktable([8]*144+[9]*112+[7]*24+[8]*8); dt=mktable([5]*30)
            else:
                hl=b.get(5)+257; hd=b.get(5)+1; hc=b.get(4)+4
                order=[16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15]
                cl=[0]*19
                for i in range(hc): cl[order[i]]=b.get(3)
                ct=mktable(cl); lens=[]
                while len(lens)<hl+hd:
                    s=decode(b,ct)
                    if s<16: lens.append(s)
                    elif s==16: lens+= [lens[-1]]*(3+b.get(2))
                    elif s==17: lens+=[0]*(3+b.get(3))
                    else: lens+=[0]*(11+b.get(7))
                lt=mktable(lens[:hl]); dt=mktable(lens[hl:])
                info['hdr_bits']=b.p-p0
            while True:
                s=decode(b,lt)
                if s<256: out.append(s); info['lits']+=1
                elif s==256: break
                else:
                    l=LBASE[s-257]+b.get(LEXT[s-257]); d=decode(b,dt); dist=DBASE[d]+b.get(DEXT[d])
                    for _ in range(l): out.append(out[-dist])
                    info['matches']+=1; info['mlen']+=l; info['maxdist']=max(info['maxdist'],dist)
                    if verbose: print('match',l,dist)
        info['bits']=b.p-p0
        blocks.append(info)
        if fin: break
    return blocks, bytes(out)

if __name__=='__main__':
    raw=open(sys.argv[1],'rb').read()
    bl,out=dump(raw,int(sys.argv[2]) if len(sys.argv)>2 else 2)
    for x in bl: print(x)
