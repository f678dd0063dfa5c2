"""Hand-built raw DEFLATE streams for the tests: one fixed-Huffman block from an explicit token list.

Token = int (literal byte) or (length, distance).  Lets a test place back-references at exact output
positions -- e.g. the resolve-pass regression of ADVICE r01 (hole at p with p % 1024 in 1021..1023,
distance 32766..32768, next hole 2048 further).  RFC 1951 section 3.2.5 / 3.2.6 tables, written out here.
"""

_LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
_LEXT = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
_DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
          8193, 12289, 16385, 24577]
_DEXT = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]


class _Bits:
    def __init__(self):
        self.acc = 0
        self.n = 0
        self.out = bytearray()

    def put(self, val, nbits):            # LSB first (extra bits, header fields)
        self.acc |= val << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 0xFF)
            self.acc >>= 8
            self.n -= 8

    def put_code(self, code, nbits):      # Huffman codes go MSB first
        rev = int(bin(code)[2:].zfill(nbits)[::-1], 2)
        self.put(rev, nbits)

    def finish(self):
        if self.n:
            self.out.append(self.acc & 0xFF)
        return bytes(self.out)


def _lit_code(s):
    if s < 144:
        return 0x30 + s, 8
    if s < 256:
        return 0x190 + (s - 144), 9
    if s < 280:
        return s - 256, 7
    return 0xC0 + (s - 280), 8


def fixed_block(tokens, final=True):
    """raw deflate stream: one fixed-Huffman block holding `tokens`"""
    b = _Bits()
    b.put(1 if final else 0, 1)
    b.put(1, 2)
    for t in tokens:
        if isinstance(t, int):
            b.put_code(*_lit_code(t))
            continue
        length, dist = t
        li = max(i for i in range(29) if _LBASE[i] <= length)
        if length == 258:
            li = 28
        b.put_code(*_lit_code(257 + li))
        b.put(length - _LBASE[li], _LEXT[li])
        di = max(i for i in range(30) if _DBASE[i] <= dist)
        b.put_code(di, 5)
        b.put(dist - _DBASE[di], _DEXT[di])
    b.put_code(*_lit_code(256))
    return b.finish()


def expand(tokens):
    """what the token list decodes to"""
    out = bytearray()
    for t in tokens:
        if isinstance(t, int):
            out.append(t)
        else:
            length, dist = t
            for _ in range(length):
                out.append(out[-dist])
    return bytes(out)
