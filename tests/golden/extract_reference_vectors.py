#!/usr/bin/env python3
"""Extract the byte-exact deflate golden vectors from the reference's own tests.

Run in the build container (needs /root/reference); writes tests/golden/deflate_vectors.json,
which is committed.  Nothing at test time reads /root/reference.

Sources (all `fuzz_based_test(input, config, expected)` / `compress_slice` + EXPECTED call sites):
  zlib-rs/src/deflate.rs               hello_world_huffman_only, hello_world_quick,
                                       hello_world_quick_random, simple_rle, fill_window_out_of_bounds,
                                       gzip_no_header, gzip_stored_block_checksum
  test-libz-rs-sys/src/deflate.rs      mod fuzz_based_tests (vectors with a non-empty `expected`)
  libz-rs-sys/src/lib.rs               compress doctest ("Ferris")
"""
import json
import os
import re
import sys

REF = "/root/reference"
OS_CODE = 3


def strip_comments(s):
    return re.sub(r"//[^\n]*", "", s)


def split_args(s):
    """split top-level comma separated arguments"""
    out, depth, cur, i, instr = [], 0, "", 0, False
    while i < len(s):
        c = s[i]
        if instr:
            cur += c
            if c == "\\":
                cur += s[i + 1]
                i += 1
            elif c == '"':
                instr = False
        elif c == '"':
            instr = True
            cur += c
        elif c in "([{":
            depth += 1
            cur += c
        elif c in ")]}":
            depth -= 1
            cur += c
        elif c == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += c
        i += 1
    if cur.strip():
        out.append(cur.strip())
    return out


def parse_str(lit):
    """Rust string literal body -> bytes (UTF-8)"""
    out, i = "", 0
    while i < len(lit):
        c = lit[i]
        if c == "\\":
            n = lit[i + 1]
            if n == "0":
                out += "\0"; i += 2
            elif n == "n":
                out += "\n"; i += 2
            elif n == "t":
                out += "\t"; i += 2
            elif n == "r":
                out += "\r"; i += 2
            elif n == "\\":
                out += "\\"; i += 2
            elif n == '"':
                out += '"'; i += 2
            elif n == "'":
                out += "'"; i += 2
            elif n == "x":
                out += chr(int(lit[i + 2:i + 4], 16)); i += 4
            elif n == "u":
                j = lit.index("}", i)
                out += chr(int(lit[i + 3:j], 16)); i = j + 1
            else:
                raise ValueError("escape " + n)
        else:
            out += c
            i += 1
    return out.encode("utf-8")


def parse_bytes(expr, consts):
    expr = expr.strip()
    expr = re.sub(r"\.as_bytes\(\)$", "", expr).strip()
    if expr.startswith("&"):
        expr = expr[1:].strip()
    if expr in consts:
        return parse_bytes(consts[expr], consts)
    if expr.startswith('b"') or expr.startswith('"'):
        body = expr[expr.index('"') + 1:expr.rindex('"')]
        if expr.startswith('b"'):
            return parse_str(body).decode("utf-8").encode("latin-1")
        return parse_str(body)
    if expr.startswith("["):
        items = split_args(expr[1:expr.rindex("]")])
        out = bytearray()
        for it in items:
            it = it.strip()
            if not it:
                continue
            if it in ("os", "gz_header::OS_CODE"):
                out.append(OS_CODE)
            else:
                out.append(int(it.replace("_", ""), 0))
        return bytes(out)
    raise ValueError("cannot parse bytes: " + expr[:60])


STRATS = {"Default": 0, "Filtered": 1, "HuffmanOnly": 2, "Rle": 3, "Fixed": 4}


def parse_config(expr):
    cfg = {"level": 6, "window_bits": 15, "mem_level": 8, "strategy": 0}
    expr = expr.strip()
    if expr in ("DeflateConfig::default()", "config"):
        return cfg if expr != "config" else None
    m = re.search(r"level:\s*(-?\d+)", expr)
    if m:
        cfg["level"] = int(m.group(1))
    m = re.search(r"window_bits:\s*([\w:]+)", expr)
    if m:
        v = m.group(1)
        cfg["window_bits"] = 15 if "MAX_WBITS" in v else int(v)
    m = re.search(r"mem_level:\s*([\w:]+)", expr)
    if m:
        v = m.group(1)
        cfg["mem_level"] = 8 if "DEF_MEM_LEVEL" in v else int(v)
    m = re.search(r"strategy:\s*Strategy::(\w+)", expr)
    if m:
        cfg["strategy"] = STRATS[m.group(1)]
    if cfg["level"] == -1:
        cfg["level"] = 6
    return cfg


def function_spans(src):
    """yield (name, body) for every `fn name(...) {` at any nesting (brace matched)"""
    for m in re.finditer(r"fn\s+(\w+)\s*\([^)]*\)[^{;]*\{", src):
        i = m.end()
        depth = 1
        instr = False
        while i < len(src) and depth:
            c = src[i]
            if instr:
                if c == "\\":
                    i += 1
                elif c == '"':
                    instr = False
            elif c == '"':
                instr = True
            elif c == "'" and src[i + 2:i + 3] == "'":
                i += 2
            elif c == "{":
                depth += 1
            elif c == "}":
                depth -= 1
            i += 1
        yield m.group(1), src[m.end():i - 1]


def consts_of(body):
    c = {}
    for m in re.finditer(r"const\s+(\w+)\s*:[^=]*=\s*", body):
        j = m.end()
        depth, instr, k = 0, False, j
        while k < len(body):
            ch = body[k]
            if instr:
                if ch == "\\":
                    k += 1
                elif ch == '"':
                    instr = False
            elif ch == '"':
                instr = True
            elif ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            elif ch == ";" and depth == 0:
                break
            k += 1
        c[m.group(1)] = body[j:k].strip()
    for m in re.finditer(r"let\s+(?:mut\s+)?(\w+)\s*=\s*", body):
        j = m.end()
        depth, instr, k = 0, False, j
        while k < len(body):
            ch = body[k]
            if instr:
                if ch == "\\":
                    k += 1
                elif ch == '"':
                    instr = False
            elif ch == '"':
                instr = True
            elif ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            elif ch == ";" and depth == 0:
                break
            k += 1
        val = body[j:k].strip()
        mm = re.match(r"\[0u8;\s*(\d+)\]$", val)
        if mm:  # zero array + element assignments (deflate_medium_bypass)
            arr = [0] * int(mm.group(1))
            for a in re.finditer(r"%s\[(\d+)\]\s*=\s*(0x[0-9a-fA-F]+|\d+)\s*;" % m.group(1), body):
                arr[int(a.group(1))] = int(a.group(2), 0)
            val = "[" + ",".join(str(x) for x in arr) + "]"
        if m.group(1) not in c and (val.startswith(("&[", "[", 'b"', '"'))):
            c[m.group(1)] = val
    return c


def extract(path, vectors):
    src = strip_comments(open(os.path.join(REF, path), encoding="utf-8").read())
    for name, body in function_spans(src):
        if name == "fuzz_based_test":
            continue
        consts = consts_of(body)
        for m in re.finditer(r"fuzz_based_test\(", body):
            i = m.end()
            depth, k, instr = 1, i, False
            while depth:
                ch = body[k]
                if instr:
                    if ch == "\\":
                        k += 1
                    elif ch == '"':
                        instr = False
                elif ch == '"':
                    instr = True
                elif ch in "([{":
                    depth += 1
                elif ch in ")]}":
                    depth -= 1
                k += 1
            args = split_args(body[i:k - 1])
            if len(args) != 3:
                continue
            try:
                exp = parse_bytes(args[2], consts)
                if not exp:
                    continue
                inp = parse_bytes(args[0], consts)
                cfgexpr = args[1]
                if cfgexpr.strip() == "config":
                    mm = re.search(r"let\s+config\s*=\s*(DeflateConfig\s*\{.*?\});", body, re.S)
                    cfgexpr = mm.group(1)
                cfg = parse_config(cfgexpr)
            except Exception as ex:  # noqa: BLE001
                print("skip %s:%s (%s)" % (path, name, ex), file=sys.stderr)
                continue
            vectors.append({"source": "%s:%s" % (path, name), "config": cfg, "input": inp.hex(), "expected": exp.hex()})
        if "EXPECTED" in consts and "compress_slice" in body:
            mm = re.search(r"let\s+config\s*=\s*(DeflateConfig\s*\{.*?\});", body, re.S)
            inp = consts.get("input") or consts.get("INPUT")
            if mm and inp:
                try:
                    vectors.append({"source": "%s:%s" % (path, name), "config": parse_config(mm.group(1)),
                                    "input": parse_bytes(inp, consts).hex(), "expected": parse_bytes(consts["EXPECTED"], consts).hex()})
                except Exception as ex:  # noqa: BLE001
                    print("skip %s:%s (%s)" % (path, name, ex), file=sys.stderr)


def main():
    vectors = []
    extract("zlib-rs/src/deflate.rs", vectors)
    extract("test-libz-rs-sys/src/deflate.rs", vectors)
    # libz-rs-sys/src/lib.rs compress doctest: "Ferris" -> 14 bytes
    src = open(os.path.join(REF, "libz-rs-sys/src/lib.rs"), encoding="utf-8").read()
    m = re.search(r'let input = "Ferris";.*?assert_eq!\(\s*dest,\s*vec!\[([^\]]*)\]', src, re.S)
    if m:
        exp = bytes(int(x.strip(), 0) for x in m.group(1).split(",") if x.strip())
        vectors.append({"source": "libz-rs-sys/src/lib.rs:compress doctest", "config": parse_config("DeflateConfig::default()"),
                        "input": b"Ferris".hex(), "expected": exp.hex()})
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "deflate_vectors.json")
    json.dump({"reference": "trifectatechfoundation/zlib-rs v0.6.7 (/root/reference)", "vectors": vectors}, open(out, "w"), indent=1)
    print("wrote %d vectors to %s" % (len(vectors), out))
    for v in vectors:
        print("  ", v["source"], v["config"], len(v["input"]) // 2, "->", len(v["expected"]) // 2)


if __name__ == "__main__":
    main()


# ---------------------------------------------------------------------------------------------
# inflate direction: hand-made bitstreams of test-libz-rs-sys/src/inflate.rs (try_inflate) and the
# small binary fixtures of test-libz-rs-sys/src/test-data (copied as data, with their expected
# CRC-32 / length computed by system zlib here).
# ---------------------------------------------------------------------------------------------
def extract_inflate():
    import base64
    import zlib as _z
    vec = []
    src = strip_comments(open(os.path.join(REF, "test-libz-rs-sys/src/inflate.rs"), encoding="utf-8").read())
    for name, body in function_spans(src):
        m = re.search(r"try_inflate\(\s*(&\[[^\]]*\])\s*,\s*(Z_\w+)\s*,?\s*\)", body, re.S)
        if not m or name in ("try_inflate",):
            continue
        data = parse_bytes(m.group(1), {})
        expect = m.group(2)
        # try_inflate(): expected_err >= 0 -> raw inflate (windowBits -15); a non-Z_OK expectation
        # means inflate() must return Z_DATA_ERROR (test-libz-rs-sys/src/inflate.rs:633-690)
        vec.append({"source": "test-libz-rs-sys/src/inflate.rs:%s" % name, "wrap": 0 if expect in ("Z_OK", "Z_STREAM_END") else 3,
                    "input": data.hex(), "expect": "ok" if expect == "Z_OK" else "data_error"})
    files = []
    td = os.path.join(REF, "test-libz-rs-sys/src/test-data")
    fixtures = [("window-match-bug.zraw", 0), ("op-len-edge-case.zraw", 0), ("text.gz", 2), ("issue-109.gz", 2)]
    fixtures += [("compression-corpus/" + f, 2) for f in sorted(os.listdir(os.path.join(td, "compression-corpus")))]
    for rel, wrap in fixtures:
        raw = open(os.path.join(td, rel), "rb").read()
        out = _z.decompress(raw, {0: -15, 2: 31}[wrap])
        files.append({"source": "test-libz-rs-sys/src/test-data/" + rel, "wrap": wrap, "data_b64": base64.b64encode(raw).decode(),
                      "out_len": len(out), "out_crc32": _z.crc32(out), "out_adler32": _z.adler32(out)})
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "inflate_vectors.json")
    json.dump({"reference": "trifectatechfoundation/zlib-rs v0.6.7 (/root/reference)", "bitstreams": vec, "files": files},
              open(out, "w"), indent=1)
    print("wrote %d bitstreams + %d files to %s" % (len(vec), len(files), out))
    for v in vec:
        print("  ", v["source"].split(":")[1], v["wrap"], v["expect"], len(v["input"]) // 2)


if __name__ == "__main__":
    extract_inflate()


# ---------------------------------------------------------------------------------------------
# ABI facts and real-data fixtures (round 2)
#   zlib_symbol_versions.json   symbol -> version node of libz-rs-sys/include/zlib.map (what `readelf --dyn-syms`
#                               of a drop-in libz must show), plus the names the map keeps local
#   fixtures/*.xz               the three real files the reference's own deflate tests round-trip
#                               (test-libz-rs-sys/src/deflate.rs:1982-2003): lcet10.txt, paper-100k.pdf,
#                               fireworks.jpg -- stored xz-compressed (data, not source) with their CRC-32 in
#                               fixtures/manifest.json
# ---------------------------------------------------------------------------------------------
def extract_abi_and_fixtures():
    import lzma
    import zlib as _z
    here = os.path.dirname(os.path.abspath(__file__))
    text = open(os.path.join(REF, "libz-rs-sys/include/zlib.map"), encoding="utf-8").read()
    versions, local = {}, []
    for m in re.finditer(r"(ZLIB_[0-9.]+)\s*\{(.*?)\}", text, re.S):
        node, body, is_local = m.group(1), m.group(2), False
        for tok in re.split(r"[;\s]+", body):
            if tok in ("global:", "local:"):
                is_local = tok == "local:"
            elif tok:
                (local.append(tok) if is_local else versions.__setitem__(tok, node))
    json.dump({"reference": "libz-rs-sys/include/zlib.map", "versions": versions, "local": local},
              open(os.path.join(here, "zlib_symbol_versions.json"), "w"), indent=1, sort_keys=True)
    print("wrote %d versioned symbols" % len(versions))
    fx = os.path.join(here, "fixtures")
    os.makedirs(fx, exist_ok=True)
    manifest = []
    for name in ("lcet10.txt", "paper-100k.pdf", "fireworks.jpg"):
        raw = open(os.path.join(REF, "test-libz-rs-sys/src/test-data", name), "rb").read()
        open(os.path.join(fx, name + ".xz"), "wb").write(lzma.compress(raw, preset=9 | lzma.PRESET_EXTREME))
        manifest.append({"name": name, "source": "test-libz-rs-sys/src/test-data/" + name, "bytes": len(raw), "crc32": _z.crc32(raw)})
    json.dump(manifest, open(os.path.join(fx, "manifest.json"), "w"), indent=1)
    print("wrote fixtures:", [(m["name"], m["bytes"]) for m in manifest])


if __name__ == "__main__":
    extract_abi_and_fixtures()
