"""Test infrastructure: a minimal BAM writer (BGZF blocks with the BC extra field + EOF block, SAM spec 4.1 / 4.2) and a
Python restatement of the reference's record order and of get_score_diff, for tests/test_bam_ingest.py.
  order:      HtsReader::get_next_read_in_order (src/utilities/hts_reader.cpp:166-303: records of one core.pos sorted by
              gt_pos_seq_same_pos, hts_utils.hpp:83-108, taken from the back) + the heap of HtsParallelReader
              (hts_parallel_reader.cpp:66-136, gt_pos_seq hts_utils.hpp:48-81); equal keys: file order (unspecified upstream)
  score diff: get_score_diff (src/typer/alignment.cpp:140-325)"""
import struct
import zlib

import numpy as np


def bgzf(data, block=60000):
    out = bytearray()
    for at in list(range(0, len(data), block)) + [None]:
        chunk = b"" if at is None else data[at:at + block]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(chunk) + c.flush()
        bsize = len(comp) + 25  # total block size - 1
        out += struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize)
        out += comp + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    return bytes(out)


def pack_seq(codes):
    """4-bit codes [L] -> BAM packed bases"""
    c = list(int(x) for x in codes) + [0]
    return bytes((c[2 * i] << 4) | c[2 * i + 1] for i in range((len(codes) + 1) // 2))


def aux_field(tag, typ, value):
    t = tag.encode() + typ.encode()
    if typ == "Z":
        return t + value.encode() + b"\0"
    if typ == "A":
        return t + value.encode()
    if typ == "B":  # (subtype, values)
        sub, vals = value
        return t + sub.encode() + struct.pack("<I", len(vals)) + b"".join(struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub], v) for v in vals)
    return t + struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[typ], value)


def record(name, flag, tid, pos, mapq, cigar, mtid, mpos, tlen, codes, aux=(), qual=None):
    """cigar: [(op, len)], op in 'MIDNSHP=X'; aux: [(tag, type, value)]"""
    nm = name.encode() + b"\0"
    cig = b"".join(struct.pack("<I", (n << 4) | "MIDNSHP=X".index(op)) for op, n in cigar)
    body = struct.pack("<iiBBHHHIiii", tid, pos, len(nm), mapq, 4680, len(cigar), flag, len(codes), mtid, mpos, tlen)
    body += nm + cig + pack_seq(codes) + (bytes([30] * len(codes)) if qual is None else bytes(int(q) for q in qual)) + b"".join(aux_field(*a) for a in aux)
    return struct.pack("<i", len(body)) + body


def write_bam(path, refs, header_text, records, index=None, poison=False, csi=None):
    """refs: [(name, length)]; records: bytes from record(), already in file order.
    index: [(tid, pos, end)] per record -> also writes path + ".bai" (SAM spec 5.2: bins, chunks, 16 kb linear index);
    poison: a BGZF member of garbage between the header and the records -- a reader that scans from the head fails on it,
    one that seeks through the index never sees it.
    csi = (min_shift, depth): the index goes to path + ".csi" instead (htslib's CSI: bins of a free geometry with their
    loffset, the whole index BGZF-compressed)."""
    head = b"BAM\1" + struct.pack("<i", len(header_text)) + header_text.encode() + struct.pack("<i", len(refs))
    for name, length in refs:
        head += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", length)
    out = bytearray(bgzf(head)[:-28])  # (without the end-of-file member)
    if poison:
        out += bgzf(b"\xff" * 300)[:-28]
    voffs = []
    block, block_start = bytearray(), len(out)
    pieces = []
    for r in records:
        if len(block) + len(r) > 60000 and block:
            pieces.append(bytes(block))
            comp = bgzf(bytes(block))[:-28]
            out += comp
            block, block_start = bytearray(), len(out)
        voffs.append((block_start << 16) | len(block))
        block += r
    if block:
        out += bgzf(bytes(block))[:-28]
    end_voff = len(out) << 16
    out += bgzf(b"")  # the end-of-file member
    open(path, "wb").write(bytes(out))
    if index is None:
        return
    assert len(index) == len(records)

    def reg2bin(beg, end):
        end -= 1
        for shift, first in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
            if beg >> shift == end >> shift:
                return first + (beg >> shift)
        return 0
    if csi is not None:
        min_shift, depth = csi

        def reg2bin_csi(beg, end):
            end -= 1
            s, t = min_shift, ((1 << (depth * 3)) - 1) // 7
            for l in range(depth, 0, -1):
                if beg >> s == end >> s:
                    return t + (beg >> s)
                s += 3
                t -= 1 << ((l - 1) * 3)
            return 0
        out_csi = bytearray(b"CSI\1" + struct.pack("<iii", min_shift, depth, 0) + struct.pack("<i", len(refs)))
        for tid in range(len(refs)):
            bins = {}
            for k, (t, pos, end) in enumerate(index):
                if t != tid:
                    continue
                v0, v1 = voffs[k], voffs[k + 1] if k + 1 < len(voffs) else end_voff
                e = bins.setdefault(reg2bin_csi(pos, max(end, pos + 1)), dict(chunks=[], lo=v0))
                if e["chunks"] and e["chunks"][-1][1] == v0:
                    e["chunks"][-1][1] = v1
                else:
                    e["chunks"].append([v0, v1])
            # loffset of a bin: the offset of the first record that overlaps the bin's interval (records in other bins too)
            for b in bins:
                level, first = 0, 0
                while not (first <= b < first + (1 << (3 * level))):
                    first += 1 << (3 * level)
                    level += 1
                shift = min_shift + 3 * (depth - level)
                b0, b1 = (b - first) << shift, ((b - first) + 1) << shift
                bins[b]["lo"] = min(voffs[k] for k, (t, pos, end) in enumerate(index) if t == tid and pos < b1 and max(end, pos + 1) > b0)
            out_csi += struct.pack("<i", len(bins))
            for b in sorted(bins):
                out_csi += struct.pack("<IQi", b, bins[b]["lo"], len(bins[b]["chunks"])) + b"".join(struct.pack("<QQ", c0, c1) for c0, c1 in bins[b]["chunks"])
        open(path + ".csi", "wb").write(bgzf(bytes(out_csi)))
        return
    bai = bytearray(b"BAI\1" + struct.pack("<i", len(refs)))
    for tid in range(len(refs)):
        bins, linear = {}, {}
        for k, (t, pos, end) in enumerate(index):
            if t != tid:
                continue
            v0, v1 = voffs[k], voffs[k + 1] if k + 1 < len(voffs) else end_voff
            chunks = bins.setdefault(reg2bin(pos, max(end, pos + 1)), [])
            if chunks and chunks[-1][1] == v0:
                chunks[-1][1] = v1
            else:
                chunks.append([v0, v1])
            for w in range(pos >> 14, ((max(end, pos + 1) - 1) >> 14) + 1):
                linear[w] = min(linear.get(w, v0), v0)
        bai += struct.pack("<i", len(bins))
        for b in sorted(bins):
            bai += struct.pack("<Ii", b, len(bins[b])) + b"".join(struct.pack("<QQ", c0, c1) for c0, c1 in bins[b])
        n_intv = max(linear) + 1 if linear else 0
        bai += struct.pack("<i", n_intv)
        last = 0
        for w in range(n_intv):
            last = linear.get(w, last)
            bai += struct.pack("<Q", last)
    open(path + ".bai", "wb").write(bytes(bai))


def score_diff(aux):
    """get_score_diff over [(tag, type, value)] in file order"""
    a = x = -1
    for tag, typ, value in aux:
        if typ in "AZ" or typ == "f":
            continue
        if typ not in "cCsSiI":
            break  # a type the parser does not know (B, H, d): it stops
        if tag == "AS":
            a = value
        elif tag == "XS":
            x = value
    if a == -1 or a < x:
        return 0
    diff = a - (0 if x == -1 else x)
    return diff if diff < 255 else 255


def merged_order(files):
    """files: per file a list of dicts (tid, pos, codes) in file order -> [(file, index)] in the order the reference reads them"""
    def seq_key(r):
        return (len(r["codes"]), pack_seq(r["codes"]))
    per_file = []
    for f, recs in enumerate(files):
        out, i = [], 0
        while i < len(recs):
            j = i
            while j < len(recs) and recs[j]["pos"] == recs[i]["pos"]:
                j += 1
            out += sorted(range(i, j), key=lambda k: seq_key(recs[k]))  # stable
            i = j
        per_file.append(out)
    heads = [0] * len(files)
    order = []
    while True:
        best = None
        for f in range(len(files)):
            if heads[f] < len(per_file[f]):
                r = files[f][per_file[f][heads[f]]]
                key = (r["tid"], r["pos"]) + seq_key(r)
                if best is None or key < best[0]:
                    best = (key, f)
        if best is None:
            return order
        f = best[1]
        order.append((f, per_file[f][heads[f]]))
        heads[f] += 1
