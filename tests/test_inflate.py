"""The BGZF readers' own DEFLATE decoder (graphtyper_amd/csrc/gtx_inflate.hpp, through gtx_inflate_raw) against zlib: every
stream zlib's deflate makes of data of several kinds, at every level and strategy, with flushes in the middle (several blocks,
empty stored blocks), is inflated to the same bytes; damaged, truncated and wrongly sized streams are refused or give what zlib
gives -- never a crash (tests/sanitize/reads_driver runs the same decoder under ASan / UBSan on damaged BAM files)."""
import zlib

import numpy as np
import pytest

from graphtyper_amd import lib as gtx


def _data(rng, kind, n):
    if kind == 0:
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()                      # incompressible: stored blocks
    if kind == 1:
        return bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n))                  # four letters: short codes
    if kind == 2:
        return (b"chr20\t1234\t.\tA\tC\t" * (n // 16 + 1))[:n]                        # text with long matches
    if kind == 3:
        return bytes(np.repeat(rng.integers(0, 256, n // 50 + 1, dtype=np.uint8), 50)[:n])  # runs: distance 1
    if kind == 4:
        z = rng.zipf(1.3, n)
        return bytes(np.minimum(z, 255).astype(np.uint8))                               # skewed alphabet: codes of up to 15 bits
    return bytes((np.arange(n) * 2654435761 >> 24).astype(np.uint8))


def _deflate(data, level, strategy, flush_at):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    out, at = b"", 0
    for cut, mode in flush_at:
        out += c.compress(data[at:cut]) + c.flush(mode)
        at = cut
    return out + c.compress(data[at:]) + c.flush()


@pytest.mark.parametrize("seed", range(6))
def test_equals_zlib(seed):
    rng = np.random.default_rng(seed)
    for it in range(120):
        n = int(rng.integers(0, 40)) if it % 8 == 0 else int(rng.integers(0, 65537))
        data = _data(rng, int(rng.integers(0, 6)), n)
        flushes = sorted((int(rng.integers(0, n + 1)), int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))) for _ in range(int(rng.integers(0, 4)))) if it % 3 == 0 else []
        comp = _deflate(data, int(rng.integers(0, 10)), int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])), flushes)
        assert gtx.inflate_raw(comp, len(data)) == data
        # damaged: refused, or what zlib makes of it
        bad = bytearray(comp)
        how = it % 4
        want = len(data)
        if how == 0 and bad:
            bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        elif how == 1 and bad:
            bad = bad[:int(rng.integers(0, len(bad)))]
        elif how == 2:
            want = want + 1 if it % 8 < 4 or want == 0 else want - 1
        else:
            bad += b"\x00\x01"
        try:
            got = gtx.inflate_raw(bytes(bad), want)
        except gtx.GtxError:
            continue
        d = zlib.decompressobj(-15)
        assert d.decompress(bytes(bad)) == got and d.eof


def test_hand_made_streams():
    assert gtx.inflate_raw(b"\x03\x00", 0) == b""                                       # an empty fixed block
    assert gtx.inflate_raw(b"\x01\x00\x00\xff\xff", 0) == b""                           # an empty stored block
    assert gtx.inflate_raw(b"\x01\x03\x00\xfc\xffabc", 3) == b"abc"
    assert gtx.inflate_raw(b"\x00\x00\x00\xff\xff" + b"\x03\x00", 0) == b""             # not-last stored block, then the last one
    for bad, n in ((b"", 0), (b"\x07\x00", 0), (b"\x01\x03\x00\xfc\xfeabc", 3), (b"\x01\x03\x00\xfc\xffab", 3), (b"\x03", 1)):
        with pytest.raises(gtx.GtxError):
            gtx.inflate_raw(bad, n)
    # a match that reaches in front of the output: fixed block, length 3 distance 1 as the first symbol
    with pytest.raises(gtx.GtxError):
        gtx.inflate_raw(bytes([0b00000011, 0b00000010, 0]), 3)
