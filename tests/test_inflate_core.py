"""The DEFLATE decoder of the GPU decode path (mapdamage_amd/csrc/mdx_inflate.h) compiled for the host and held
against zlib: valid streams of every block type, random bytes and bit-flipped streams (accepted / rejected alike, same
bytes out).  The same header runs one BGZF block per wavefront on the GPU (tests/test_gpu_decode.py)."""
import os
import pathlib
import random
import struct
import subprocess
import zlib

ROOT = pathlib.Path(__file__).resolve().parent.parent


def _raw(data, level, strategy=zlib.Z_DEFAULT_STRATEGY):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    return c.compress(data) + c.flush()


def test_inflate_core_agrees_with_zlib(tmp_path):
    exe = tmp_path / "inflate_check"
    subprocess.check_call(["g++", "-O2", "-I", str(ROOT / "mapdamage_amd" / "csrc"), str(ROOT / "tests" / "native" / "inflate_check.cpp"),
                           "-lz", "-o", str(exe)])
    rnd = random.Random(1)
    datas = [b"", b"a", b"abc" * 1000, bytes(rnd.getrandbits(8) for _ in range(60000)),
             bytes(rnd.choice(b"ACGT") for _ in range(65536)), b"\x00" * 65536, os.urandom(100),
             bytes(rnd.choice(b"ACGTN!#$%&'()*+,-./0123456789") for _ in range(30000))]
    # one stream, several deflate blocks of different kinds (a stored one behind compressed ones and the other way round)
    datas.append(b"ACGT" * 3000 + os.urandom(30000) + b"TTTTGGGG" * 2000)
    datas.append(os.urandom(20000) + bytes(rnd.choice(b"ACGT") for _ in range(40000)))
    blob = bytearray()
    n = 0
    for d in datas:
        for level in (0, 1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                comp = _raw(d, level, strategy)
                blob += struct.pack("<II", len(comp), len(d)) + comp + d
                n += 1
    for _ in range(300):                                   # random bytes
        g = os.urandom(rnd.randint(1, 300))
        blob += struct.pack("<II", len(g), 0xFFFFFFFF) + g
        n += 1
    for _ in range(1500):                                  # valid streams with a few bits flipped
        d = bytes(rnd.choice(b"ACGTACGTACGTNIIIIII#####") for _ in range(rnd.randint(1, 3000)))
        comp = bytearray(_raw(d, rnd.choice([1, 6, 9])))
        for _ in range(rnd.randint(1, 3)):
            comp[rnd.randrange(len(comp))] ^= 1 << rnd.randrange(8)
        blob += struct.pack("<II", len(comp), 0xFFFFFFFF) + bytes(comp)
        n += 1
    out = subprocess.run([str(exe)], input=bytes(blob), capture_output=True, timeout=120)
    assert out.returncode == 0, out.stdout.decode()[-2000:]
    assert out.stdout.decode().strip().endswith("%d streams, 0 bad" % n)


def test_crc32_pieces_join_to_zlibs_value(tmp_path):
    """mapdamage_amd/csrc/mdx_crc32.h (the GPU decode path's block check): CRCs of 1 KiB pieces joined with the
    append-zeros operator equal zlib's crc32 of the whole, for every length class of a BGZF block."""
    exe = tmp_path / "crc_check"
    subprocess.check_call(["g++", "-O2", "-I", str(ROOT / "mapdamage_amd" / "csrc"), str(ROOT / "tests" / "native" / "crc_check.cpp"),
                           "-lz", "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, timeout=60)
    assert out.returncode == 0 and out.stdout.decode().strip().endswith("0 bad"), out.stdout.decode()
