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


def test_deflate_core_is_read_by_zlib(tmp_path):
    """The DEFLATE encoder of the device BGZF writer (mapdamage_amd/csrc/mdx_deflate.h: one block per lane on the GPU,
    tests/test_gpu_decode.py) compiled for the host: every block it writes — BGZF member around it — must inflate under zlib
    (which checks the gzip header, CRC32 and ISIZE itself) to the bytes that went in.  Empty and tiny blocks, incompressible
    ones (a stored block), runs, repeats at every distance, alphabets skewed enough to need the length limit of 15 bits,
    BAM-like records."""
    exe = tmp_path / "deflate_check"
    subprocess.check_call(["g++", "-O2", "-I", str(ROOT / "mapdamage_amd" / "csrc"), str(ROOT / "tests" / "native" / "deflate_check.cpp"),
                           "-lz", "-o", str(exe)])
    rnd = random.Random(1)
    full = 0xFF00
    datas = [b"", b"a", b"ab", b"abc", b"abcd", b"aaaa", b"abc" * 1000, bytes(rnd.getrandbits(8) for _ in range(60000)),
             bytes(rnd.choice(b"ACGT") for _ in range(full)), b"\x00" * full, os.urandom(100), os.urandom(full),
             bytes(rnd.choice(b"ACGTN!#$%&'()*+,-./0123456789") for _ in range(30000)),
             b"ACGT" * 3000 + os.urandom(30000) + b"TTTTGGGG" * 2000, bytes([i % 251 for i in range(65000)]),
             bytes(rnd.choice(b"ab") for _ in range(full)), b"x" * 258 + b"y" + b"x" * 259 + b"z" + b"x" * 40000]
    fib = [1, 1]
    while len(fib) < 40:
        fib.append(fib[-1] + fib[-2])
    skew = bytearray()
    for i, f in enumerate(fib[:24]):
        skew += bytes([i]) * min(f, 20000)
    skew = list(skew)
    rnd.shuffle(skew)
    datas.append(bytes(skew[:full]))
    for _ in range(120):
        n = rnd.randint(0, full)
        kind = rnd.randint(0, 3)
        if kind == 0:
            d = os.urandom(n)
        elif kind == 1:
            d = bytes(rnd.choice(b"ACGT") for _ in range(n))
        elif kind == 2:
            w = os.urandom(rnd.randint(1, 300))
            d = (w * (n // len(w) + 1))[:n]
        else:
            d = bytearray()
            while len(d) < n:
                if d and rnd.random() < 0.5:
                    st = rnd.randint(0, len(d) - 1)
                    d += d[st:st + rnd.randint(1, 400)]
                else:
                    d += os.urandom(rnd.randint(1, 50))
            d = bytes(d[:n])
        datas.append(d)
    blob = bytearray()
    for d in datas:
        blob += struct.pack("<I", len(d)) + d
    path = tmp_path / "cases.bin"
    path.write_bytes(blob)
    out = subprocess.run([str(exe), str(path)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok %d cases" % len(datas)), out.stdout + out.stderr
