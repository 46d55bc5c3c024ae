"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/mdx.h declares,
and refuses to run without a GPU (no silent fallback)."""

import ctypes
import pathlib
import re

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    from mapdamage_amd import build, engine
    build.build_lib()
    return engine.load_library()


def declared_symbols():
    text = (ROOT / "include" / "mdx.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mdx_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    from mapdamage_amd import engine
    names = declared_symbols()
    assert len(names) >= 19
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(engine.EXPORTS) == names


def test_abi_version_and_strerror(lib):
    assert lib.mdx_abi_version() == 6       # 6: the FASTA loader, mdx_warm, mdx_host_threads
    import re
    hdr = (pathlib.Path(__file__).resolve().parent.parent / "include" / "mdx.h").read_text()
    assert int(re.search(r"#define MDX_ABI_VERSION (\d+)", hdr).group(1)) == 6
    assert lib.mdx_strerror(0) == b"ok"
    assert b"contig" in lib.mdx_strerror(-6)


def test_ctypes_mirrors_match_the_header(tmp_path):
    """The ctypes structures of mapdamage_amd/engine.py against what a C compiler makes of include/mdx.h: sizes, and the
    offsets of the fields behind the columns (the library writes a whole mdx_batch into the binder's memory)."""
    import shutil
    import subprocess

    from mapdamage_amd.engine import MdxBatch, MdxConfig
    cc = shutil.which("gcc") or shutil.which("cc")
    if not cc:
        pytest.skip("no C compiler")
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mdx.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(mdx_config), sizeof(mdx_batch), offsetof(mdx_batch, qual), '
                   'offsetof(mdx_batch, seq_format), offsetof(mdx_batch, reserved), offsetof(mdx_batch, lowq), '
                   'offsetof(mdx_batch, libsort)); return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.check_call([cc, "-I", str(ROOT / "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == [ctypes.sizeof(MdxConfig), ctypes.sizeof(MdxBatch), MdxBatch.qual.offset, MdxBatch.seq_format.offset,
                   MdxBatch.reserved.offset, MdxBatch.lowq.offset, MdxBatch.libsort.offset]


def test_create_rejects_bad_config(lib):
    from mapdamage_amd.engine import MdxConfig
    ctx = ctypes.c_void_p()
    cfg = MdxConfig(0, 10, 0, 1, 1024, 0, 16)  # length 0
    assert lib.mdx_create(ctypes.byref(cfg), ctypes.byref(ctx)) == -1
    assert not ctx.value


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mapdamage_amd.engine import DamageEngine, MdxError
    with pytest.raises(MdxError):
        DamageEngine([("s", "l")])


def test_product_does_not_import_oracle():
    for path in (ROOT / "mapdamage_amd").rglob("*.py"):
        text = path.read_text()
        assert "import oracle" not in text and "from oracle" not in text, path
