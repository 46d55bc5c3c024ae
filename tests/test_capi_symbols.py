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
    assert lib.mdx_abi_version() == 4       # 4: mdx_batch::lowq
    import re
    hdr = (pathlib.Path(__file__).resolve().parent.parent / "include" / "mdx.h").read_text()
    assert int(re.search(r"#define MDX_ABI_VERSION (\d+)", hdr).group(1)) == 4
    assert lib.mdx_strerror(0) == b"ok"
    assert b"contig" in lib.mdx_strerror(-6)


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
