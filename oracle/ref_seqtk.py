"""Loader for oracle/_ref/seqtk*.so — the reference's own native extension
(mapdamage/seqtk/seqtk.c) compiled by `make -C oracle ref`.  TEST INFRASTRUCTURE ONLY."""

import glob
import importlib.util
import pathlib
import subprocess

_HERE = pathlib.Path(__file__).resolve().parent


def load(build=True):
    found = glob.glob(str(_HERE / "_ref" / "seqtk*.so"))
    if not found and build and pathlib.Path("/root/reference/mapdamage/seqtk/seqtk.c").exists():
        subprocess.check_call(["make", "-s", "-C", str(_HERE), "ref"])
        found = glob.glob(str(_HERE / "_ref" / "seqtk*.so"))
    if not found:
        return None
    spec = importlib.util.spec_from_file_location("seqtk", found[0])
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module
