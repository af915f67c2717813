"""The C-ABI library loads and exports every symbol include/knncf.h declares (no GPU needed)."""
import ctypes
import importlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "knncf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(knncf_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    build = importlib.import_module(pkg.__name__ + ".build")
    build.build()
    kn = importlib.import_module(pkg.__name__ + ".knncf")
    lib = kn.load_library()  # (the package's loader: it binds the library to the HIP runtime PyTorch bundles, one per process)
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/knncf.h but not exported"
    assert sorted(kn.EXPORTS) == names


def test_status_strings_and_create_without_gpu(pkg):
    kn = importlib.import_module(pkg.__name__ + ".knncf")
    lib = kn.load_library()
    assert lib.knncf_status_string(0) == b"ok"
    assert b"gfx950" in lib.knncf_version()
    import torch

    if not torch.cuda.is_available():
        # the product path fails loudly without a device: no CPU fallback
        try:
            kn.Engine(k=3)
        except kn.KnncfError as e:
            assert e.status == kn.E_NODEVICE
        else:
            raise AssertionError("Engine() must fail without a gfx950 device")


def test_group_create_fails_loudly_without_gpu(pkg):
    """knncf_group_*: the exports exist, and without a device (or without librccl) creation fails with a status — no
    fallback path forms a "group" on the CPU"""
    import torch

    kn = importlib.import_module(pkg.__name__ + ".knncf")
    kn.load_library()
    if torch.cuda.is_available():
        return
    try:
        kn.Group([0], k=3)
    except kn.KnncfError as e:
        assert e.status in (kn.E_NODEVICE, kn.E_UNSUPPORTED)
    else:
        raise AssertionError("Group() must fail without a gfx950 device")
