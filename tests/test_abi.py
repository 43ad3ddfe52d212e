"""The C-ABI library loads and exports exactly the entry points include/speakerguard_hip.h declares.
CPU only: no compute call is made (hipcc cross-compiles gfx950 without a GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from speakerguard_amd import _native


def header_functions():
    text = open(os.path.join(ROOT, "include", "speakerguard_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_native.LIB_PATH), "run `make` (or __graft_entry__.build()) first"
    lib = ctypes.CDLL(_native.LIB_PATH)
    names = header_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n
    assert sorted(_native.EXPORTS) == names, "python binding list and header disagree"


def test_version_and_loader():
    lib = _native.load()
    assert lib.sg_version() == 100
    assert lib.sg_xv_num_frames(48000) == 300
    assert lib.sg_xv_num_frames(52960) == 331
    assert lib.sg_xv_num_frames(100) == 0


def test_struct_layouts_match_header(tmp_path):
    """sizeof of every struct that crosses the boundary, as gcc lays out the header's definition (x86-64 SysV), against
    the ctypes mirror: catches field-order / padding drift."""
    import subprocess
    pairs = [("sg_loss_spec", _native.LossSpec), ("sg_dither", _native.Dither), ("sg_pgd_params", _native.PgdParams),
             ("sg_xv_weights", _native.XvWeights), ("sg_feco_params", _native.FecoParams)]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "speakerguard_hip.h"\nint main(void) {\n' +
                   "".join('    printf("%%zu\\n", sizeof(%s));\n' % c for c, _ in pairs) + "    return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    sizes = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    for (cname, ctype), size in zip(pairs, sizes):
        assert ctypes.sizeof(ctype) == size, (cname, ctypes.sizeof(ctype), size)
    assert ctypes.sizeof(_native.LossSpec) == 32 and ctypes.sizeof(_native.Dither) == 48


def test_missing_library_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_native.NativeError):
        _native.load()


def test_model_refuses_cpu_device():
    from speakerguard_amd import synth
    from speakerguard_amd.model.xv_plda import xv_plda
    with pytest.raises(_native.NativeError):
        xv_plda.from_weights(synth.make_xv_weights(), device="cpu")


def test_header_is_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: include/speakerguard_hip.h must compile as C99 on its own."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "speakerguard_hip.h"\nint use(void) { return sg_version(); }\n')
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                        "-c", str(src), "-o", str(tmp_path / "hdr.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
