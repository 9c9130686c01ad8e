"""The C-ABI library loads and exports every symbol include/ofdm_hip.h declares.
No compute calls here (no GPU in this container)."""
import ctypes
import os
import re

from ofdm_uhd_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    hdr = open(os.path.join(ROOT, "include", "ofdm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(ofdm_[a-z_0-9]+)\s*\(", hdr)))


def test_header_and_python_export_lists_agree():
    assert _declared_functions() == sorted(_abi.EXPORTS)


def test_library_loads_and_exports_everything():
    lib = _abi.load()
    for name in _declared_functions():
        assert hasattr(lib, name), name
    assert lib.ofdm_abi_version() == _abi.OFDM_ABI_VERSION
    assert lib.ofdm_kernel_name(_abi.K_SYNC) == b"k_sync"


def test_struct_layout_matches_header(tmp_path):
    """sizeof / offsetof as the C compiler sees include/ofdm_hip.h == the ctypes mirror."""
    import subprocess
    structs = {"ofdm_cfg": _abi.ofdm_cfg, "ofdm_chan": _abi.ofdm_chan, "ofdm_stats": _abi.ofdm_stats,
               "ofdm_sense_cfg": _abi.ofdm_sense_cfg, "ofdm_c32": _abi.ofdm_c32}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "ofdm_hip.h"', 'int main(void){']
    for name, st in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (name, name))
        for f, _ in st._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (name, f, name, f))
    lines.append('return 0;}')
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = str(tmp_path / "layout")
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", exe, str(src)])
    got = dict(l.split() for l in subprocess.check_output([exe]).decode().splitlines())
    for name, st in structs.items():
        assert int(got[name]) == ctypes.sizeof(st), name
        for f, _ in st._fields_:
            assert int(got["%s.%s" % (name, f)]) == getattr(st, f).offset, (name, f)
    assert ctypes.sizeof(_abi.ofdm_cfg) == 41040 + 1032 + 16   # (+ sync_mode, fixed_nsymbols, fixed_freq_offset, reserved0)


def test_create_rejects_bad_abi_or_config_without_a_gpu():
    lib = _abi.load()
    h = ctypes.c_void_p(None)
    cfg = _abi.ofdm_cfg()
    cfg.struct_size = 12                      # wrong size: refused before any HIP call
    assert lib.ofdm_create(ctypes.byref(cfg), ctypes.byref(h)) == _abi.OFDM_E_INVAL
    assert b"ABI" in lib.ofdm_last_error(None)
