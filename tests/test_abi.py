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


def test_struct_layout_matches_header():
    # sizes implied by include/ofdm_hip.h (natural alignment, 8-byte pad_seed at the end)
    assert ctypes.sizeof(_abi.ofdm_c32) == 8
    assert ctypes.sizeof(_abi.ofdm_chan) == 40
    assert ctypes.sizeof(_abi.ofdm_stats) == 72
    cfg = _abi.ofdm_cfg
    assert cfg.constellation.offset == 28
    assert cfg.known_symbol.offset == 28 + 8 * 256
    assert cfg.tx_amplitude.offset == 28 + 8 * 256 + 8 * 4096
    assert cfg.taps.offset == cfg.ntaps.offset + 4
    assert cfg.whitening_mask.offset == cfg.taps.offset + 4 * 512
    assert ctypes.sizeof(cfg) == 41040


def test_create_rejects_bad_abi_or_config_without_a_gpu():
    lib = _abi.load()
    h = ctypes.c_void_p(None)
    cfg = _abi.ofdm_cfg()
    cfg.struct_size = 12                      # wrong size: refused before any HIP call
    assert lib.ofdm_create(ctypes.byref(cfg), ctypes.byref(h)) == _abi.OFDM_E_INVAL
    assert b"ABI" in lib.ofdm_last_error(None)
