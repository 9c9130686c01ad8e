#!/usr/bin/env python3
"""Generate tests/golden/reference_constants.json from the reference tree.

Runs ONLY in the build container (needs /root/reference); the resulting JSON
is committed and is what travels to the GPU box.  What is taken:

* ``known_symbols_4512_3``   -- parsed as text from ofdm.py:310-325 (no exec)
* ``random_mask_tuple``      -- parsed as text from ofdm_packet_utils.py:195-452
* ``psk.gray_constellation`` -- by importing psk.py (pure ``math``; psk.py:27-60)
* ``qam.constellation``      -- by importing qam.py (pure ``math``; qam.py:29-73)
* sensing KAT                -- the literal input of final_hex_conv.py:40

Everything written is data (numbers), never source text.
"""
import ast
import importlib.util
import json
import os
import re
import sys

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_constants.json")


def _literal_after(path, name, open_ch, close_ch):
    src = open(path, "r", errors="replace").read()
    m = re.search(r"^%s\s*=\s*\%s" % (re.escape(name), open_ch), src, re.M)
    if not m:
        raise SystemExit("cannot find %s in %s" % (name, path))
    start = m.end() - 1
    end = src.index(close_ch, start)
    return ast.literal_eval(src[start:end + 1])


def _import_pure(path, modname):
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ks = list(_literal_after(os.path.join(REF, "ofdm.py"), "known_symbols_4512_3", "[", "]"))
    mask = list(_literal_after(os.path.join(REF, "ofdm_packet_utils.py"), "random_mask_tuple", "(", ")"))
    assert len(ks) == 4512 and set(ks) == {-1, 1}
    assert len(mask) == 4096 and mask[:8] == [255, 63, 0, 16, 0, 12, 0, 5]

    psk = _import_pure(os.path.join(REF, "psk.py"), "ref_psk")
    qam = _import_pure(os.path.join(REF, "qam.py"), "ref_qam")

    def pts(lst):
        return [[float(c.real), float(c.imag)] for c in lst]

    out = {
        "_about": "constants extracted from rubiruchi/ofdm_uhd by tests/golden/make_fixtures.py",
        "known_symbols_4512_3": "".join("+" if v > 0 else "-" for v in ks),
        "random_mask_hex": bytes(mask).hex(),
        "psk_gray_constellation": {str(k): pts(v) for k, v in psk.gray_constellation.items()},
        "psk_constellation": {str(k): pts(v) for k, v in psk.constellation.items()},
        "qam_constellation": {str(k): pts(v) for k, v in qam.constellation.items()},
        # ofdm.py:91 / ofdm.py:225
        "mods": {"bpsk": 2, "qpsk": 4, "8psk": 8, "qam8": 8, "qam16": 16, "qam64": 64, "qam256": 256},
        # ofdm.py:96 literal
        "qpsk_rot": [0.707, 0.707],
    }
    with open(OUT, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
