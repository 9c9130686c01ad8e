#!/usr/bin/env python3
"""Extract the recorded spectrum-sensing blocks from the reference's run logs
(output.txt, output_with_detection.txt, crap.txt) into tests/golden/sense_blocks.json.

Each block in those logs is the stdout of one pass of sense_loop
(sensing_and_tramsmitting_first.py:204-248): 256 lines "freq power bit" printed at
:231/:238 after the 10-message average, the 1e-4 threshold (:222) and the half swap
(:229-241), followed by the "Carrier map = <hex>" line run_transmiter prints (:288)
from hex_conv's return value (:242,:251-276).  Only the numbers are kept: the text of
each float exactly as Python 2 printed it, the bit, and the hex string.

Run in the build container (needs /root/reference); the JSON is committed.
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sense_blocks.json")

row = re.compile(r"^(\d+\.\d+) (\S+) ([01])$")
blocks = []
for name in ("output.txt", "output_with_detection.txt", "crap.txt"):
    cur = None
    with open(os.path.join(REF, name), "r", errors="replace") as f:
        for lineno, line in enumerate(f, 1):
            line = line.rstrip("\r\n")
            if line.startswith("Sensing the spectrum"):
                cur = {"file": name, "line": lineno, "freq": [], "power": [], "bit": []}
                continue
            m = row.match(line)
            if m and cur is not None:
                cur["freq"].append(m.group(1))
                cur["power"].append(m.group(2))
                cur["bit"].append(int(m.group(3)))
                continue
            if line.startswith("Carrier map =") and cur is not None and cur["freq"]:
                cur["carrier_map"] = line.split("=", 1)[1].strip()
                blocks.append(cur)
                cur = None
    # a trailing block without its "Carrier map" line is incomplete: dropped

grids = {}
for b in blocks:
    assert len(b["freq"]) == len(b["power"]) == len(b["bit"]) == 256, (b["file"], b["line"], len(b["freq"]))
    assert len(b["carrier_map"]) == 64
    # the printed grid is identical for every block at one centre frequency (bin 127 of
    # the in-order list is the centre itself): keep one copy per centre
    centre = b["freq"][127]
    assert grids.setdefault(centre, b["freq"]) == b["freq"]
    b["center_freq"] = centre
    del b["freq"]

doc = {
    "source": "rubiruchi/ofdm_uhd run logs output.txt / output_with_detection.txt / crap.txt",
    "producer": "sensing_and_tramsmitting_first.py:204-248 (sense_loop) and :288 (Carrier map print)",
    "fft_size": 256,
    "samp_rate": 6250000.0,
    "freq_grids": grids,
    "threshold": 0.00010,
    "final_hex_conv": {"bits": "0000111111111111", "note": "final_hex_conv.py:37 input; LSB-first nibbles"},
    "blocks": blocks,
}
with open(OUT, "w") as f:
    json.dump(doc, f, separators=(",", ":"))
print("wrote", OUT, len(blocks), "blocks", os.path.getsize(OUT), "bytes")
