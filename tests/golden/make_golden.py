#!/usr/bin/env python3
"""Generates tests/golden/ntt_tables.json from the reference's twiddle-table DATA.

Run in the build container only (it reads /root/reference, which does not exist on the GPU box):
    python3 tests/golden/make_golden.py
The fixture holds data only -- SHA-256 of each of the 8 rows of `tables[]`
(/root/reference/src/constants.cpp:16, layout /root/reference/src/core.cpp:6-17) serialised as
little-endian u64, plus 16 sampled entries per row and the scalar constants of include/values.h
that the hot path depends on.  tests/test_oracle_tables.py checks the oracle's regenerated tables
and constants against it.
"""
import hashlib
import json
import os
import re
import struct

REF = "/root/reference"
N = 2048


def main():
    src = open(os.path.join(REF, "src/constants.cpp")).read()
    m = re.search(r"uint64_t tables\[\]\s*=\s*\{(.*?)\};", src, re.S)
    vals = [int(x.strip().rstrip("UL")) for x in m.group(1).split(",") if x.strip()]
    assert len(vals) == 8 * N
    names = ["inv_p_w", "inv_p_wscaled", "inv_b_w", "inv_b_wscaled", "fwd_p_w", "fwd_p_wscaled", "fwd_b_w", "fwd_b_wscaled"]
    sample_idx = [0, 1, 2, 3, 7, 64, 255, 256, 1023, 1024, 1025, 1500, 2000, 2045, 2046, 2047]
    rows = {}
    for r, name in enumerate(names):
        row = vals[r * N:(r + 1) * N]
        rows[name] = {
            "sha256": hashlib.sha256(struct.pack("<%dQ" % N, *row)).hexdigest(),
            "samples": {str(i): row[i] for i in sample_idx},
        }
    vh = open(os.path.join(REF, "include/values.h")).read()

    def const(name):
        mm = re.search(r"constexpr\s+\w+\s+%s\s*=\s*([0-9]+)" % name, vh)
        return int(mm.group(1))

    consts = {k: const(k) for k in ["p_i", "b_i", "cr0_Q", "cr1_Q", "cr1_p", "cr1_b", "n0", "n1", "n2"]}
    mm = re.search(r"pa_inv_b_i\s*=\s*([0-9]+)UL", vh)
    consts["pa_inv_b_factor"] = int(mm.group(1))
    mm = re.search(r"b_inv_pa_i\s*=\s*([0-9]+)UL", vh)
    consts["b_inv_pa_factor"] = int(mm.group(1))
    mm = re.search(r"qprime_mods\[37\]\s*=\s*\{(.*?)\}", vh, re.S)
    consts["qprime_mods"] = [int(x) for x in mm.group(1).split(",")]
    out = {"source": "menonsamir/spiral src/constants.cpp tables[] + include/values.h", "rows": rows, "constants": consts}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ntt_tables.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote ntt_tables.json")


if __name__ == "__main__":
    main()
