#!/usr/bin/env python3
"""sha256 (first 16 hex digits) over everything the device code is built from: jrl-walkgen_amd/csrc/* and the two headers the
kernels include (include/wg_mpc.h, include/wg_trig.h), in name order -- with comments and white space taken out first, so that
rewording a comment does not make a measured profile set look stale (string literals are not special-cased: none of these
files holds a "//" or "/*" inside one that matters to the device code).  Filed audits and profiles carry it (tools/isa_audit.py,
tools/save_round_profiles.py): a file whose hash is not HEAD's describes another kernel (tests/test_docs_numbers.py fails on it,
bench.py marks the traffic figure it scales from such a file as stale)."""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "jrl-walkgen_amd", "csrc")
    files = [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith((".hip", ".hpp", ".cpp"))]
    files += [os.path.join(ROOT, "include", f) for f in ("wg_mpc.h", "wg_trig.h")]
    for f in files:
        txt = open(f, encoding="utf-8").read()
        txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)          # block comments
        txt = re.sub(r"//[^\n]*", " ", txt)                        # line comments
        txt = re.sub(r"\s+", " ", txt)
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(txt.encode("utf-8"))
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_hash())
