#!/usr/bin/env python
"""SHA-256 of the sources libeco_hip.so is built from -- the same digest csrc/Makefile compiles into the library
(eco_source_digest()): csrc/*.hip listed in the Makefile's SRCS, csrc/eco_common.h, csrc/eco_device.h, include/eco_hip.h
and the Makefile itself, concatenated in sorted path order.  `python tools/srcdigest.py` prints it."""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "eco-efficient-video-understanding_amd", "csrc")


def digest_files():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    names = re.findall(r"\$\(CSRC\)(\w+\.(?:hip|h))", re.search(r"^SRCS\s*:=(.*)$", mk, re.M).group(1) + " " +
                       re.search(r"^HDRS\s*:=(.*)$", mk, re.M).group(1))
    files = [os.path.join(CSRC, n) for n in names] + [os.path.join(ROOT, "include", "eco_hip.h"), os.path.join(CSRC, "Makefile")]
    return sorted(set(files))       # make's $(sort) orders the absolute paths bytewise, as sorted() does for ASCII


def source_digest() -> str:
    h = hashlib.sha256()
    for p in digest_files():
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


if __name__ == "__main__":
    print(source_digest())
