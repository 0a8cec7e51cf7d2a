"""Fingerprint of the kernel sources a library build comes from, so that measurements kept under ``profiles/`` can
say which build they belong to and a benchmark line can refuse to quote counters of another build."""
import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def kernel_sources_sha():
    """First 16 hex digits of the SHA-256 over the HIP sources and headers of the library, in name order."""
    h = hashlib.sha256()
    for name in sorted(os.listdir(_CSRC)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            with open(os.path.join(_CSRC, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]
