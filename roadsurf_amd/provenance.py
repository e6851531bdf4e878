"""Which kernel sources a measurement belongs to: one hash over the device and host sources the
library is built from.  tools/make_traffic_json.py stamps the committed counter summary with it and
bench.py reports counter-derived figures only when the stamp matches the sources it runs."""
from __future__ import annotations

import hashlib
import os

ROOT = os.path.dirname(os.path.abspath(__file__))


def csrc_sha16() -> str:
    h = hashlib.sha256()
    for sub in ("csrc", "fortran"):
        d = os.path.join(ROOT, sub)
        for name in sorted(os.listdir(d)):
            path = os.path.join(d, name)
            if os.path.isfile(path):
                h.update(name.encode())
                h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_sha16())
