#!/usr/bin/env python3
"""include/sfk.abi = "<SFK_ABI_VERSION> <sha256 of include/sfk.h's declarations>".

The hash covers everything a binding depends on -- struct layouts, prototypes, enum / #define values -- with comments
and whitespace removed, so that editing documentation does not move it.  tests/test_abi_cpu.py recomputes it: a layout
or prototype change without a new SFK_ABI_VERSION fails the CPU suite.  This tool writes the lock, and REFUSES to
write a different hash under a version number that is already locked (bump SFK_ABI_VERSION first).

    python tools/abi_lock.py            # check
    python tools/abi_lock.py --write    # after bumping SFK_ABI_VERSION
"""
import hashlib
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sfk.h")
LOCK = os.path.join(ROOT, "include", "sfk.abi")


def declarations(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"#define\s+SFK_ABI_VERSION\s+\d+", " ", text)     # the version itself is the other half of the lock
    return re.sub(r"\s+", " ", text).strip()


def header_state():
    src = open(HEADER).read()
    version = int(re.search(r"#define\s+SFK_ABI_VERSION\s+(\d+)", src).group(1))
    return version, hashlib.sha256(declarations(src).encode()).hexdigest()


def locked():
    if not os.path.exists(LOCK):
        return None
    v, h = open(LOCK).read().split()
    return int(v), h


def main(argv):
    version, digest = header_state()
    lock = locked()
    if "--write" in argv:
        if lock is not None and lock[0] == version and lock[1] != digest:
            print(f"include/sfk.h changed but SFK_ABI_VERSION is still {version}: bump it, then re-run", file=sys.stderr)
            return 1
        with open(LOCK, "w") as f:
            f.write(f"{version} {digest}\n")
        print(f"locked ABI {version} {digest[:16]}")
        return 0
    if lock != (version, digest):
        print(f"lock {lock} != header {(version, digest)}", file=sys.stderr)
        return 1
    print(f"ABI {version} matches include/sfk.abi")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
