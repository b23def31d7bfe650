"""Hash of the sources that decide what a training step does on the GPU (csrc/, the engine, the step): profiles record it and
bench.py compares it with the tree it runs from, so a PMC summary measured on another tree is labelled stale instead of winning
silently."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tree_hash() -> str:
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "video-classification_amd")
    files = sorted(glob.glob(os.path.join(pkg, "csrc", "*"))) + [os.path.join(pkg, f) for f in ("engine.py", "train.py", "plan.py", "arch.py")]
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(tree_hash())
