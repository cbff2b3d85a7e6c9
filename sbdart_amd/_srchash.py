"""Content hash of the kernel sources: profiles/ record it next to the counters they hold, bench.py refuses counters
recorded for other kernels (constants read from profiles/ go stale the moment a kernel changes)."""
import glob
import hashlib
import os


def kernel_source_hash() -> str:
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.hpp"))
                    + glob.glob(os.path.join(root, "*.inc"))):      # (generated at build time: sbd_band{1,4}_take.inc)
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
