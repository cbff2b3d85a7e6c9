"""Content hash of the kernel sources: profiles/ record it next to the counters they hold, bench.py refuses counters
recorded for other kernels (constants read from profiles/ go stale the moment a kernel changes)."""
import glob
import hashlib
import os


def kernel_source_hash() -> str:
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.join(here, "csrc")
    tools = os.path.join(os.path.dirname(here), "tools")
    h = hashlib.sha1()
    # the *.inc files are generated at build time and git-ignored: a fresh checkout running the prebuilt .so has none, a
    # built tree has three.  Their GENERATORS are the source: hashed instead, so both trees agree.
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.hpp"))
                    + glob.glob(os.path.join(tools, "gen_band*.py"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
