"""MIOpen performance database for the networks' convolutions on gfx950.

The ROCm 7.2 image ships MIOpen find databases for gfx90a / gfx942 but none for gfx950, so every convolution of the
encoders / decoders is chosen by MIOpen's fallback heuristic (fp32 Winograd where it applies, NHWC implicit-GEMM
weight gradients wrapped in layout transposes).  `miopen_db/` holds the user find-db / perf-db that
`tools/miopen_tune.sh` recorded on an MI355X for the shapes of BASELINE.json's configurations (text, committed) and
the compiled kernels of the winning solvers (`cache/*.ukdb`, a build product: git-ignored like the `.so`).  With the
database in place MIOpen's immediate mode (torch.backends.cudnn.benchmark = False, the reference's setting,
train.py:21) picks the measured-fastest solver for a known shape and falls back to its heuristic for any other.

`use_shipped_db()` is an EXPLICIT call (Trainer.__init__, bench.py, train.py make it before the first convolution;
importing the package changes nothing).  It never writes into the checkout: MIOpen appends to its user database and
keeps lock files beside it, so the shipped files are copied ONCE into a per-user cache directory (reused by later
processes and by all ranks of a job; MIOpen's own locking serialises their writes) and MIOpen is pointed there.  A
caller's own MIOPEN_USER_DB_PATH always wins.
"""
import hashlib
import os
import shutil
import tempfile

DB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")


def _cache_root():
    root = os.environ.get("BBD_MIOPEN_CACHE")
    if root:
        return root
    home = os.path.expanduser("~")
    if home and home != "~" and os.access(home, os.W_OK):
        return os.path.join(home, ".cache", "baseboostdepth_amd")
    return os.path.join(tempfile.gettempdir(), "baseboostdepth_amd_%d" % os.getuid())


def _fingerprint():
    h = hashlib.sha256()
    for base, _, files in sorted(os.walk(DB_DIR)):
        for f in sorted(files):
            if f.endswith((".lock", ".time")):
                continue
            st = os.stat(os.path.join(base, f))
            h.update(("%s:%d:%d;" % (os.path.relpath(os.path.join(base, f), DB_DIR), st.st_size, int(st.st_mtime))).encode())
    return h.hexdigest()[:12]


def use_shipped_db():
    """Point MIOpen at a private copy of the in-tree database (call before the process runs its first convolution).
    Returns the directory in use, or None when BBD_MIOPEN_DB=0 or no database is shipped."""
    if os.environ.get("BBD_MIOPEN_DB", "1") == "0" or not os.path.isdir(DB_DIR):
        return None
    if not any(f.endswith(".ufdb.txt") for f in os.listdir(DB_DIR)):
        return None
    if "MIOPEN_USER_DB_PATH" in os.environ:
        return os.environ["MIOPEN_USER_DB_PATH"]
    db = os.path.join(_cache_root(), "miopen_db_" + _fingerprint())
    if not os.path.isdir(db):
        os.makedirs(os.path.dirname(db), exist_ok=True)
        tmp = tempfile.mkdtemp(prefix="miopen_db_", dir=os.path.dirname(db))
        shutil.copytree(DB_DIR, os.path.join(tmp, "db"), ignore=shutil.ignore_patterns("*.lock", "*.time"))
        try:
            os.rename(os.path.join(tmp, "db"), db)         # atomic: concurrent ranks race here, one wins
        except OSError:
            pass
        shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(os.path.join(db, "cache"), exist_ok=True)
    os.environ["MIOPEN_USER_DB_PATH"] = db
    os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(db, "cache"))
    return db
