"""MIOpen performance database for the networks' convolutions on gfx950.

The ROCm 7.2 image ships MIOpen find databases for gfx90a / gfx942 but none for gfx950, so every convolution of the
encoders / decoders is chosen by MIOpen's fallback heuristic (fp32 Winograd where it applies, NHWC implicit-GEMM
weight gradients wrapped in layout transposes).  `miopen_db/` holds the user find-db / perf-db that
`tools/miopen_tune.sh` recorded on an MI355X for the shapes of BASELINE.json's configurations (text, committed) and
the compiled kernels of the winning solvers (`cache/*.ukdb`, a build product: git-ignored like the `.so`).  With the
database in place MIOpen's immediate mode (torch.backends.cudnn.benchmark = False, the reference's setting,
train.py:21) picks the measured-fastest solver for a known shape and falls back to its heuristic for any other.

`use_shipped_db()` only sets the two MIOpen environment variables, and only when the caller has not set them.
"""
import os

DB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")


def use_shipped_db():
    """Point MIOpen at the in-tree database (before the process runs its first convolution).  Returns the
    directory in use, or None when BBD_MIOPEN_DB=0 or no database is shipped."""
    if os.environ.get("BBD_MIOPEN_DB", "1") == "0" or not os.path.isdir(DB_DIR):
        return None
    if not any(f.endswith(".ufdb.txt") for f in os.listdir(DB_DIR)):
        return None
    if "MIOPEN_USER_DB_PATH" in os.environ:
        return os.environ["MIOPEN_USER_DB_PATH"]
    db = DB_DIR
    if not os.access(DB_DIR, os.W_OK):
        # MIOpen keeps lock files next to its databases and appends what it learns: a read-only checkout gets a
        # private writable copy (a few MB) instead
        import shutil
        import tempfile
        db = os.path.join(tempfile.mkdtemp(prefix="bbd_miopen_"), "miopen_db")
        shutil.copytree(DB_DIR, db)
    os.makedirs(os.path.join(db, "cache"), exist_ok=True)
    os.environ["MIOPEN_USER_DB_PATH"] = db
    os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(db, "cache"))
    return db
