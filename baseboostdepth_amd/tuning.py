"""MIOpen performance database for the networks' convolutions on gfx950.

The ROCm 7.2 image ships MIOpen find databases for gfx90a / gfx942 but none for gfx950, so every convolution of the
encoders / decoders is chosen by MIOpen's fallback heuristic (fp32 Winograd where it applies, NHWC implicit-GEMM
weight gradients wrapped in layout transposes).  `miopen_db/` holds the user find-db / perf-db that
`tools/miopen_tune.sh` recorded on an MI355X for the shapes of BASELINE.json's configurations (text, committed) and
the compiled kernels of the winning solvers (`cache/*.ukdb`, a build product: git-ignored like the `.so`).  With the
database in place MIOpen's immediate mode (torch.backends.cudnn.benchmark = False, the reference's setting,
train.py:21) picks the measured-fastest solver for a known shape and falls back to its heuristic for any other.

`use_shipped_db()` is an EXPLICIT call (Trainer.__init__, bench.py and evaluation.evaluate make it before the first
convolution; importing the package changes nothing).  It never writes into the checkout: MIOpen appends to its user database and
keeps lock files beside it, so the shipped files are copied ONCE into a per-user cache directory (reused by later
processes and by all ranks of a job; MIOpen's own locking serialises their writes) and MIOpen is pointed there.  A
caller's own MIOPEN_USER_DB_PATH always wins.
"""
import hashlib
import os
import shutil
import tempfile

DB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")
# Row counts of the batched pose pass for which `miopen_db/` holds find results (tools/miopen_tune_pose.sh: 9-12 minutes
# of find mode each).  `Trainer._pose_pairs` rounds the pass up to the next of these (beyond them: to a multiple of 32), so
# the pose network's convolutions only ever meet problems MIOpen has measured solvers for.  The epoch-15 draws of the
# boosted recipe need 192-288 rows (24 + 4 * sum(m - 1) per batch of 12), epochs 10-12 about 100-160, the early curriculum 24-48.
# Round 6 added 24-44 in steps of 4 (the early curriculum's pass has 2 * sum(m) rows), 104-176 in steps of 8 (epochs 10-13) and 200 /
# 216 / 232 / 248 / 264 / 280: where a phase's draws are dense the buckets are 8 rows apart (the padding is the one cost
# the pooled step graphs still pay against a frozen batch: 3 % of a boosted step at 16-row buckets).
POSE_ROW_COUNTS = (24, 28, 32, 36, 40, 44, 48, 64, 96, 104, 112, 120, 128, 136, 144, 152, 160, 168, 176, 192, 200, 208, 216, 224, 232, 240,
                   248, 256, 264, 272, 280, 288, 320)


def padded_pose_rows(n, quantum=32):
    """Rows the batched pose pass of `n` real rows runs with: the next measured row count if it is no further away than the
    next multiple of `quantum`, else that multiple (a row count MIOpen meets for the first time costs tens of seconds of solver
    compilation ONCE per machine - its user database keeps the result)."""
    q = -(-n // quantum) * quantum
    for r in POSE_ROW_COUNTS:
        if n <= r <= q:
            return r
    return q


def _cache_root():
    root = os.environ.get("BBD_MIOPEN_CACHE")
    if root:
        return root
    home = os.path.expanduser("~")
    if home and home != "~" and os.access(home, os.W_OK):
        return os.path.join(home, ".cache", "baseboostdepth_amd")
    return os.path.join(tempfile.gettempdir(), "baseboostdepth_amd_%d" % os.getuid())


def _fingerprint():
    """Content hash of the shipped database (names + bytes, not mtimes: every clone of one commit maps to the SAME private
    copy instead of leaving a new one behind)."""
    h = hashlib.sha256()
    for base, _, files in sorted(os.walk(DB_DIR)):
        for f in sorted(files):
            if f.endswith((".lock", ".time")):
                continue
            h.update((os.path.relpath(os.path.join(base, f), DB_DIR) + ";").encode())
            with open(os.path.join(base, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:12]


# what the last use_shipped_db() / use_shipped_gemm_db() did, for callers that report it (bench.py): the private MIOpen
# copy is one MIOpen appends to, so a run that found it already there also sees earlier runs' find results
STATUS = {"miopen_db": None, "miopen_db_prewarmed": None, "miopen_db_accepted": None, "miopen_db_why": None,
          "gemm_db": None, "gemm_db_accepted": None, "gemm_db_why": None}


def shipped_db_tag():
    """Version tag in the shipped find database's file name, `<arch><CUs>.HIP.<tag>.ufdb.txt` (MIOpen only opens the file
    whose tag is its OWN `major_minor_patch_tweak`: a database recorded with another build is ignored without a word)."""
    for f in sorted(os.listdir(DB_DIR)) if os.path.isdir(DB_DIR) else []:
        if f.endswith(".ufdb.txt") and ".HIP." in f:
            return f.split(".HIP.", 1)[1][:-len(".ufdb.txt")]
    return None


def running_miopen_tag():
    """`major_minor_patch_tweak` of the MIOpen this process will use: PyTorch-ROCm bundles its own libMIOpen.so, whose
    version banner (" MIOpen version 3.5.0.<tweak>") is read from the library file (one scan of ~0.3 s, remembered in the
    per-user cache directory by the file's size and mtime).  Falls back to `torch.backends.cudnn.version()`'s
    major_minor_patch when the banner cannot be found; None when neither is available."""
    import mmap
    import re
    import torch
    lib = os.path.join(os.path.dirname(torch.__file__), "lib", "libMIOpen.so")
    if os.path.isfile(lib):
        st = os.stat(lib)
        memo = os.path.join(_cache_root(), "miopen_tag_%d_%d.txt" % (st.st_size, int(st.st_mtime)))
        try:
            with open(memo) as fh:
                tag = fh.read().strip()
            if tag:
                return tag
        except OSError:
            pass
        try:
            with open(lib, "rb") as fh:
                mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
                i = mm.find(b" MIOpen version ")
                tag = None
                if i >= 0:
                    m = re.match(rb" MIOpen version (\d+)\.(\d+)\.(\d+)\.([0-9A-Za-z\-]+)", mm[i:i + 96])
                    if m:
                        tag = "_".join(g.decode() for g in m.groups())
                mm.close()
            if tag:
                try:
                    os.makedirs(os.path.dirname(memo), exist_ok=True)
                    with open(memo, "w") as fh:
                        fh.write(tag)
                except OSError:
                    pass
                return tag
        except (OSError, ValueError):
            pass
    try:
        v = torch.backends.cudnn.version()
    except Exception:
        v = None
    if v:
        return "%d_%d_%d" % (v // 1000000, (v // 1000) % 1000, v % 1000)
    return None


def miopen_db_accepted():
    """(True, None) when the shipped find database carries the running MIOpen's version tag - the condition under which
    MIOpen reads it; otherwise (False, why).  Like `gemm_db_accepted` for the TunableOp table."""
    shipped, running = shipped_db_tag(), running_miopen_tag()
    if shipped is None:
        return False, "no find database shipped"
    if running is None:
        return False, "the running MIOpen's version could not be determined"
    if shipped == running or (running.count("_") == 2 and shipped.startswith(running + "_")):
        return True, None
    return False, "database recorded with MIOpen %s, this process runs MIOpen %s" % (shipped, running)


_warned_rows = set()


def note_pose_rows(rows):
    """Called with the row count of every batched pose pass: warns ONCE per row count the shipped database has no find
    results for (MIOpen compiles that problem's solvers at first use - tens of seconds on the training thread, once per
    machine; `POSE_ROW_COUNTS` are measured for batch size 12, the reference's)."""
    if rows in POSE_ROW_COUNTS or rows in _warned_rows:
        return False
    _warned_rows.add(rows)
    import warnings
    warnings.warn("batched pose pass with %d rows: the shipped MIOpen database has find results for %s rows only (batch size "
                  "12); MIOpen compiles this row count's solvers at first use - expect a stall of tens of seconds, once per "
                  "machine (tools/miopen_tune_pose.sh records more row counts)" % (rows, list(POSE_ROW_COUNTS)))
    return True


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
    if STATUS["miopen_db"] != db:
        STATUS["miopen_db"], STATUS["miopen_db_prewarmed"] = db, os.path.isdir(db)
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
    if STATUS["miopen_db_accepted"] is None:
        # MIOpen drops a database of another build without a word: say so (once per process)
        ok, why = miopen_db_accepted()
        STATUS.update(miopen_db_accepted=ok, miopen_db_why=why)
        if not ok:
            import warnings
            warnings.warn("the shipped MIOpen find database will be ignored (%s): convolutions take MIOpen's heuristic solvers "
                          "(slower, and every new pose-pass row count compiles solvers for tens of seconds); re-record it with "
                          "tools/miopen_tune.sh" % why)
    os.environ["MIOPEN_USER_DB_PATH"] = db
    os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(db, "cache"))
    return db


# ------------------------------------------------------------------------------------------------ GEMM solutions
GEMM_DB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_db", "tunableop_gfx950.csv")


def use_shipped_gemm_db():
    """Point PyTorch's TunableOp at the in-tree table of measured-fastest hipBLASLt / rocBLAS solutions for the token
    GEMMs of MonoViT (BASELINE configs[4]: qkv / proj / MLP Linear layers forward, data and weight gradients; recorded
    on an MI355X by `tools/gemm_tune.sh`).  hipBLASLt's own heuristic picks poorly for these tall-skinny fp32 shapes
    ([92 160 x 64] x [64 x 256] ...): the MonoViT step runs 161.8 -> 180.2 images/s with the table
    (profiles/r03/bench_vit_gemm_db_ab.txt).  Tuning stays OFF at run time: a known shape takes its recorded solution, an
    unknown one the library default - nothing is measured or written during training, so the step graph capture is
    unaffected.  The table is validated by TunableOp against the PyTorch / HIP / hipBLASLt / rocBLAS versions and the GPU
    architecture in its header and ignored if they differ.  Explicit call (Trainer.__init__, bench.py), a private copy
    like the MIOpen database; BBD_GEMM_DB=0 or a caller's own PYTORCH_TUNABLEOP_ENABLED wins.  Returns the file in use."""
    if os.environ.get("BBD_GEMM_DB", "1") == "0" or not os.path.isfile(GEMM_DB):
        return None
    if "PYTORCH_TUNABLEOP_ENABLED" in os.environ:
        return os.environ.get("PYTORCH_TUNABLEOP_FILENAME")
    import torch
    if not torch.cuda.is_available():
        return None
    import torch.cuda.tunable as tunable
    with open(GEMM_DB, "rb") as fh:
        tag = hashlib.sha256(fh.read()).hexdigest()[:12]
    dst = os.path.join(_cache_root(), "gemm_db_%s.csv" % tag)
    if not os.path.isfile(dst):
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        fd, tmp = tempfile.mkstemp(prefix="gemm_db_", dir=os.path.dirname(dst))
        os.close(fd)
        shutil.copyfile(GEMM_DB, tmp)
        try:
            os.rename(tmp, dst)
        except OSError:
            os.unlink(tmp)
    tunable.set_filename(dst, insert_device_ordinal=False)
    tunable.enable(True)
    tunable.tuning_enable(False)
    # TunableOp drops a table without a word when the versions in its header are not this process's: say so
    ok, why = gemm_db_accepted(dst)
    STATUS.update(gemm_db=dst, gemm_db_accepted=ok, gemm_db_why=why)
    if not ok:
        import warnings
        warnings.warn("the shipped GEMM solution table will be ignored by TunableOp (%s): MonoViT's token GEMMs take "
                      "the library defaults (about -10 %% step throughput); re-record it with tools/gemm_tune.sh" % why)
    return dst


def gemm_db_accepted(path=None):
    """(True, None) when every `Validator` line of the table equals what this process's TunableOp reports
    (torch.cuda.tunable.get_validators(): PyTorch / HIP / hipBLASLt / rocBLAS versions, GPU architecture) - the condition
    under which TunableOp uses the table; otherwise (False, which entries differ)."""
    import torch.cuda.tunable as tunable
    path = path or GEMM_DB
    want = {}
    with open(path) as fh:
        for line in fh:
            parts = line.strip().split(",")
            if len(parts) >= 3 and parts[0] == "Validator":
                want[parts[1]] = ",".join(parts[2:])
    try:
        have = {str(k): str(v) for k, v in tunable.get_validators()}
    except Exception as e:                       # never fail a run on the report
        return False, "validators unavailable (%s)" % type(e).__name__
    bad = ["%s: table %s, here %s" % (k, v, have.get(k)) for k, v in want.items() if have.get(k) != v]
    return (not bad), ("; ".join(bad) or None)
