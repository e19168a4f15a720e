"""The shipped MIOpen performance database (baseboostdepth_amd/miopen_db, tools/miopen_tune.sh) is wired in by an explicit
call (never at import, never over a caller's own setting, never writing into the checkout) and covers every configuration
bench.py names."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = ("import os, baseboostdepth_amd; before = os.environ.get('MIOPEN_USER_DB_PATH'); "
         "from baseboostdepth_amd import tuning; tuning.use_shipped_db(); "
         "print(before, os.environ.get('MIOPEN_USER_DB_PATH'), os.environ.get('MIOPEN_CUSTOM_CACHE_DIR'))")


def _probe(**env):
    e = {k: v for k, v in os.environ.items() if not k.startswith("MIOPEN_") and k != "BBD_MIOPEN_DB"}
    e.update(env)
    out = subprocess.run([sys.executable, "-c", PROBE], cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout.strip().split()


def test_explicit_call_points_miopen_at_a_private_copy_of_the_shipped_database(tmp_path):
    before, db, cache = _probe(BBD_MIOPEN_CACHE=str(tmp_path))
    assert before == "None"                                   # importing the package does not touch the environment
    assert db.startswith(str(tmp_path)) and cache == os.path.join(db, "cache")
    shipped = os.path.join(ROOT, "baseboostdepth_amd", "miopen_db")
    assert sorted(f for f in os.listdir(db) if f.endswith(".txt")) == sorted(f for f in os.listdir(shipped) if f.endswith(".txt"))
    again = _probe(BBD_MIOPEN_CACHE=str(tmp_path))
    assert again[1] == db and len([d for d in os.listdir(str(tmp_path)) if d.startswith("miopen_db_")]) == 1       # reused, not re-copied


def test_caller_settings_win_and_the_switch_disables():
    assert _probe(MIOPEN_USER_DB_PATH="/tmp/mine")[1] == "/tmp/mine"
    assert _probe(BBD_MIOPEN_DB="0") == ["None", "None", "None"]


def test_database_holds_forward_backward_and_weight_gradient_records_for_md2_shapes():
    d = os.path.join(ROOT, "baseboostdepth_amd", "miopen_db")
    fdb = [f for f in os.listdir(d) if f.endswith(".ufdb.txt")]
    assert len(fdb) == 1 and fdb[0].startswith("gfx950")
    text = open(os.path.join(d, fdb[0])).read()
    # ResNet-18 layer1 3x3 (64 -> 64 at 48x160, batch 12) in all three directions, and the 7x7 stem
    for direction in ("F", "B", "W"):
        assert "64-48-160-3x3-64-48-160-12-1x1-1x1-1x1-0-NCHW-FP32-%s=" % direction in text, direction
    assert "-7x7-64-96-320-12-3x3-2x2-1x1-0-NCHW-FP32-F=" in text


def test_gemm_table_is_shipped_and_wired_only_on_explicit_call():
    """The TunableOp table of the MonoViT token GEMMs: committed text with its validator header; importing the package
    leaves TunableOp alone, and on a machine without a GPU the explicit call is a no-op."""
    import os
    import subprocess
    import sys
    from baseboostdepth_amd import tuning
    assert os.path.isfile(tuning.GEMM_DB)
    lines = open(tuning.GEMM_DB).read().splitlines()
    assert any(l.startswith("Validator,GCN_ARCH_NAME,gfx950") for l in lines)
    assert sum(l.startswith("Gemm") for l in lines) >= 60
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTORCH_TUNABLEOP")}
    out = subprocess.run([sys.executable, "-c",
                          "import os, torch, baseboostdepth_amd; print(int(any(k.startswith('PYTORCH_TUNABLEOP') for k in os.environ))); "
                          "from baseboostdepth_amd import tuning; print(tuning.use_shipped_gemm_db() if not torch.cuda.is_available() else 'gpu')"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.stdout.split()[:2] in (["0", "None"], ["0", "gpu"]), (out.stdout, out.stderr[-800:])


def test_a_database_of_another_miopen_build_is_reported(monkeypatch, tmp_path):
    """MIOpen opens only the find database whose file name carries its OWN version tag and ignores any other without a
    word: `use_shipped_db()` compares the shipped tag with the running library's and says so (warning + STATUS), like
    `gemm_db_accepted` does for the TunableOp table."""
    import warnings
    from baseboostdepth_amd import tuning
    assert tuning.shipped_db_tag().startswith("3_5_0_")
    ok, why = tuning.miopen_db_accepted()
    assert ok and why is None, why                      # this image's PyTorch bundles the MIOpen the database was recorded with
    monkeypatch.setattr(tuning, "running_miopen_tag", lambda: "3_6_0_20260101-1-2-gdeadbeef")
    ok, why = tuning.miopen_db_accepted()
    assert not ok and "3_6_0_20260101" in why and tuning.shipped_db_tag() in why
    monkeypatch.setenv("BBD_MIOPEN_CACHE", str(tmp_path))
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    monkeypatch.delenv("MIOPEN_CUSTOM_CACHE_DIR", raising=False)
    monkeypatch.setitem(tuning.STATUS, "miopen_db_accepted", None)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        tuning.use_shipped_db()
    assert any("will be ignored" in str(w.message) for w in caught)
    assert tuning.STATUS["miopen_db_accepted"] is False and "3_6_0" in tuning.STATUS["miopen_db_why"]
    # major_minor_patch alone (the fallback when the library's banner cannot be read) accepts any tweak of that version
    monkeypatch.setattr(tuning, "running_miopen_tag", lambda: "3_5_0")
    assert tuning.miopen_db_accepted() == (True, None)


def test_an_untuned_pose_row_count_warns_once():
    import warnings
    from baseboostdepth_amd import tuning
    tuning._warned_rows.discard(100)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        assert tuning.note_pose_rows(96) is False              # measured
        assert tuning.note_pose_rows(100) is True
        assert tuning.note_pose_rows(100) is False             # once
    assert len(caught) == 1 and "100 rows" in str(caught[0].message) and "tens of seconds" in str(caught[0].message)
    # the rounding rule only ever lands on a measured row count inside the table's range
    for n in range(1, 321):
        assert tuning.padded_pose_rows(n, 32) in tuning.POSE_ROW_COUNTS
