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
    assert again[1] == db and len(os.listdir(str(tmp_path))) == 1       # reused, not re-copied


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
