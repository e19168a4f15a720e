"""Command-line options with the reference's flag names and defaults (options.py:11-258) for the part
of the system this build covers.  Flags that select other model zoos / datasets of the reference are
accepted so existing command lines still parse, and rejected with a clear message when set."""
import argparse
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

# (flag, kwargs)
_FLAGS = [
    # paths
    ("--data_path", dict(type=str, default="data/KITTI_RAW")),
    ("--kt_path", dict(type=str, default=os.path.join(os.path.dirname(_HERE), "kitti_data"))),
    ("--log_dir", dict(type=str, default=os.path.join(os.path.dirname(_HERE), "paper"))),
    ("--training_file", dict(type=str, default="train_files_baselines")),
    ("--splits_dir", dict(type=str, default=os.path.join(os.path.dirname(_HERE), "splits"))),
    # BaseBoostDepth switches
    ("--rand", dict(action="store_true")), ("--trimin", dict(action="store_true")),
    ("--decomp", dict(action="store_true")), ("--partial_skip", dict(action="store_true")),
    ("--incremental_skip", dict(action="store_true")), ("--pose_error", dict(type=float, default=1)),
    ("--naive_mix", dict(action="store_true")), ("--kt", dict(action="store_true")),
    # model / loss
    ("--model_name", dict(type=str, default="mdp")), ("--num_layers", dict(type=int, default=18, choices=[18, 34, 50, 101, 152])),
    ("--height", dict(type=int, default=192)), ("--width", dict(type=int, default=640)),
    ("--disparity_smoothness", dict(type=float, default=1e-3)), ("--scales", dict(nargs="+", type=int, default=[0, 1, 2, 3])),
    ("--min_depth", dict(type=float, default=0.1)), ("--max_depth", dict(type=float, default=100.0)),
    ("--frame_ids", dict(nargs="+", type=int, default=[0, -1, 1])), ("--no_ssim", dict(action="store_true")),
    ("--weights_init", dict(type=str, default="pretrained", choices=["pretrained", "scratch"])),
    # optimisation
    ("--batch_size", dict(type=int, default=12)), ("--learning_rate", dict(type=float, default=1e-4)),
    ("--num_epochs", dict(type=int, default=20)), ("--pytorch_random_seed", dict(type=int, default=42)),
    # system
    ("--cuda", dict(type=int, default=0)), ("--no_cuda", dict(action="store_true")),
    ("--num_workers", dict(type=int, default=12)),
    # loading / logging
    ("--load_weights_folder", dict(type=str, default="None")),
    ("--models_to_load", dict(nargs="+", type=str, default=["encoder", "depth", "pose_encoder", "pose"])),
    ("--log_frequency", dict(type=int, default=250)), ("--save_frequency", dict(type=int, default=1)),
    # evaluation
    ("--eval_stereo", dict(action="store_true")), ("--eval_mono", dict(action="store_true")),
    ("--disable_median_scaling", dict(action="store_true")), ("--pred_depth_scale_factor", dict(type=float, default=1)),
    ("--eval_split", dict(type=str, default="eigen")), ("--save_pred_disps", dict(action="store_true")),
    ("--post_process", dict(action="store_true")),
    # this build
    # MonoViT (BASELINE configs[4]): MPViT-small encoder + HR decoder, AdamW with two LR groups
    ("--ViT", dict(action="store_true")), ("--mpvit_checkpoint", dict(type=str, default="./ckpt/mpvit_small.pth")),
    ("--materialize_warps", dict(action="store_true")), ("--synthetic", dict(action="store_true")),
    ("--step_graph", dict(action="store_true")), ("--loader_workers", dict(type=str, default="process", choices=["process", "thread"])),
    # the step's pose-network calls as one batched pass with per-call BatchNorm statistics (default) or one by one
    ("--separate_pose_calls", dict(dest="batched_pose", action="store_false")),
    # --rand recipes run the step in pooled form (pooled.py: one frame pool, static step tables, one step graph per pose-row
    # bucket, captured by Trainer.prewarm() at the start of every epoch) and replay step graphs by default
    ("--per_signature_step", dict(dest="pooled_step", action="store_false", default=None)),
    ("--no_prewarm", dict(dest="prewarm", action="store_false")),
    ("--no_step_graph", dict(action="store_true")),
    # early curriculum: 0 = pad the pose pass to the next measured row count (32 | 48 for batch 12); "max" or a number = ONE
    # row count for the whole phase (one graph; the pass then always runs the phase's largest size)
    ("--early_pose_rows", dict(type=str, default="0")),
]
# other zoos / datasets of the reference: parsed, refused when set
_OUT_OF_SCOPE = ["--SYNS_eval", "--SQL", "--SQL_L", "--CA_depth", "--DIFFNet", "--chamfer", "--stereo_guide",
                 "--x_min", "--png", "--use_stereo", "--eval_eigen_to_benchmark"]


class MonodepthOptions:
    def __init__(self):
        self.parser = argparse.ArgumentParser(description="BaseBoostDepth options (MI355X build)")
        for flag, kw in _FLAGS:
            self.parser.add_argument(flag, **kw)
        for flag in _OUT_OF_SCOPE:
            self.parser.add_argument(flag, action="store_true", help="not part of this build")
        self.parser.add_argument("--debug", action="store_true")
        self.parser.add_argument("--syns_path", type=str, default="data/KITTI_RAW")
        self.parser.add_argument("--x_val", type=int, default=3)

    def parse(self, argv=None):
        self.options = self.parser.parse_args(argv)
        bad = [f for f in _OUT_OF_SCOPE if getattr(self.options, f[2:])]
        if bad:
            self.parser.error("%s select parts of the reference that are outside this build's scope" % ", ".join(bad))
        if self.options.rand and not self.options.no_step_graph:
            self.options.step_graph = True
        if self.options.early_pose_rows != "max":
            self.options.early_pose_rows = int(self.options.early_pose_rows)
        return self.options
