"""the command-line entry point (main.py / functions.py) drives a reference-style config end to end on the GPU:
training for a few steps (checkpoint written), resume with --load_last, evaluation and eval_time modes."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_find_last_checkpoint(tmp_path):
    sys.path.insert(0, ROOT)
    import functions
    assert functions.find_last_checkpoint(str(tmp_path)) is None
    for name in ("checkpoints_epoch_1_step_10.ckpt", "checkpoints_epoch_2_step_5.ckpt", "checkpoints_epoch_2_step_30.ckpt", "other.ckpt"):
        (tmp_path / name).write_bytes(b"")
    assert functions.find_last_checkpoint(str(tmp_path)) == "checkpoints_epoch_2_step_30.ckpt"


@pytest.mark.gpu
def test_main_training_resume_evaluation(tmp_path):
    env = dict(os.environ, AVEC_TEST_CALLBACKS=str(tmp_path), PYTHONPATH=ROOT)
    cfg = os.path.join("tests", "configs", "av_synthetic.py")
    out = _run(["-c", cfg, "-m", "training", "--steps_per_epoch", "2", "--step_log_period", "1", "--eval_steps", "1"], env)
    assert "Mode: training" in out and "epoch 1 step 1" in out
    ckpts = [f for f in os.listdir(tmp_path) if f.endswith(".ckpt")]
    assert ckpts == ["checkpoints_epoch_1_step_2.ckpt"], ckpts
    out = _run(["-c", cfg, "-m", "evaluation", "--load_last", "--eval_steps", "2"], env)
    assert "Evaluation:" in out and "loss" in out and "'wer'" in out
    out = _run(["-c", cfg, "-m", "eval_time", "--load_last", "--eval_steps", "1"], env)
    assert "Eval time:" in out
