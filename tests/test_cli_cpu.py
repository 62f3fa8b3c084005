"""CPU: the CLI keeps the reference's flag names and defaults (UVC/joint_train.py:684-879; fixture
tests/golden/cli_flags.json extracted from the reference by make_cli_golden.py)."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def test_cli_flags_match_reference():
    from uvc_amd.cli import build_parser
    ref = json.load(open(os.path.join(HERE, "golden", "cli_flags.json")))
    parser = build_parser()
    mine = {}
    for act in parser._actions:
        for opt in act.option_strings:
            if opt.startswith("--"):
                mine[opt] = act
    missing = [f for f in ref if f not in mine]
    assert not missing, f"flags of the reference missing from the CLI: {missing}"
    for flag, ent in ref.items():
        act = mine[flag]
        if "default" in ent and ent["default"] != "<expr>" and flag not in ("--local_rank", "--model_type"):
            assert act.default == ent["default"], (flag, act.default, ent["default"])
        if "choices" in ent and ent["choices"] != "<expr>":
            assert list(act.choices) == list(ent["choices"]), flag


def test_default_args_are_the_readme_command():
    """run_uvc_train.sh:4-38 (values that differ from the argparse defaults)."""
    from uvc_amd.stage1 import default_args
    a = default_args()
    assert (a.model_type, a.budget, a.distillation_type, a.distillation_alpha) == ("deit_tiny_patch16_224", 0.5, "soft", 0.1)
    assert (a.glr, a.gating_weight, a.gating_interval, a.z_grad_clip, a.seed) == (0.1, 5e-4, 50, 0.5, 730)
    assert (a.zlr_schedule_list, a.num_epochs, a.warmup_epochs, a.eps, a.eps_decay) == ("1,5,9,13,17", 30, 5, 0.1, 0.92)
