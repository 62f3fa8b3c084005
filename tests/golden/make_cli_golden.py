#!/usr/bin/env python3
"""Extract the Stage-1 argparse surface (flag names, defaults, choices) of the reference's
UVC/joint_train.py:684-879 into tests/golden/cli_flags.json.  Build-container only (needs
/root/reference).  The file cannot be imported (apex/timm), so the add_argument calls are read
from its AST; only names and literal defaults are kept."""
import ast
import json
import os

SRC = "/root/reference/UVC/joint_train.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    tree = ast.parse(open(SRC).read())
    flags = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument" and node.args:
            name = ast.literal_eval(node.args[0])
            if not name.startswith("--"):
                continue
            ent = {}
            for kw in node.keywords:
                if kw.arg in ("default", "choices", "action"):
                    try:
                        ent[kw.arg] = ast.literal_eval(kw.value)
                    except Exception:
                        ent[kw.arg] = "<expr>"
                elif kw.arg == "type":
                    ent["type"] = getattr(kw.value, "id", "<expr>")
            flags[name] = ent
    json.dump(flags, open(os.path.join(HERE, "cli_flags.json"), "w"), indent=1, sort_keys=True)
    print(len(flags), "flags")


if __name__ == "__main__":
    main()
