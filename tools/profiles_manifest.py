#!/usr/bin/env python3
"""Write profiles/MANIFEST.json: WHICH committed profile files bench.py's JSON line cites, and which model / batch they were measured on.

    python tools/profiles_manifest.py r6end [--model deit_tiny_patch16_224 --batch 512]

Run once when a round's profile run (tools/round_profiles.sh <tag>) is copied into profiles/<tag>_*.  bench.py reads the manifest instead of guessing
"the newest" file from the names (`sorted(glob)` put r5zz behind r5end) and refuses to attach a PMC number measured on another model."""
import argparse
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KINDS = {"pmc_traffic": "{tag}_pmc_traffic.json", "pmc_mfma": "{tag}_pmc_mfma.json", "steps_only": "{tag}_kernel_stats_uvc_train_steps_only.csv"}


def main():
    p = argparse.ArgumentParser()
    p.add_argument("tag")
    p.add_argument("--model", default="deit_tiny_patch16_224")
    p.add_argument("--batch", type=int, default=512)
    a = p.parse_args()
    path = os.path.join(ROOT, "profiles", "MANIFEST.json")
    try:
        man = json.load(open(path))
    except Exception:
        man = {}
    for kind, pat in KINDS.items():
        f = pat.format(tag=a.tag)
        if not os.path.exists(os.path.join(ROOT, "profiles", f)):
            print(f"profiles/{f} missing: '{kind}' keeps {man.get(kind, {}).get('file')}")
            continue
        commit = None
        if f.endswith(".json"):
            try:
                commit = json.load(open(os.path.join(ROOT, "profiles", f))).get("commit")
            except Exception:
                pass
        man[kind] = {"file": f, "tag": a.tag, "model": a.model, "batch": a.batch, "commit": commit}
    json.dump(man, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(man, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
