#!/usr/bin/env python3
"""Generate tests/golden/digests_<size>.json: SHA-256 of the bits of every output of the CPU oracle (oracle/*.c) for the
scenarios of tests/digest_scenarios.py at full size (benchmark 360x180x75, a 1440x64x75 band of the OM4 grid, a
48x1080x3 strip).  Run in the build container (minutes of CPU); the GPU tests (tests/test_golden_digests.py) compare the
library's outputs with these digests, so every multi-block launch path has a bitwise check without the oracle having
to run at that size on the GPU box.

    python tools/make_golden_digests.py [size ...]
    python tools/make_golden_digests.py --part model_rk2b [size ...]     # only this part, merged into the existing files
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import digest_scenarios as ds  # noqa: E402


def main():
    args = sys.argv[1:]
    part = None
    if "--part" in args:
        k = args.index("--part"); part = args[k + 1]; del args[k:k + 2]
    sizes = args or list(ds.SIZES)
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    for size in sizes:
        t0 = time.time()
        prog = lambda n: print(f"  [{time.time() - t0:6.1f} s] {n}", flush=True)
        path = os.path.join(ROOT, "tests", "golden", f"digests_{size}.json")
        if part:
            new = ds.run(ds.OracleOps, size, parts=(part,), progress=prog)
            old = json.load(open(path))
            assert all(old["fields"][k] == v for k, v in new.items() if k.startswith("input.")), "the inputs changed"
            old["fields"].update(new)
            old["git_head"] = f"{old['git_head']} + {part} at {head}"
            with open(path, "w") as f:
                json.dump(old, f, indent=0, separators=(",", ":"))
            print(f"{path}: +{len(new)} fields, {time.time() - t0:.0f} s")
            continue
        out = ds.run(ds.OracleOps, size, progress=prog)
        with open(path, "w") as f:
            json.dump({"size": size, "shape": list(ds.SIZES[size]), "generator": "tools/make_golden_digests.py (oracle/*.c, gcc -O2 "
                       "-ffp-contract=off)", "git_head": head, "fields": out}, f, indent=0, separators=(",", ":"))
        print(f"{path}: {len(out)} fields, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
