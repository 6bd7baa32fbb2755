"""profiles/<tag>_pmc.json from the two rocpd_pmc.py listings (FETCH_SIZE / WRITE_SIZE, in KB) of tools/profile_round.sh:
    python tools/pmc_json.py TAG gpurun_out/prof_TAG profiles"""
import json, re, sys
tag, src, dst = sys.argv[1:4]
kern = {}
for c, key in (("FETCH_SIZE", "fetch_bytes"), ("WRITE_SIZE", "write_bytes")):
    for line in open(f"{src}/{tag}_pmc_{c}.txt"):
        m = re.match(r"(.{70}) (\S+)\s+dispatches\s+(\d+)\s+avg\s+([\d.]+)", line)
        if not m or m.group(2) != c:
            continue
        name = m.group(1).strip()
        if name.startswith("at::") or name.startswith("__amd"):
            continue
        e = kern.setdefault(name, {"dispatches": int(m.group(3))})
        e[key] = float(m.group(4)) * 1024.0
# the SQ pass of tools/profile_round.sh (instruction and cycle counters), when present
import os
sq = f"{src}/{tag}_pmc_sq.txt"
if os.path.exists(sq):
    for line in open(sq):
        m = re.match(r"(.{70}) (\S+)\s+dispatches\s+(\d+)\s+avg\s+([\d.]+)", line)
        if m and m.group(1).strip() in kern:
            kern[m.group(1).strip()][m.group(2)] = float(m.group(4))
import hashlib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "mom6_amd", "csrc")
out = {
    "source_sha256": {f: hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest() for f in sorted(os.listdir(csrc))
                      if f.endswith((".hip", ".hpp"))},
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python3 bench.py --steps 4 --warmup 0 "
              "--no-cpu-baseline --no-roofline; MI355X, om4_025 1440x1080x75 (tools/profile_round.sh)",
    "units": "bytes per launch, averaged over the launches of the run (the counters are in KB; x1024).  Calibration of the gfx950 "
             "FETCH_SIZE factor on this code's 8-byte-per-lane loads (round 1): cont_conv_kernel<1> reads hin + uh (1.87 GB) and "
             "reports 1.94 GB -> factor 1.0 (the x2 of the guide applies to 16-byte-per-lane loads), so no correction is applied.",
    "kernels": {k: v for k, v in kern.items() if "fetch_bytes" in v and "write_bytes" in v},
}
json.dump(out, open(f"{dst}/{tag}_pmc.json", "w"), indent=1)
print(len(out["kernels"]), "kernels")
