"""copies gpurun_out/lines_r04/*.json (probes/bench_lines_r04.sh) into profiles/bench_r04_<name>.json"""
import os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "lines_r04")
for f in sorted(os.listdir(src)):
    if f.endswith(".json") and os.path.getsize(os.path.join(src, f)) > 0:
        shutil.copy(os.path.join(src, f), os.path.join(ROOT, "profiles", "bench_r04_" + f))
        print("profiles/bench_r04_" + f)
