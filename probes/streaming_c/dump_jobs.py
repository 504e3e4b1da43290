"""Writes two sets of 2000 synthetic 5000-event jobs (the headline workload's generator) to jobs_<i>.bin for stream.c."""
import os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sa_cases as cases
out = sys.argv[1]
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
for s in range(2):
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, n_reads, 5000, first_index=100000 * s)
    with open(os.path.join(out, "jobs_%d.bin" % s), "wb") as f:
        f.write(struct.pack("<q", len(jobs)))
        for j in jobs:
            ref = j["ref"].encode()
            ev = np.ascontiguousarray(j["events"], dtype=np.float64)
            ax = np.ascontiguousarray(j["ax"], dtype=np.int64)
            ay = np.ascontiguousarray(j["ay"], dtype=np.int64)
            f.write(struct.pack("<qqq", len(ref), ev.shape[0], len(ax)))
            f.write(ref); f.write(ev.tobytes()); f.write(ax.tobytes()); f.write(ay.tobytes())
            f.write(struct.pack("<ddd", j["scale"], j["shift"], j["var"]))
print(cases.MODEL_6MER)
