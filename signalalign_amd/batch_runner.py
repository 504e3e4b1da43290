"""Multi-GPU front door for `signalMachine --batch`: one process per GPU, each with its share of the manifest.

Reads are independent, so the node-level run is a partition of the manifest (longest-processing-time first on the
size of the .npRead files, which is proportional to the number of events) and N processes `signalMachine --batch
<part> --device <g> ...`; no collective, nothing shared but the read-only model and reference files.  This is the
replacement for the reference's worker pool (src/signalalign/signalAlignment.py:740-848).

    python -m signalalign_amd.batch_runner --gpus 8 manifest.tsv -- -T template.model -f ref.fa -x 50 -D 0.01 -g 100
"""
import argparse
import os
import subprocess
import sys
import tempfile

from .shard import shard_indices

BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "signalMachine")


def read_manifest(path):
    """Returns the manifest's read lines (comments and blank lines dropped), verbatim."""
    with open(path) as f:
        return [ln.rstrip("\n") for ln in f if ln.strip() and not ln.startswith("#")]


def split_manifest(lines, n_parts):
    """Partition manifest lines into n_parts lists, balanced by the size of each read's .npRead file (column 2).
    Reads whose file is missing weigh 1: the aligner reports them, the runner does not hide them."""
    def cost(ln):
        f = ln.split("\t")
        try:
            return max(os.path.getsize(f[1]), 1) if len(f) > 1 else 1
        except OSError:
            return 1
    costs = [cost(ln) for ln in lines]
    return [[lines[int(i)] for i in shard_indices(costs, r, n_parts)] for r in range(n_parts)]


def run(manifest, aligner_args, n_gpus, binary=BIN, workdir=None):
    """Runs one `signalMachine --batch` per GPU; returns the list of CompletedProcess-like (returncode, stdout, stderr)."""
    lines = read_manifest(manifest)
    parts = split_manifest(lines, n_gpus)
    tmp = workdir or tempfile.mkdtemp(prefix="sa_batch_")
    procs = []
    for g, part in enumerate(parts):
        if not part:
            continue
        mp = os.path.join(tmp, "manifest.gpu%d.tsv" % g)
        with open(mp, "w") as f:
            f.write("\n".join(part) + "\n")
        procs.append(subprocess.Popen([binary, "--batch", mp, "--device", str(g)] + list(aligner_args),
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    out = []
    for p in procs:
        so, se = p.communicate()
        out.append((p.returncode, so, se))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("manifest")
    ap.add_argument("aligner_args", nargs=argparse.REMAINDER, help="options passed to signalMachine (after --)")
    a = ap.parse_args(argv)
    extra = a.aligner_args[1:] if a.aligner_args[:1] == ["--"] else a.aligner_args
    rc = 0
    for code, so, se in run(a.manifest, extra, a.gpus):
        sys.stdout.write(so)
        sys.stderr.write(se)
        rc = rc or code
    return rc


if __name__ == "__main__":
    sys.exit(main())
