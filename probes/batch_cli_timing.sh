# probe: signalMachine --batch (the front door) on N copies of a bundled read, self-reference, with the per-stage breakdown
# (SA_CLI_TIMING=1).  $1 = reads (default 10000), $2 = npRead (default: the 492-event R9 read; r9p4_oneD.npRead = 10.9k events),
# $3 = model.  Outputs go to a scratch directory under /dev/shm (removed afterwards) so that the file system is not what is timed.
set -e
N=${1:-10000}
NP=${2:-c2925_ecoli_ch34_read1023.npRead}
MODEL=${3:-testModelR9_5mer_acgt_template.model}
W=$(mktemp -d -p /dev/shm)
trap 'rm -rf "$W"' EXIT
python3 - "$W" "$N" "$NP" <<'PY'
import sys, os
w, n, npn = sys.argv[1], int(sys.argv[2]), sys.argv[3]
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
np_path = os.path.join(root, "tests/golden/npReads", npn)
lines = open(np_path).read().split("\n")
read = lines[2].strip()
n_events = int(lines[0].split()[1])
L = len(read) - 20
contig = "ACGT" * 10 + read + "TTTT"
with open(os.path.join(w, "ref.fa"), "w") as f:
    f.write(">chrT\n")
    for i in range(0, len(contig), 60):
        f.write(contig[i:i + 60] + "\n")
with open(os.path.join(w, "ref.fa.fai"), "w") as f:
    f.write("chrT\t%d\t6\t60\t61\n" % len(contig))
with open(os.path.join(w, "g.cigar"), "w") as f:
    f.write("cigar: r 5 %d + chrT 45 %d + 1 M %d\n" % (5 + L, 45 + L, L))
with open(os.path.join(w, "manifest.tsv"), "w") as f:
    for i in range(n):
        f.write("\t".join(["read%d" % i, np_path, os.path.join(w, "g.cigar"), os.path.join(w, "out%d.tsv" % i)]) + "\n")
open(os.path.join(w, "n_events"), "w").write(str(n_events))
PY
M=$GRAFT_REPO_ROOT/tests/golden/models/$MODEL
BIN=$GRAFT_REPO_ROOT/signalalign_amd/bin/signalMachine
T0=$(date +%s.%N)
SA_CLI_TIMING=1 $BIN --batch $W/manifest.tsv -T $M -f $W/ref.fa -g 100 $SA_CLI_EXTRA > $W/stdout.txt 2> $W/stderr.txt
T1=$(date +%s.%N)
tail -2 $W/stderr.txt; if [ -n "$SA_CLI_KEEP_STDERR" ]; then cp $W/stderr.txt $SA_CLI_KEEP_STDERR; fi
BYTES=$(find $W -name 'out*.tsv' -print0 | xargs -0 cat | wc -c)
python3 - <<PY
n, ev, dt, b = $N, int(open("$W/n_events").read()), $T1 - $T0, $BYTES
print("front door: %d reads x %d events in %.2f s wall (process start to exit) = %.3g events/s, %.1f reads/s; %.2f GB of TSV = %.2f GB/s"
      % (n, ev, dt, n * ev / dt, n / dt, b / 1e9, b / 1e9 / dt))
PY
# the text-I/O bound of the same rows: how fast this host re-reads and re-writes them with no formatting at all
T2=$(date +%s.%N); find $W -name 'out*.tsv' -print0 | xargs -0 cat > $W/all.tsv; T3=$(date +%s.%N)
python3 -c "print('plain copy of the same bytes (cat): %.2f s' % ($T3 - $T2))"
