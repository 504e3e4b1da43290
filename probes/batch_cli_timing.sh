# probe: wall time of signalMachine --batch on N copies of the bundled R9.4 read (10.9k events), self-reference
set -e
N=${1:-200}
W=$(mktemp -d)
python3 - "$W" "$N" <<'PY'
import sys, os
w, n = sys.argv[1], int(sys.argv[2])
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
np_path = os.path.join(root, "tests/golden/npReads/r9p4_oneD.npRead")
read = open(np_path).read().split("\n")[2].strip()
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
PY
M=$GRAFT_REPO_ROOT/tests/golden/models/testModelR9.4_450bps.nucleotide.6mer.template.model
BIN=$GRAFT_REPO_ROOT/signalalign_amd/bin/signalMachine
TIMEFORMAT="batch of $N reads: %R s wall, %U s user"; time $BIN --batch $W/manifest.tsv -T $M -f $W/ref.fa -g 100 > $W/stdout.txt 2> $W/stderr.txt
tail -1 $W/stderr.txt
TIMEFORMAT="single read: %R s wall"; time $BIN -T $M -q $GRAFT_REPO_ROOT/tests/golden/npReads/r9p4_oneD.npRead -p $W/g.cigar -f $W/ref.fa -n chrT -u $W/single.tsv -L s -g 100 > /dev/null 2> $W/e1.txt
tail -1 $W/e1.txt
wc -l $W/out0.tsv $W/single.tsv | head -2
rm -rf $W
