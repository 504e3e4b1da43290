set -e
W=$(mktemp -d)
python3 - "$W" <<'PY'
import sys, os
w = sys.argv[1]
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
    for i in range(12):
        f.write("\t".join(["read%d" % i, np_path, os.path.join(w, "g.cigar"), os.path.join(w, "out%d.tsv" % i)]) + "\n")
PY
cd $GRAFT_REPO_ROOT
python -m signalalign_amd.batch_runner --gpus 1 $W/manifest.tsv -- -T tests/golden/models/testModelR9.4_450bps.nucleotide.6mer.template.model -f $W/ref.fa -g 100 > $W/so.txt 2> $W/se.txt; echo rc=$?
tail -1 $W/se.txt; wc -l $W/so.txt $W/out11.tsv | head -2
rm -rf $W
