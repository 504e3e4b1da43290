#!/bin/sh
# Builds the lastz that signalAlign vendors (externalTools/lastz-distrib-1.03.54/src, plain C, integer scoring) from the
# sources where they lie under /root/reference into oracle/_ref/cPecanLastz (git-ignored).  lastz is NOT part of the
# path and not the oracle: it only generates the anchor cigars the reference's whole-read tests feed to
# getAlignedPairsUsingAnchors (impl/pairwiseAligner.c:1660-1740).  Only its output is committed
# (tests/golden/cigars/, written by tests/golden/make_lastz_cigars.py).
set -e
SRC=${SA_REFERENCE:-/root/reference}/externalTools/lastz-distrib-1.03.54/src
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
mkdir -p "$OUT/lastz_obj"
FILES="lastz infer_scores seeds pos_table quantum seed_search diag_hash chain gapped_extend tweener masking segment
 edit_script identity_dist coverage_dist continuity_dist output gfa lav axt maf cigar sam genpaf text_align align_diffs
 utilities dna_utilities sequences capsule"
OBJS=""
for f in $FILES; do
  gcc -c -O3 -w -D_FILE_OFFSET_BITS=64 -D_LARGEFILE_SOURCE -DVERSION_MAJOR='"1"' -DVERSION_MINOR='"03"' \
      -DVERSION_SUBMINOR='"54"' -DREVISION_DATE='"20140128"' -DSUBVERSION_REV='"1827:1830"' "-Dscore_type='I'" \
      "$SRC/$f.c" -o "$OUT/lastz_obj/$f.o"
  OBJS="$OBJS $OUT/lastz_obj/$f.o"
done
gcc $OBJS -lm -o "$OUT/cPecanLastz"
echo "built $OUT/cPecanLastz"
