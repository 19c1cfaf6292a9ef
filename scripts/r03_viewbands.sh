# (view, band) segments: all eight ranks of C5 cut eight ways + the four of four ways (kernel sums per rank); then bands of all views
set -e
SPECS="0/8 1/8 2/8 3/8 4/8 5/8 6/8 7/8 0/4 1/4 2/4 3/4" bash scripts/prof_shard.sh vb2/c5 C5 view_bands
SPECS="0/8 3/8 7/8 0/4 2/4" bash scripts/prof_shard.sh vb2/c5b C5 bands
