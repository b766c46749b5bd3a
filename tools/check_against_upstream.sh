#!/bin/bash
# Pins the "parity unpinned" seeding stage against the real reference -- for a machine that has what this image lacks:
# a Rust nightly toolchain (cargo), network access for the reference's crates (incl. the unpinned git dependency
# rust-seq2kminmers, Cargo.toml:30) and an MI355X.  Not run in the authoring container (no cargo, no network).
#
#   tools/check_against_upstream.sh <reads.fa|fq[.gz]> <reference.fa> [extra mapquik flags...]
#
# Builds ekimb/mapquik, maps the reads with it and with this repo's native driver using the same flags, and diffs the PAFs
# byte for byte in input order (the reference's default seq_io path writes in input order, src/closures.rs:117-123).
# It also records the resolved revision of rust-seq2kminmers so that a divergence can be tied to a crate version.
set -euo pipefail
READS=${1:?reads}; REF=${2:?reference}; shift 2
HERE=$(cd "$(dirname "$0")/.." && pwd)
WORK=${WORK:-$(mktemp -d)}
command -v cargo >/dev/null || { echo "cargo not found: this script needs a Rust toolchain (rustup install nightly)"; exit 2; }
if [ ! -d "$WORK/mapquik" ]; then git clone https://github.com/ekimb/mapquik "$WORK/mapquik"; fi
( cd "$WORK/mapquik" && cargo +nightly build --release && grep -A2 'name = "rust-seq2kminmers"' Cargo.lock | tee "$WORK/seq2kminmers.rev" )
python3 -c "import sys; sys.path.insert(0, '$HERE'); from mapquik_amd import build; build.build_cli()"
"$WORK/mapquik/target/release/mapquik" "$READS" --reference "$REF" -p "$WORK/upstream" "$@"
"$HERE/mapquik_amd/lib/mapquik" "$READS" --reference "$REF" -p "$WORK/hip" "$@"
if cmp -s "$WORK/upstream.paf" "$WORK/hip.paf"; then
  echo "IDENTICAL: $(wc -l < "$WORK/hip.paf") PAF lines"
else
  echo "DIFFERENT: see $WORK/upstream.paf vs $WORK/hip.paf"; diff "$WORK/upstream.paf" "$WORK/hip.paf" | head -20; exit 1
fi
