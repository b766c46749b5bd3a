#!/bin/bash
# Pins the "parity unpinned" seeding stage against the real reference -- for a machine that has what this image lacks:
# a Rust nightly toolchain (cargo), network access for the reference's crates (incl. the unpinned git dependency
# rust-seq2kminmers, Cargo.toml:30) and an MI355X.  Not run in the authoring container (no cargo, no network).
#
#   tools/check_against_upstream.sh <reads.fa|fq[.gz]> <reference.fa> [extra mapquik flags: -k -l -d --nohpc ...]
#
# Step 1 (localises a divergence): the reference is patched to print every reference k-min-mer it indexes
#   (src/mers.rs:29: the commented `println!("{:?}", kminmer)` becomes a tab-separated line under MQ_DUMP), the same tuples
#   are produced by this repo's CPU oracle in its frozen reading (variant 0) and in twelve combinations of the diagnostic
#   variants of the unpinned decisions (oracle/mapquik_oracle.c, mqo_set_variant; DESIGN.md section 2 says which kernel constant
#   each one would change):
#      bit 1  D3  strict `<` on the density bound      bit 8   D5  position = end of the homopolymer run (not its head)
#      bit 2  D2  the bound computed in f32            bit 16  D6  end = raw position of the window's last compressed base
#      bit 4  D2/D12  32-bit ntHash and bound          bit 32  D8  rev on `<=` (palindromic tuples)
#   The matching variant is printed; with none, the first differing tuple per variant says where to look: start/end wrong =>
#   D5-D7 (HPC positions); a missing/extra tuple => D2/D3 (bound) or D1 (ntHash); only `rev`/hash wrong => D8/D9 (orientation,
#   tuple hash).  With --nosimd as an extra flag the scalar HashMode of the crate is exercised instead of the SIMD one (D12):
#   run both -- if they match different variants, the reference's results depend on the CPU it runs on.
# Step 2: maps the reads with the reference and with this repo's native driver using the same flags and diffs the PAFs byte
#   for byte in input order (the reference's default seq_io path writes in input order, src/closures.rs:117-123).
# The resolved revision of rust-seq2kminmers is recorded so that a divergence can be tied to a crate version.
set -euo pipefail
READS=${1:?reads}; REF=${2:?reference}; shift 2
HERE=$(cd "$(dirname "$0")/.." && pwd)
WORK=${WORK:-$(mktemp -d)}
command -v cargo >/dev/null || { echo "cargo not found: this script needs a Rust toolchain (rustup install nightly)"; exit 2; }
if [ ! -d "$WORK/mapquik" ]; then git clone https://github.com/ekimb/mapquik "$WORK/mapquik"; fi
( cd "$WORK/mapquik" &&
  sed -i 's|^\(\s*\)//println!("{:?}", kminmer);|\1if std::env::var("MQ_DUMP").is_ok() { eprintln!("KMM\\t{}\\t{}\\t{}\\t{}\\t{}", kminmer.start, kminmer.end, kminmer.offset, kminmer.rev, kminmer.get_hash()); }|' src/mers.rs &&
  cargo +nightly build --release && grep -A2 'name = "rust-seq2kminmers"' Cargo.lock | tee "$WORK/seq2kminmers.rev" )
python3 -c "import sys; sys.path.insert(0, '$HERE'); from mapquik_amd import build; build.build_cli()"

# ---- step 1: reference k-min-mer tuples (single worker thread so that the dump is in file order)
MQ_DUMP=1 "$WORK/mapquik/target/release/mapquik" "$READS" --reference "$REF" -p "$WORK/dump" --threads 1 "$@" 2> "$WORK/upstream.kmm.raw" > /dev/null || true
grep '^KMM' "$WORK/upstream.kmm.raw" > "$WORK/upstream.kmm" || true
OFLAGS=()
args=("$@"); i=0
while [ $i -lt ${#args[@]} ]; do
  case "${args[$i]}" in
    -k|-l) OFLAGS+=("${args[$i]}" "${args[$((i+1))]}"); i=$((i+2));;
    -d|--density) OFLAGS+=(-d "${args[$((i+1))]}"); i=$((i+2));;
    --nohpc) OFLAGS+=(--nohpc); i=$((i+1));;
    *) i=$((i+1));;
  esac
done
best=""
for v in 32 24 16 8 7 6 5 4 3 2 1 0; do   # (variant 0 last: when several readings match this input, the frozen one is the one kept)
  python3 "$HERE/tools/dump_kminmers.py" "$REF" --variant $v "${OFLAGS[@]}" > "$WORK/oracle.v$v.kmm"
  if cmp -s "$WORK/upstream.kmm" "$WORK/oracle.v$v.kmm"; then
    echo "k-min-mer tuples: oracle variant $v IDENTICAL to the reference ($(wc -l < "$WORK/upstream.kmm") tuples)"; best=$v
  else
    echo "k-min-mer tuples: oracle variant $v differs; first difference:"
    diff "$WORK/upstream.kmm" "$WORK/oracle.v$v.kmm" | head -4 || true
  fi
done
[ "$best" = "0" ] && echo "seeding stage PINNED: the frozen reading (variant 0) reproduces the crate" || echo "seeding stage NOT pinned by variant 0 (matching variant: '${best:-none}'): run the product with --seeding-variant ${best:-?} (mq_params.flags bits 8..13; DESIGN.md section 2)"

# ---- step 2: PAF identity
"$WORK/mapquik/target/release/mapquik" "$READS" --reference "$REF" -p "$WORK/upstream" "$@"
# the HIP product runs the reading that step 1 matched (variant 0 when none did: the diff below then shows what the mismatch costs)
"$HERE/mapquik_amd/lib/mapquik" "$READS" --reference "$REF" -p "$WORK/hip" --seeding-variant "${best:-0}" "$@"
if cmp -s "$WORK/upstream.paf" "$WORK/hip.paf"; then
  echo "IDENTICAL: $(wc -l < "$WORK/hip.paf") PAF lines"
else
  echo "DIFFERENT: see $WORK/upstream.paf vs $WORK/hip.paf"; diff "$WORK/upstream.paf" "$WORK/hip.paf" | head -20
  python3 "$HERE/tools/paf_concordance.py" "$WORK/upstream.paf" "$WORK/hip.paf" || true
  exit 1
fi
