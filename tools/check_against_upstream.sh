#!/bin/bash
# Pins the "parity unpinned" seeding stage against the real reference -- for a machine that has what this image lacks:
# a Rust nightly toolchain (cargo), network access for the reference's crates (incl. the unpinned git dependency
# rust-seq2kminmers, Cargo.toml:30) and an MI355X.  Not run in the authoring container (no cargo, no network).
#
#   tools/check_against_upstream.sh <reads.fa|fq[.gz]> <reference.fa> [extra mapquik flags: -k -l -d --nohpc ...]
#
# Step 1 (decides the reading): the reference is patched to print every reference k-min-mer it indexes (src/mers.rs:29: the commented
#   `println!("{:?}", kminmer)` becomes a tab-separated line under MQ_DUMP) and is run TWICE -- in its default hash mode (HashMode::HpcSimd /
#   Simd, src/mers.rs:22-23) and with --nosimd (the crate's scalar modes; D12) -- and tools/upstream_compare.py compares each dump with this
#   repo's CPU oracle for ALL 64 combinations of the six switchable decisions (oracle/mapquik_oracle.c mqo_set_variant = mq_params.flags bits
#   8..13; DESIGN.md section 2 says which kernel constant each one changes), in two layers:
#      positions       (start, end, offset, rev) per tuple                  -- D1-D8
#      hash partition  the hash column only as "which tuples are equal"     -- all the reference asks of the tuple hash (src/index.rs:100-104)
#   so that a crate whose tuple hash is NOT SipHash-1-3 over [len, m...] (a guess) cannot hide a variant that reproduces every
#   position, offset and strand.  One machine-readable line per mode; if the two modes pick different variants the reference's results
#   depend on the CPU it runs on, and that is reported.
#      bit 1  D3  strict `<` on the density bound      bit 8   D5  position = end of the homopolymer run (not its head)
#      bit 2  D2  the bound computed in f32            bit 16  D6  end = raw position of the window's last compressed base
#      bit 4  D2/D12  32-bit ntHash and bound          bit 32  D8  rev on `<=` (palindromic tuples)
# Step 2: maps the reads with the reference and with this repo's native driver (at the variant step 1 found for the mode the extra flags
#   select) and diffs the PAFs byte for byte in input order (the reference's default seq_io path writes in input order, src/closures.rs:117-123).
# The last line is ONE JSON object: {"variant": v, "positions": ..., "hash_partition": ..., "hash_values": ..., "simd_variant": ..,
#   "nosimd_variant": .., "modes_agree": bool, "paf": "identical" | "different", "crate": "<resolved revision of rust-seq2kminmers>"}; exit code 0 iff the PAFs are identical.
# On a whole human genome the 64-variant search seeds the first record once per wrong variant and the whole reference once per matching one:
#   use E. coli or one chromosome for step 1 if that is too long.
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

# ---- step 1: reference k-min-mer tuples, in both hash modes (single worker thread so that the dump is in file order)
OFLAGS=()
args=("$@"); i=0; USER_NOSIMD=0
while [ $i -lt ${#args[@]} ]; do
  case "${args[$i]}" in
    -k|-l) OFLAGS+=("${args[$i]}" "${args[$((i+1))]}"); i=$((i+2));;
    -d|--density) OFLAGS+=(-d "${args[$((i+1))]}"); i=$((i+2));;
    --nohpc) OFLAGS+=(--nohpc); i=$((i+1));;
    --nosimd) USER_NOSIMD=1; i=$((i+1));;
    *) i=$((i+1));;
  esac
done
FLAGS_NO_NOSIMD=(); for a in "$@"; do [ "$a" = "--nosimd" ] || FLAGS_NO_NOSIMD+=("$a"); done
for mode in simd nosimd; do
  EXTRA=(); [ $mode = nosimd ] && EXTRA=(--nosimd)
  MQ_DUMP=1 "$WORK/mapquik/target/release/mapquik" "$READS" --reference "$REF" -p "$WORK/dump_$mode" --threads 1 "${FLAGS_NO_NOSIMD[@]}" "${EXTRA[@]}" 2> "$WORK/upstream.$mode.raw" > /dev/null || true
  grep '^KMM' "$WORK/upstream.$mode.raw" > "$WORK/upstream.$mode.kmm" || true
  echo "== hash mode: $mode ($(wc -l < "$WORK/upstream.$mode.kmm") reference k-min-mers dumped)"
  python3 "$HERE/tools/upstream_compare.py" "$WORK/upstream.$mode.kmm" "$REF" "${OFLAGS[@]}" --tag $mode | tee "$WORK/compare.$mode.txt" || true
  tail -n 1 "$WORK/compare.$mode.txt" > "$WORK/compare.$mode.json"
done
V_SIMD=$(python3 -c "import json; print(json.load(open('$WORK/compare.simd.json'))['variant'])")
V_NOSIMD=$(python3 -c "import json; print(json.load(open('$WORK/compare.nosimd.json'))['variant'])")
if [ "$V_SIMD" != "$V_NOSIMD" ]; then
  echo "THE TWO HASH MODES PICK DIFFERENT READINGS (default: $V_SIMD, --nosimd: $V_NOSIMD): the reference's output depends on the CPU features it was built for"
fi
if [ $USER_NOSIMD = 1 ]; then best=$V_NOSIMD; MODE=nosimd; else best=$V_SIMD; MODE=simd; fi
[ "$best" = "None" ] && best=""
[ "$best" = "0" ] && echo "seeding stage PINNED ($MODE): the frozen reading (variant 0) reproduces the crate" || echo "seeding stage NOT pinned by variant 0 ($MODE; matching variant: '${best:-none}'): run the product with --seeding-variant ${best:-?} (mq_params.flags bits 8..13; DESIGN.md section 2)"

# ---- step 2: PAF identity
"$WORK/mapquik/target/release/mapquik" "$READS" --reference "$REF" -p "$WORK/upstream" "$@"
# the HIP product runs the reading that step 1 matched (variant 0 when none did: the diff below then shows what the mismatch costs)
"$HERE/mapquik_amd/lib/mapquik" "$READS" --reference "$REF" -p "$WORK/hip" --seeding-variant "${best:-0}" "$@"
PAF=different; cmp -s "$WORK/upstream.paf" "$WORK/hip.paf" && PAF=identical
if [ $PAF = identical ]; then
  echo "IDENTICAL: $(wc -l < "$WORK/hip.paf") PAF lines"
else
  echo "DIFFERENT: see $WORK/upstream.paf vs $WORK/hip.paf"; diff "$WORK/upstream.paf" "$WORK/hip.paf" | head -20 || true
  python3 "$HERE/tools/paf_concordance.py" "$WORK/upstream.paf" "$WORK/hip.paf" || true
fi
CRATE=$(tr '\n' ' ' < "$WORK/seq2kminmers.rev" | tr -d '"')
python3 - "$WORK/compare.simd.json" "$WORK/compare.nosimd.json" "$MODE" "$PAF" "$CRATE" <<'PY'
import json, sys
simd, nosimd = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
a = nosimd if sys.argv[3] == "nosimd" else simd
print(json.dumps(dict(variant=a["variant"], positions=a["positions"], hash_partition=a["hash_partition"], hash_values=a["hash_values"],
                      matching_variants=a["matching_variants"], tuples=a["tuples"], mode=sys.argv[3], simd_variant=simd["variant"], nosimd_variant=nosimd["variant"],
                      modes_agree=simd["variant"] == nosimd["variant"], paf=sys.argv[4], crate=sys.argv[5])))
PY
[ $PAF = identical ]
