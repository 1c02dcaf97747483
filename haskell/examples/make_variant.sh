#!/bin/bash
# usage: make_variant.sh /path/to/ALCHEMY-checkout OUTDIR
# Writes OUTDIR/Arithmetic.hs, HomomRLWR.hs and Tunnel.hs = the reference's examples with the lol-cpp tensor `CT`
# replaced by the MI355X tensor `GT` (import line and plaintext alias; reference examples/Arithmetic.hs:19,23,
# examples/HomomRLWR.hs:24,47, examples/Tunnel.hs:20,41).  The reference's sources are read from the checkout, never
# stored in this repository.
set -e
ref=${1:?reference checkout}; out=${2:?output directory}
mkdir -p "$out"
for f in Arithmetic HomomRLWR Tunnel; do
  sed -e 's/^import Crypto\.Lol\.Cyclotomic\.Tensor\.CPP$/import Crypto.Lol.Cyclotomic.Tensor.GT/' \
      -e 's/^\(type PT = PNoiseCyc PNZ \)CT /\1GT /' \
      -e 's/\(PNoiseCyc [A-Za-z0-9]* \)CT /\1GT /g' \
      "$ref/examples/$f.hs" > "$out/$f.hs"
  if grep -n 'Tensor\.CPP\| CT ' "$out/$f.hs" | grep -v SymmSHE | grep -q 'Tensor\.CPP'; then echo "$f.hs: CPP import left" >&2; exit 1; fi
done
cp "$ref/examples/Common.hs" "$out/Common.hs"
echo "wrote $out/{Arithmetic,HomomRLWR,Tunnel,Common}.hs"
