#!/bin/bash
# harness/check.sh [SHA256SUMS] -- the reference's `sha256sum -c` gate
# (runtests.sh:24-28 there) run on the GPU path: the six test_vectors streams
# against SHA256SUMS (default: tests/golden/SHA256SUMS, a copy of the
# reference's fixture) and KAT record 0 against tests/golden/meta.json.
set -e -o pipefail
here=$(cd "$(dirname "$0")" && pwd)
sums=${1:-$here/../tests/golden/SHA256SUMS}
fail=0
while read -r want name; do
  set_=${name#tvecs}
  got=$("$here/test_vectors_hip" "$set_" | sha256sum | cut -d' ' -f1)
  if [ "$got" = "$want" ]; then echo "$name: OK"; else echo "$name: FAILED"; fail=1; fi
  kat=$("$here/kat_hip" "$set_" --record0 | sha256sum | cut -d' ' -f1)
  wantkat=$(python3 -c "import json,sys; print(json.load(open('$here/../tests/golden/meta.json'))['Kyber$set_']['nistkat-sha256'])")
  if [ "$kat" = "$wantkat" ]; then echo "kat$set_ record 0: OK"; else echo "kat$set_ record 0: FAILED"; fail=1; fi
done < "$sums"
exit $fail
