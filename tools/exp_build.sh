#!/bin/bash
# Builds one experiment library per line of the list (NAME:-Dflags ...), here (no GPU needed); tools/gpu_exp.sh times them.
set -eu
LIST=${1:-tools/exp_list.txt}
while IFS= read -r line; do
  [ -z "$line" ] && continue
  case "$line" in *:@*) continue;; esac
  DMI_EXP="$line" python -c "
from cudadepthmapintegration_amd import build
print(build.build(verbose=False))" 2>&1 | grep -v hip-link
done < "$LIST"
