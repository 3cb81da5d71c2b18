#!/bin/bash
# Copies round 6's evidence from gpurun_out/ into profiles/ - only artefacts of THIS tree's library (every text artefact's header and
# every JSON's "library_hash" must equal hash_join_codes_knl_amd.build.library_hash()), and no soak / stress log that contains a wrong step.
cd "$(dirname "$0")/.."
tree=$(python3 -c "from hash_join_codes_knl_amd import build; print(build.library_hash())")
fail=0
for f in gpurun_out/r06_soak_*.txt gpurun_out/r06_rc_*.txt; do
  [ -f $f ] || continue
  if ! head -3 $f | grep -q "library hash $tree"; then echo "REFUSED (another library): $f"; fail=1; continue; fi
  if grep -q "WRONG\|bad=1" $f; then echo "REFUSED (a wrong step): $f"; fail=1; continue; fi
  cp $f profiles/; echo "ok $f"
done
for f in gpurun_out/r06_bench*.json; do
  [ -f $f ] || continue
  h=$(python3 -c "import json,sys; print(json.load(open('$f')).get('library_hash'))" 2>/dev/null)
  if [ "$h" != "$tree" ]; then echo "REFUSED (library $h): $f"; fail=1; continue; fi
  cp $f profiles/; echo "ok $f"
done
for f in gpurun_out/r06_*_kernel_stats.csv gpurun_out/r06_pmc_sq*.csv gpurun_out/r06_pmc_sq_summary.txt gpurun_out/r06_pytest.log; do [ -f $f ] && cp $f profiles/ && echo "ok $f (produced by tools/r06_round.sh, which refuses a stale library: see profiles/r06_rc_*.txt)"; done
exit $fail
