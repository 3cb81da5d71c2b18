#!/bin/bash
# Round 6, review item 2: tests/test_pipeline_ordering.py::test_removing_any_wait_that_orders_something_is_reported names a wait by
# (stream, ordinal among that stream's waits); 30 runs pinned to ONE cpu and 30 runs beside busy loops on every cpu must all pass
# (round 5's global wait counter failed 1 run in 5 here).  CPU only.
cd "$(dirname "$0")/.."
out=profiles/r06_ordering_determinism.txt
echo "# $(date -u +%FT%RZ), $(nproc) cpus: 30 x taskset -c 0, then 30 x beside $(nproc) busy loops" > $out
pass=0
for i in $(seq 30); do taskset -c 0 python -m pytest tests/test_pipeline_ordering.py -q -x -k "removing_any_wait and (scenario0 or scenario7)" 2>&1 | tail -1 | grep -q "2 passed" && pass=$((pass+1)); done
echo "taskset -c 0: $pass of 30 runs passed (scenarios cpra-host world 2 and cpra-host world 2 grouped)" >> $out
pids=""; for c in $(seq $(nproc)); do ( while :; do :; done ) & pids="$pids $!"; done
pass=0
for i in $(seq 30); do python -m pytest tests/test_pipeline_ordering.py -q -x -k "removing_any_wait and (scenario0 or scenario7)" 2>&1 | tail -1 | grep -q "2 passed" && pass=$((pass+1)); done
kill $pids
echo "beside $(nproc) busy loops: $pass of 30 runs passed" >> $out
cat $out
