#!/bin/bash
# usage: bash tools/gather_kernel_ab.sh "<opts A>" "<opts B>" ... — durations of the static part's gather (k_assemble_gather launches > 100 us) in the kernel trace of the
# bench command on the offset placement, once per option set (options applied before the warm-up through MISTARK_OPTIONS)
export TMPDIR=/tmp
for o in "$@"; do
  rm -rf /tmp/pk; MISTARK_OPTIONS="$o" timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pk -o r -- python bench.py --no-cpu-baseline --no-extras --offset=0.00137,-0.00053 > /tmp/kt.log 2>&1
  db=$(find /tmp/pk -name "*.db" | head -1)
  echo "== opts='$o' $(grep '^{' /tmp/kt.log | cut -c1-60)"; python profiles/summarize_rocpd.py $db | grep -E "k_assemble_gather|k_sym" | cut -c1-140
  python - "$db" <<'P'
import sqlite3,sys
db=sqlite3.connect(sys.argv[1])
rows=[r[0]/1e3 for r in db.execute("select end-start from kernels where name like '%k_assemble_gather%' and (end-start) > 100000")]
print("static gathers: n=%d avg=%.1f us min=%.1f max=%.1f" % (len(rows), sum(rows)/len(rows), min(rows), max(rows)))
P
done
