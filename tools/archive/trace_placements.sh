export TMPDIR=/tmp; mkdir -p gpurun_out
for tag in c o; do
  off="0,0"; [ $tag = o ] && off="0.00137,-0.00053"
  rm -rf /tmp/pk; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pk -o r -- python bench.py --no-cpu-baseline --no-extras --offset=$off > /tmp/kt.log 2>&1
  db=$(find /tmp/pk -name "*.db" | head -1)
  python profiles/summarize_rocpd.py $db > gpurun_out/s3_${tag}_kernel_stats.txt
  python profiles/iter_rocpd.py $db > gpurun_out/s3_${tag}_iter.txt
  for w in 15 16 17; do python profiles/abs_rocpd.py $db $w; done > gpurun_out/s3_${tag}_abs.txt
  { python profiles/solve_rocpd.py $db; python profiles/idle_rocpd.py $db 8 0.5; } > gpurun_out/s3_${tag}_timeline.txt
  grep '^{' /tmp/kt.log | cut -c1-200
done
