export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/r02_base_bench.json 2> gpurun_out/r02_base_bench.err
tail -c 300 gpurun_out/r02_base_bench.err
rm -rf /tmp/prof_kt
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o r -- python bench.py --no-cpu-baseline > /tmp/kt.log 2>&1
db=$(find /tmp/prof_kt -name "*.db" | head -1)
python profiles/summarize_rocpd.py $db > gpurun_out/r02_base_kernel_stats.txt
python profiles/iter_rocpd.py $db > gpurun_out/r02_base_iter.txt
python profiles/iter_rocpd.py $db 8 > gpurun_out/r02_base_iter8.txt
{ python profiles/solve_rocpd.py $db; python profiles/idle_rocpd.py $db 8 0.5; } > gpurun_out/r02_base_timeline.txt
