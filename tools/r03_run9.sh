export MISTARK_SHIM_STATS=1 SHIM_THREADS=16
for scene in benchclamped benchblock; do
echo "== shim $scene 1M"; SHIM_GRID=44,44,43 timeout 900 oracle/_ref/shim_check $scene 6 2>&1 | grep -v "^shim_check" | tail -3
done
