for o in "" "atomic_projection=1" "no_dyn_pool=1" "no_row_order=1" "atomic_projection=1,no_dyn_pool=1,no_row_order=1"; do
echo "== MISTARK_OPTIONS=$o"; MISTARK_OPTIONS=$o python tools/config_rates.py cfg4 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k: (round(v,3) if isinstance(v,float) else v) for k,v in d.items() if not isinstance(v,(list))})"
done
