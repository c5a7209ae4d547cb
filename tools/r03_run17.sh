for o in "" "cg_variant=1"; do
echo "== MISTARK_OPTIONS=$o"; MISTARK_OPTIONS=$o python tools/config_rates.py cfg0 cfg1 cfg2 cfg4 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k: (round(v,3) if isinstance(v,float) else v) for k,v in d.items() if not isinstance(v,(dict,list))})"
done
