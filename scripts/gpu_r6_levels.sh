for c in 3 5 4; do for m in full moved; do echo "== config $c $m"; ABC_DIAG=1 ABC_WX_DEBUG=1 timeout 300 python3 scripts/trace_step.py $c $m 1 2>&1 | grep WX_LEVELS | sort | uniq -c; done; done
