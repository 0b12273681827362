for i in 1 2 3 4 5 6; do timeout 100 python tools/dbg_nw.py 2>&1 | grep "bad iterations" | tr '\n' ' '; echo; done
timeout 300 python tools/dbg_flaky.py 40 | head -3
