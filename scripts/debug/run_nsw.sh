for n in 3 4 5 6 7; do echo "== nsw $n"; DFM_PASS_NSW=$n timeout 120 python scripts/pf_prof.py 2>/dev/null | grep -E "stream wave0|stream last|iteration|span"; done
