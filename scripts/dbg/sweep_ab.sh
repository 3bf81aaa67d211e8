# quick numbers for a change in Grid::sweep_inverse: headline, EM, 10 %-missing pass, VAR(4) companion EM (Rp = 16), parity subset
R=$GRAFT_REPO_ROOT; cd $R
p() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],4), d['roofline']['kernels_ms'])"; }
python bench.py --no-cpu-baseline --repeats 5 2>/dev/null | p headline
python bench.py --missing 0.1 --steps 20 --warmup 3 --repeats 5 --no-cpu-baseline 2>/dev/null | p missing10
python bench.py --mode em --steps 20 --warmup 3 --repeats 5 --no-cpu-baseline 2>/dev/null | p em
python scripts/dbg/varp_ab.py 2>&1 | grep -v amdgpu.ids | tr "\n" " "; echo
[ -n "$TESTS" ] && timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -5
