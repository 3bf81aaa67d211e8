# A/B of prebuilt libraries gpurun_tmp/lib*.so inside ONE gpurun call (box-to-box variance is +-5 %): 10 %-missing pass,
# headline, VAR(4) companion EM; two rounds
R=$GRAFT_REPO_ROOT; cd $R
p() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],4), d['roofline']['kernels_ms'])"; }
for round in 1 2; do for L in gpurun_tmp/lib*.so; do cp $L dynamic_factor_models_amd/lib/libdfmhip.so; echo "== $L"
python bench.py --missing 0.1 --steps 20 --warmup 3 --repeats 5 --no-cpu-baseline 2>/dev/null | p missing10
[ -z "$SHORT" ] && python bench.py --no-cpu-baseline --repeats 5 2>/dev/null | p headline
[ -n "$C4" ] && python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --missing 0.1 --steps 3 --warmup 1 --repeats 3 --no-cpu-baseline 2>/dev/null | p c4_missing10
[ -z "$SHORT" ] && python scripts/dbg/varp_ab.py 2>&1 | grep -v amdgpu.ids | tr "\n" " "; echo
done; done
