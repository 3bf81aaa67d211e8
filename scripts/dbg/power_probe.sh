#!/bin/bash
# is the box power-capped?  the cap, and clocks / power sampled while the headline pass runs in a loop
cd $GRAFT_REPO_ROOT
rocm-smi --showmaxpower --showpower --showperflevel 2>/dev/null | grep -v "^=\|^$" | head
rocm-smi --showclkfrq 2>/dev/null | grep -i "sclk\|S:\|[0-9]: " | head -12
echo "driver: $(cat /sys/module/amdgpu/version 2>/dev/null) kernel: $(uname -r)"
rocm-smi --showvbios --showcomputepartition --showmemorypartition --showdriverversion 2>/dev/null | grep -v "^=\|^$" | tr -s '\t ' ' ' | head -8
rocm-smi --showfwinfo 2>/dev/null | grep -v "^=\|^$" | tr -s '\t ' ' ' | tr '\n' ';' ; echo
cat /proc/cmdline | tr ' ' '\n' | grep -i "amdgpu\|iommu\|pci" | tr '\n' ' '; echo
cat > /tmp/loop.py <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
from dynamic_factor_models_amd import DfmContext
c = DfmContext(0)
panel, par = c.synth_panels(1, 0, 1024, 500, 200, 8)
for _ in range(50): c.ks_pass_batch(panel, *par, may_have_missing=False)
torch.cuda.synchronize()
print("loop start", flush=True)
t_end = time.time() + float(sys.argv[1])
n = 0; t0 = time.time()
while time.time() < t_end:
    for _ in range(200): c.ks_pass_batch(panel, *par, may_have_missing=False)
    torch.cuda.synchronize(); n += 200
print("ms per pass %.4f" % ((time.time() - t0) / n * 1e3), flush=True)
PY
python /tmp/loop.py 9 > /tmp/loop.log 2>&1 &
LP=$!
sleep 0.5
while ! grep -q "loop start" /tmp/loop.log 2>/dev/null; do sleep 0.2; kill -0 $LP 2>/dev/null || break; done
for k in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "sclk\|Power\|junction\|mclk" | tr -s '\t ' ' ' | tr '\n' '|'; echo
  sleep 1
done
wait $LP
grep "ms per" /tmp/loop.log
timeout 60 scripts/microbench/icache
timeout 60 python bench.py --no-cpu-baseline --no-secondary --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('bench value=%.4g'%d['value'])"
