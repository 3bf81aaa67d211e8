#!/bin/bash
# the forced 1-rank communicator run that printed nothing inside gpu_r3.sh: same sequence, exit codes and stderr kept
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --driver lib --mode em --steps 20 --warmup 3 --repeats 5 > /tmp/a.json 2> /tmp/a.err; echo "plain rc=$? bytes=$(wc -c < /tmp/a.json)"
timeout 300 python bench.py --driver lib --mode em --force-comm --steps 20 --warmup 3 --repeats 5 > /tmp/b.json 2> /tmp/b.err; echo "comm rc=$? bytes=$(wc -c < /tmp/b.json)"; tail -5 /tmp/b.err
timeout 300 python -X faulthandler bench.py --driver lib --mode em --force-comm --steps 20 --warmup 3 --repeats 5 > /tmp/c.json 2> /tmp/c.err; echo "comm+fh rc=$? bytes=$(wc -c < /tmp/c.json)"; tail -30 /tmp/c.err
NCCL_DEBUG=WARN timeout 300 python bench.py --driver lib --mode em --force-comm --steps 20 --warmup 3 --repeats 5 > /tmp/d.json 2> /tmp/d.err; echo "comm+debug rc=$? bytes=$(wc -c < /tmp/d.json)"; tail -10 /tmp/d.err
