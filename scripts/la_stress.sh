cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${RUNS:-24}); do TSL_PARAMS=direct_lookahead=${LA:-15} STEPS=14 python scripts/probe_flow_abort.py 2>&1 | python -c "
import sys, re
slow = []; ab = ''
for ln in sys.stdin:
    m = re.match(r'step (\d+): +([0-9.]+) ms', ln)
    if m and int(m.group(1)) > 1 and float(m.group(2)) > 215: slow.append((int(m.group(1)), float(m.group(2))))
    if 'tsl' in ln: ab = ln[ln.find('(workgroup'):].strip()
print('slow steps', slow, ab)"; done
