#!/bin/bash
# round 3, GPU session 22: the three-wave pair build de-phased on every grid above 1 024 workgroups — phase groups x cycles for the
# short grids (1 025 .. 4 096 workgroups), one session.  a_base = the tree before the change.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s22; mkdir -p $out
sizes="147456 163840 196608 229376 262144 294912 327680 360448 393216 458752 524288 1000000"
for lib in a_base c_9x7000 c_9x4000 c_6x7000 c_12x5000 c_9x10000; do
  NPF16_LIB=tools/microbench/libs/$lib.so timeout 300 python tools/microbench/mid_n.py --variants auto --steps 300 --out $out/${lib}_auto.json $sizes 2>/dev/null | grep "N=" | sed "s/^/$lib /"
done > $out/table.log 2>&1
python - <<'P'
import re,collections
t=collections.defaultdict(dict)
for l in open('gpurun_out/r03_s22/table.log'):
    m=re.match(r'(\S+)\s+N=\s*(\d+).*kernel\s+([\d.]+) us', l)
    if m: t[int(m.group(2))][m.group(1)]=float(m.group(3))
libs=['a_base','c_9x7000','c_9x4000','c_6x7000','c_12x5000','c_9x10000']
print('N'.rjust(8),*[x.rjust(10) for x in libs])
for n in sorted(t): print(str(n).rjust(8),*[f"{t[n].get(x,0):10.1f}" for x in libs])
P
