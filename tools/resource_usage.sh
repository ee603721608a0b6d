#!/bin/bash
# Register / spill / LDS / occupancy figures of every kernel of libpokerl_hip.so as the compiler reports them
# (-Rpass-analysis=kernel-resource-usage), with the library's own build flags.  usage: tools/resource_usage.sh > profiles/rNN_resource_usage.txt
cd "$(dirname "$0")/.."
FLAGS=$(python3 -c "from pokerl_amd import build; print(' '.join(f for f in build.FLAGS if f not in ('-shared',)))")
/opt/rocm/bin/hipcc $FLAGS -Rpass-analysis=kernel-resource-usage -c pokerl_amd/csrc/pk_api.hip -o /tmp/pk_api_ru.o 2>&1 | python3 -c "
import re, sys
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r'remark: .*Function Name: (\S+)', line)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r'remark:\s+([A-Za-z][A-Za-z /\[\]]*?): (\S+) \[-Rpass', line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
import subprocess
names = {k: subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip() for k in rows}
keys = ['VGPRs', 'AGPRs', 'TotalSGPRs', 'VGPRs Spill', 'SGPRs Spill', 'ScratchSize [bytes/lane]', 'Occupancy [waves/SIMD]', 'LDS Size [bytes/block]']
print('%-62s' % 'kernel' + ''.join('%12s' % k.split(' [')[0][:11] for k in keys))
for k in sorted(rows, key=lambda x: names[x]):
    short = re.sub(r'\(.*', '', names[k]).replace('void ', '')
    print('%-62s' % short[:62] + ''.join('%12s' % rows[k].get(kk, '-') for kk in keys))
"
