#!/bin/bash
mkdir -p /tmp/w
# compile the device code to ISA and summarise the wave-specialised stepper: registers, scratch, spill ops between barriers, spill check
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 "$ROOT/__graft_entry__.py" --asm /tmp/w/duo.s "$@" 2>&1 | grep -v "argument unused" | head -30
cd "$ROOT"
K=_ZN3fbd10k_step_duoILi0ELb0ELb0EEEvNS_5KArgsEi   # k_step_duo<WA, false, false>
awk "/\.amdhsa_kernel $K/,/\.end_amdhsa_kernel/" /tmp/w/duo.s | grep -E "next_free_vgpr|accum_offset|private_segment_fixed|group_segment"
awk "/^$K:/,/\.end_amdhsa_kernel/" /tmp/w/duo.s > /tmp/w/duo_k.s
grep -n "s_barrier\|scratch_" /tmp/w/duo_k.s | python3 -c "
import sys
cnt=0
for l in sys.stdin:
    if 'scratch_' in l: cnt+=1
    else:
        print(l.split(':')[0], 'barrier; scratch ops since last:', cnt); cnt=0
print('tail', cnt)"
python3 tools/check_isa_spills.py /tmp/w/duo.s 2>&1 | tail -4
