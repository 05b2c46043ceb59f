"""Registers, scratch and LDS of every kernel in a device-assembly file.   python tools/kernel_resources.py [file.s]
Without an argument the assembly is produced first (python __graft_entry__.py --asm: the shipped build's flags)."""
import os, re, subprocess, sys
if len(sys.argv) < 2:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs("/tmp/w", exist_ok=True)
    subprocess.run([sys.executable, os.path.join(root, "__graft_entry__.py"), "--asm", "/tmp/w/lib.s"], check=True)
    sys.argv.append("/tmp/w/lib.s")
s = open(sys.argv[1]).read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: re.search(r'\.amdhsa_' + k + r' (\d+)', body).group(1)
    try:
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    except OSError:
        dem = name
    print('%-78s vgpr %3s (acc from %3s) scratch %4s B  lds %6s B' % (dem[:78], g('next_free_vgpr'), g('accum_offset'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
